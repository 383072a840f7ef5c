//! `fidget_mesh::Octree::build(..).walk_dual()` as one call (`fhip_mesh_build`): cell classification, leaf sampling, octree
//! assembly (check_done / collapsible, octree.rs:256-470) and the dual walk (dc.rs) all run on the device; the host receives the
//! finished `Mesh`, the single-threaded recursion's, element for element.
use fidget_core::shape::BoundShape;
use fidget_mesh::{Mesh, Settings};

use crate::{axis_slots, ffi, var_key, HipFunction, CTX};

/// Builds the mesh of `shape` at `settings.depth`; `None` when the variables do not bind or the build fails
pub fn mesh(b: BoundShape<HipFunction, f32>, settings: &Settings) -> Option<Mesh> {
    let f = b.shape().inner();
    let tape = f.tape();
    let m = settings.world_to_model.transpose();
    let axes = axis_slots(fidget_core::eval::Function::vars(f));
    let mut keys = vec![];
    let mut vals = vec![];
    for (k, v) in b.vars() {
        keys.push(var_key(*k));
        vals.push(*v);
    }
    let mut h = std::ptr::null_mut();
    let st = CTX.with(|c| unsafe {
        ffi::fhip_mesh_build(c.raw(), tape.raw(), settings.depth as u32, m.as_ptr(), axes.as_ptr(), keys.as_ptr(), vals.as_ptr(),
                             keys.len() as u32, &mut h)
    });
    if st != 0 {
        return None;
    }
    let mut n = [0u64; 8];
    unsafe { ffi::fhip_mesh_counts(h, n.as_mut_ptr()) };
    let mut vertices = vec![nalgebra::Vector3::<f32>::zeros(); n[6] as usize]; // repr(C) [f32; 3]
    let mut triangles = vec![nalgebra::Vector3::<usize>::zeros(); n[7] as usize]; // [u64; 3] on 64-bit targets
    unsafe {
        ffi::fhip_mesh_vertices(h, vertices.as_mut_ptr().cast());
        ffi::fhip_mesh_triangles(h, triangles.as_mut_ptr().cast());
        ffi::fhip_mesh_free(h);
    }
    Some(Mesh { vertices, triangles })
}
