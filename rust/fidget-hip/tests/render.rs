//! The fused renders against the reference's VM on the same shapes: `fidget_hip::render::render3d` must give the image of
//! `fidget_raster::voxel::render` on `VmShape` bit for bit (a 3D image does not depend on the tile sizes), `render2d` the image of
//! `pixel::render` with the HIP shape's tile sizes (a 2D fill carries the level it was decided at, pixel.rs:225-229).
use fidget_core::{context::Tree, render::TileSizes, shape::Shape, vm::VmFunction};
use fidget_hip::{HipFunction, HipShape};
use fidget_raster::{pixel, voxel};

fn sphere() -> Tree {
    let (x, y, z) = Tree::axes();
    (x.square() + y.square() + z.square()).sqrt() - 0.6
}

#[test]
fn render3d_equals_the_vm() {
    let t = sphere();
    let hip: HipShape = Shape::<HipFunction>::from(t.clone());
    let vm = Shape::<VmFunction>::from(t);
    let cfg = voxel::RenderConfig { image_size: voxel::RenderSize::from(128), world_to_model: nalgebra::Matrix4::identity() };
    let eval = voxel::EvalConfig { tile_sizes: None, threads: None, cancel: Default::default() };
    let a = fidget_hip::render::render3d(hip.try_into().unwrap(), &cfg, &eval).unwrap();
    let b = voxel::render(vm.try_into().unwrap(), &cfg, &eval).unwrap();
    for (p, q) in a.iter().zip(b.iter()) {
        assert_eq!(p.depth, q.depth);
        assert_eq!(p.normal.map(f32::to_bits), q.normal.map(f32::to_bits));
    }
}

#[test]
fn render2d_equals_the_vm_with_the_same_tile_sizes() {
    let t = sphere();
    let hip: HipShape = Shape::<HipFunction>::from(t.clone());
    let vm = Shape::<VmFunction>::from(t);
    let cfg = pixel::RenderConfig { image_size: pixel::RenderSize::from(256), world_to_model: nalgebra::Matrix3::identity(), z: 0.0, pixel_perfect: false };
    let tiles = TileSizes::new(&[128, 16]).unwrap();
    let eval = pixel::EvalConfig { tile_sizes: Some(tiles), threads: None, cancel: Default::default() };
    let a = fidget_hip::render::render2d(hip.try_into().unwrap(), &cfg, &eval).unwrap();
    let b = pixel::render(vm.try_into().unwrap(), &cfg, &eval).unwrap();
    assert!(a.as_bytes() == b.as_bytes());
}
