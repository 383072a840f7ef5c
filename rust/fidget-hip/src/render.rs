//! The throughput path: `fidget_raster::voxel::render` / `pixel::render` as ONE call each - the whole tile recursion runs on the
//! device (`fhip_render3d` / `fhip_render2d`), no host round trip per tile.  Same arguments, same image (bit for bit for the HIP
//! shape's tile sizes; a 3D image does not depend on the tile sizes at all), `None` when cancelled.
use fidget_core::shape::BoundShape;
use fidget_raster::{pixel, voxel, voxel::GeometryPixel};

use crate::{axis_slots, ffi, var_key, HipFunction, CTX};

/// The EvalConfig's cancel token (voxel.rs:52-61) handed to the context for the duration of a render: the token is an Arc<AtomicBool>,
/// `into_raw` gives the flag's address, the library's per-level checks read it beside the context's own flag (fhip_cancel_watch) - a cancel
/// from any thread ends the render with `None`, as the reference's workers do.  Reclaimed when the guard drops.
struct Watch(*const std::sync::atomic::AtomicBool);
impl Watch {
    fn new(token: &fidget_core::render::CancelToken) -> Self {
        let raw = token.clone().into_raw();
        CTX.with(|ctx| unsafe { ffi::fhip_cancel_watch(ctx.raw(), raw.cast()) });
        Watch(raw)
    }
}
impl Drop for Watch {
    fn drop(&mut self) {
        CTX.with(|ctx| unsafe { ffi::fhip_cancel_watch(ctx.raw(), std::ptr::null()) });
        drop(unsafe { fidget_core::render::CancelToken::from_raw(self.0) });
    }
}

fn bound_vars(b: &BoundShape<HipFunction, f32>) -> (Vec<u64>, Vec<f32>) {
    // ShapeVars<f32>: Var::V index -> value (shape/mod.rs:190-232); the keys the tape asks for are b.shape().inner().vars()'s
    let vars = b.vars();
    let mut keys = vec![];
    let mut vals = vec![];
    for (k, v) in vars {
        keys.push(var_key(*k));
        vals.push(*v);
    }
    (keys, vals)
}

/// `fidget_raster::voxel::render(shape, &cfg, &eval)` (voxel.rs:500-553) on the device; `eval.tile_sizes` None = the HIP hints
pub fn render3d(
    b: BoundShape<HipFunction, f32>,
    cfg: &voxel::RenderConfig,
    eval: &voxel::EvalConfig,
) -> Option<voxel::Image> {
    let f = b.shape().inner();
    let tape = f.tape();
    let m = cfg.world_to_model.transpose(); // nalgebra is column major, the ABI row major
    let axes = axis_slots(fidget_core::eval::Function::vars(f));
    let (keys, vals) = bound_vars(&b);
    let tiles: Option<Vec<u32>> = eval.tile_sizes.as_ref().map(|t| t.iter().map(|&v| v as u32).collect());
    let c = ffi::fhip_render3d_config {
        width: cfg.image_size.width(),
        height: cfg.image_size.height(),
        depth: cfg.image_size.depth(),
        world_to_model: m.as_ptr(),
        tile_sizes: tiles.as_ref().map(|t| t.as_ptr()).unwrap_or(std::ptr::null()),
        n_tile_sizes: tiles.as_ref().map(|t| t.len() as u32).unwrap_or(0),
        var_keys: keys.as_ptr(),
        var_values: vals.as_ptr(),
        n_vars: keys.len() as u32,
        axis_slots: axes.as_ptr(),
    };
    let mut out = voxel::Image::new(cfg.image_size); // repr(C) {normal: [f32; 3], depth: u32}: 16 B
    if eval.cancel.is_cancelled() {
        return None;
    }
    let _watch = Watch::new(&eval.cancel);
    let st = CTX.with(|ctx| unsafe { ffi::fhip_render3d(ctx.raw(), tape.raw(), &c, (&mut out[0] as *mut GeometryPixel).cast(), 0) });
    match st {
        0 => Some(out),
        8 => None, // FHIP_ERR_CANCELLED
        e => panic!("fidget-hip: fhip_render3d status {e}"),
    }
}

/// `fidget_raster::pixel::render` (pixel.rs:452-492) on the device: raw distance pixels
pub fn render2d(
    b: BoundShape<HipFunction, f32>,
    cfg: &pixel::RenderConfig,
    eval: &pixel::EvalConfig,
) -> Option<pixel::Image> {
    let f = b.shape().inner();
    let tape = f.tape();
    let m = cfg.world_to_model.transpose();
    let axes = axis_slots(fidget_core::eval::Function::vars(f));
    let (keys, vals) = bound_vars(&b);
    let tiles: Option<Vec<u32>> = eval.tile_sizes.as_ref().map(|t| t.iter().map(|&v| v as u32).collect());
    let c = ffi::fhip_render2d_config {
        width: cfg.image_size.width(),
        height: cfg.image_size.height(),
        world_to_model: m.as_ptr(),
        z: cfg.z,
        pixel_perfect: cfg.pixel_perfect as i32,
        tile_sizes: tiles.as_ref().map(|t| t.as_ptr()).unwrap_or(std::ptr::null()),
        n_tile_sizes: tiles.as_ref().map(|t| t.len() as u32).unwrap_or(0),
        var_keys: keys.as_ptr(),
        var_values: vals.as_ptr(),
        n_vars: keys.len() as u32,
        axis_slots: axes.as_ptr(),
    };
    let mut out = pixel::Image::new(cfg.image_size); // f32 bit patterns (pixel.rs:159-241)
    if eval.cancel.is_cancelled() {
        return None;
    }
    let _watch = Watch::new(&eval.cancel);
    let st = CTX.with(|ctx| unsafe { ffi::fhip_render2d(ctx.raw(), tape.raw(), &c, (&mut out[0] as *mut pixel::RawDistancePixel).cast(), 0) });
    match st {
        0 => Some(out),
        8 => None,
        e => panic!("fidget-hip: fhip_render2d status {e}"),
    }
}
