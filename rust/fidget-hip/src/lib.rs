//! MI355X evaluators for Fidget: `HipFunction` implements `fidget_core::eval::{Function, MathFunction}` and
//! `fidget_core::render::RenderHints` on top of `libfidget_hip.so`, so that `fidget-raster` and `fidget-mesh` run unchanged on
//! `HipShape`; `render` / `mesh` replace the raster and mesh entry points with the library's fused, device-resident versions.
//!
//! The crate mirrors `fidget-jit` (fidget-jit/src/lib.rs:869-1384): the function keeps a `GenericVmFunction<255>`, so
//! `simplify`, `vars`, `size`, `can_simplify` are the VM's own (vm/mod.rs:126-230), and swaps the four tapes and evaluators for
//! device ones.  One `fhip_ctx` per host thread (evaluators are per-thread in the reference too: fidget-raster/src/lib.rs:129-133).
//!
//! NOT COMPILED in the repository this file comes from (no Rust toolchain in its build image); the C side of every call below
//! is exercised by that repository's ctypes binding and C11 caller, and `src/ffi.rs` is held against the header by a test there.
#![warn(missing_docs)]

pub mod ffi;
pub mod mesh;
pub mod render;

use std::ops::Deref;
use std::sync::Arc;

use fidget_core::{
    Context,
    context::{BadNode, Node},
    eval::{
        BulkEvalError, BulkEvaluator, BulkOutput, Function, MathFunction,
        Tape, TracingEvalError, TracingEvaluator,
    },
    render::{RenderHints, TileSizes},
    shape::Shape,
    types::{Grad, Interval},
    var::VarMap,
    vm::{BadTrace, Choice, GenericVmFunction, VmData, VmTrace, VmWorkspace},
};

/// Register count of the host-side VM tape (the device re-allocates densely on import; see `fhip_tape_from_bytecode`)
pub const REGISTER_LIMIT: usize = 255;

/// One device context per host thread
pub(crate) struct Ctx(*mut ffi::fhip_ctx);
impl Ctx {
    fn new(device: i32) -> Self {
        let mut c = std::ptr::null_mut();
        let st = unsafe { ffi::fhip_ctx_create(device, std::ptr::null_mut(), &mut c) };
        assert_eq!(st, 0, "fhip_ctx_create failed: status {st} (no MI355X, or libfidget_hip.so could not load its kernels)");
        Ctx(c)
    }
    pub(crate) fn raw(&self) -> *mut ffi::fhip_ctx {
        self.0
    }
}
impl Drop for Ctx {
    fn drop(&mut self) {
        unsafe { ffi::fhip_ctx_destroy(self.0) }
    }
}
thread_local! {
    pub(crate) static CTX: Ctx = Ctx::new(
        std::env::var("FIDGET_HIP_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0));
}

/// Panics with the library's message: the C ABI reports argument-shape errors only, and those are checked on the host first
/// (`VarMap::check_*_arguments`), so a non-zero status here is a device failure.
pub(crate) fn check(st: ffi::fhip_status) {
    if st != 0 {
        let msg = CTX.with(|c| unsafe {
            let p = ffi::fhip_last_error(c.raw());
            if p.is_null() { String::new() } else { std::ffi::CStr::from_ptr(p).to_string_lossy().into_owned() }
        });
        panic!("fidget-hip: status {st}: {msg}");
    }
}

/// An owned device tape (`fhip_tape`); freed when the last `HipTape` goes
struct DeviceTape(*mut ffi::fhip_tape);
// SAFETY: a device tape is immutable after creation; the library's evaluators take it by const pointer
unsafe impl Send for DeviceTape {}
unsafe impl Sync for DeviceTape {}
impl Drop for DeviceTape {
    fn drop(&mut self) {
        unsafe { ffi::fhip_tape_free(self.0) }
    }
}

/// Handle to a tape in device memory; one serves all four evaluators
#[derive(Clone)]
pub struct HipTape {
    dev: Arc<DeviceTape>,
    vars: Arc<VarMap>,
    choice_count: usize,
    output_count: usize,
}
impl HipTape {
    pub(crate) fn raw(&self) -> *const ffi::fhip_tape {
        self.dev.0
    }
}
impl Tape for HipTape {
    type Storage = ();
    fn recycle(self) -> Option<()> {
        None
    }
    fn vars(&self) -> &VarMap {
        &self.vars
    }
    fn output_count(&self) -> usize {
        self.output_count
    }
}

/// Function for use with the MI355X evaluators
#[derive(Clone)]
pub struct HipFunction(GenericVmFunction<REGISTER_LIMIT>);

impl HipFunction {
    /// Uploads the function as a device tape: `Bytecode::new` of the VM tape is the wire format `fhip_tape_from_bytecode` takes
    /// (fidget-bytecode/src/lib.rs:203-332)
    pub fn tape(&self) -> HipTape {
        let bc = fidget_bytecode::Bytecode::new(self.0.data())
            .expect("register 255 in use");
        let words = bc.data();
        let mut t = std::ptr::null_mut();
        CTX.with(|c| check(unsafe { ffi::fhip_tape_from_bytecode(c.raw(), words.as_ptr(), words.len(), &mut t) }));
        HipTape {
            dev: Arc::new(DeviceTape(t)),
            vars: self.0.data().vars.clone(),
            choice_count: self.0.choice_count(),
            output_count: self.0.output_count(),
        }
    }
}

impl Function for HipFunction {
    type Trace = VmTrace;
    type Storage = VmData<REGISTER_LIMIT>;
    type Workspace = VmWorkspace<REGISTER_LIMIT>;
    type TapeStorage = ();

    type IntervalEval = HipIntervalEval;
    type PointEval = HipPointEval;
    type FloatSliceEval = HipFloatSliceEval;
    type GradSliceEval = HipGradSliceEval;

    fn point_tape(&self, _: ()) -> HipTape {
        self.tape()
    }
    fn interval_tape(&self, _: ()) -> HipTape {
        self.tape()
    }
    fn float_slice_tape(&self, _: ()) -> HipTape {
        self.tape()
    }
    fn grad_slice_tape(&self, _: ()) -> HipTape {
        self.tape()
    }
    fn simplify(
        &self,
        trace: &VmTrace,
        storage: VmData<REGISTER_LIMIT>,
        workspace: &mut VmWorkspace<REGISTER_LIMIT>,
    ) -> Result<Self, BadTrace> {
        // host side, as fidget-jit does (fidget-jit/src/lib.rs:944-952); the fused renders simplify on the device
        self.0.simplify(trace, storage, workspace).map(HipFunction)
    }
    fn recycle(self) -> Option<VmData<REGISTER_LIMIT>> {
        self.0.recycle()
    }
    fn size(&self) -> usize {
        self.0.size()
    }
    fn vars(&self) -> &VarMap {
        self.0.vars()
    }
    fn can_simplify(&self) -> bool {
        self.0.choice_count() > 0
    }
    fn output_count(&self) -> usize {
        self.0.output_count()
    }
}

impl RenderHints for HipFunction {
    /// fan-out 4^3 = 64 children: one wavefront per parent tile
    fn tile_sizes_3d() -> TileSizes {
        TileSizes::new(&[128, 32, 8]).unwrap()
    }
    /// as fidget-jit (fan-out 8^2 = 64)
    fn tile_sizes_2d() -> TileSizes {
        TileSizes::new(&[128, 16]).unwrap()
    }
    fn simplify_tree_during_meshing(_: usize) -> bool {
        true
    }
}

impl MathFunction for HipFunction {
    fn new(ctx: &Context, nodes: &[Node]) -> Result<Self, BadNode> {
        GenericVmFunction::new(ctx, nodes).map(HipFunction)
    }
}

impl From<GenericVmFunction<REGISTER_LIMIT>> for HipFunction {
    fn from(v: GenericVmFunction<REGISTER_LIMIT>) -> Self {
        Self(v)
    }
}

/// A [`Shape`] which uses the MI355X evaluators
pub type HipShape = Shape<HipFunction>;

////////////////////////////////////////////////////////////////////////////////

/// Tracing evaluators: `Interval` is `repr(C) {lower, upper}` (types/interval.rs:12-15) and `Choice` a `u8` with Left = 1,
/// Right = 2, Both = 3 (vm/choice.rs:15-29): the buffers cross the ABI without conversion.
struct HipTracingEval<T> {
    choices: VmTrace,
    out: Vec<T>,
}
impl<T> Default for HipTracingEval<T> {
    fn default() -> Self {
        Self { choices: VmTrace::default(), out: Vec::default() }
    }
}
impl<T: From<f32> + Clone> HipTracingEval<T> {
    fn eval(
        &mut self,
        tape: &HipTape,
        vars: &[T],
        call: unsafe extern "C" fn(*mut ffi::fhip_ctx, *const ffi::fhip_tape, *const f32, u32, u32, *mut f32, *mut u8, *mut u8) -> ffi::fhip_status,
    ) -> (&[T], Option<&VmTrace>) {
        let mut simplify = 0u8;
        self.choices.resize(tape.choice_count, Choice::Unknown);
        self.choices.fill(Choice::Unknown);
        self.out.resize(tape.output_count, f32::NAN.into());
        self.out.fill(f32::NAN.into());
        CTX.with(|c| check(unsafe {
            call(c.raw(), tape.raw(), vars.as_ptr() as *const f32, vars.len() as u32, 1,
                 self.out.as_mut_ptr() as *mut f32, self.choices.as_mut_ptr() as *mut u8, &mut simplify)
        }));
        (&self.out, if simplify != 0 { Some(&self.choices) } else { None })
    }
}

/// Tracing evaluator for interval values (`fhip_interval_eval`)
#[derive(Default)]
pub struct HipIntervalEval(HipTracingEval<Interval>);
impl TracingEvaluator for HipIntervalEval {
    type Data = Interval;
    type Tape = HipTape;
    type Trace = VmTrace;
    type TapeStorage = ();
    fn eval(&mut self, tape: &HipTape, vars: &[Interval]) -> Result<(&[Interval], Option<&VmTrace>), TracingEvalError> {
        tape.vars().check_tracing_arguments(vars)?;
        Ok(self.0.eval(tape, vars, ffi::fhip_interval_eval))
    }
}

/// Tracing evaluator for point values (`fhip_point_eval`)
#[derive(Default)]
pub struct HipPointEval(HipTracingEval<f32>);
impl TracingEvaluator for HipPointEval {
    type Data = f32;
    type Tape = HipTape;
    type Trace = VmTrace;
    type TapeStorage = ();
    fn eval(&mut self, tape: &HipTape, vars: &[f32]) -> Result<(&[f32], Option<&VmTrace>), TracingEvalError> {
        tape.vars().check_tracing_arguments(vars)?;
        Ok(self.0.eval(tape, vars, ffi::fhip_point_eval))
    }
}

////////////////////////////////////////////////////////////////////////////////

/// Bulk evaluators: `Grad` is `repr(C) {v, dx, dy, dz}` (types/grad.rs:4-13)
struct HipBulkEval<T> {
    input_ptrs: Vec<*const f32>,
    output_ptrs: Vec<*mut f32>,
    lens: Vec<u32>,
    out: Vec<Vec<T>>,
}
// SAFETY: the pointers are transient and only scoped to a single evaluation
unsafe impl<T> Sync for HipBulkEval<T> {}
unsafe impl<T> Send for HipBulkEval<T> {}
impl<T> Default for HipBulkEval<T> {
    fn default() -> Self {
        Self { input_ptrs: vec![], output_ptrs: vec![], lens: vec![], out: vec![] }
    }
}
impl<T: From<f32> + Copy> HipBulkEval<T> {
    fn eval<V: Deref<Target = [T]>>(
        &mut self,
        tape: &HipTape,
        vars: &[V],
        call: unsafe extern "C" fn(*mut ffi::fhip_ctx, *const ffi::fhip_tape, *const *const f32, *const u32, u32, *const *mut f32) -> ffi::fhip_status,
    ) -> BulkOutput<'_, T> {
        let n = vars.first().map(|v| v.deref().len()).unwrap_or(0);
        self.out.resize_with(tape.output_count, Vec::new);
        for o in &mut self.out {
            o.resize(n, f32::NAN.into());
            o.fill(f32::NAN.into());
        }
        self.input_ptrs.clear();
        self.input_ptrs.extend(vars.iter().map(|v| v.as_ptr() as *const f32));
        self.lens.clear();
        self.lens.extend(vars.iter().map(|v| v.len() as u32));
        self.output_ptrs.clear();
        self.output_ptrs.extend(self.out.iter_mut().map(|v| v.as_mut_ptr() as *mut f32));
        if n > 0 {
            CTX.with(|c| check(unsafe {
                call(c.raw(), tape.raw(), self.input_ptrs.as_ptr(), self.lens.as_ptr(), self.input_ptrs.len() as u32, self.output_ptrs.as_ptr())
            }));
        }
        BulkOutput::new(&self.out, n)
    }
}

/// Bulk evaluator for arrays of points, yielding point values (`fhip_float_eval`)
#[derive(Default)]
pub struct HipFloatSliceEval(HipBulkEval<f32>);
impl BulkEvaluator for HipFloatSliceEval {
    type Data = f32;
    type Tape = HipTape;
    type TapeStorage = ();
    fn eval<V: Deref<Target = [f32]>>(&mut self, tape: &HipTape, vars: &[V]) -> Result<BulkOutput<'_, f32>, BulkEvalError> {
        tape.vars().check_bulk_arguments(vars)?;
        Ok(self.0.eval(tape, vars, ffi::fhip_float_eval))
    }
}

/// Bulk evaluator for arrays of points, yielding gradient values (`fhip_grad_eval`)
#[derive(Default)]
pub struct HipGradSliceEval(HipBulkEval<Grad>);
impl BulkEvaluator for HipGradSliceEval {
    type Data = Grad;
    type Tape = HipTape;
    type TapeStorage = ();
    fn eval<V: Deref<Target = [Grad]>>(&mut self, tape: &HipTape, vars: &[V]) -> Result<BulkOutput<'_, Grad>, BulkEvalError> {
        tape.vars().check_bulk_arguments(vars)?;
        Ok(self.0.eval(tape, vars, ffi::fhip_grad_eval))
    }
}

/// Key of a `Var::V` as the C ABI takes it (`VarIndex` is a transparent `u64` without a getter)
pub(crate) fn var_key(i: fidget_core::var::VarIndex) -> u64 {
    serde_json::to_value(i).ok().and_then(|v| v.as_u64()).expect("VarIndex is a u64")
}

/// VarMap slots of X, Y, Z as the fused entry points take them (-1 = absent)
pub(crate) fn axis_slots(vars: &VarMap) -> [i32; 3] {
    use fidget_core::var::Var;
    [Var::X, Var::Y, Var::Z].map(|v| vars.get(&v).map(|i| i as i32).unwrap_or(-1))
}

////////////////////////////////////////////////////////////////////////////////

#[cfg(test)]
mod test {
    use super::*;
    // the reference's own conformance suites, attached the way fidget-jit attaches them (fidget-jit/src/lib.rs:1381-1384)
    fidget_core::grad_slice_tests!(HipFunction);
    fidget_core::interval_tests!(HipFunction);
    fidget_core::float_slice_tests!(HipFunction);
    fidget_core::point_tests!(HipFunction);
}
