//! The reference's own conformance suites against the MI355X back end, attached the way fidget-jit attaches them
//! (fidget-jit/src/lib.rs:1381-1384) - as an integration test, so that `cargo test -p fidget-hip --test eval` runs exactly these.
//! Needs fidget-core's `eval-tests` feature (dev-dependency) and a GPU.
use fidget_hip::HipFunction;

fidget_core::interval_tests!(HipFunction);
fidget_core::float_slice_tests!(HipFunction);
fidget_core::grad_slice_tests!(HipFunction);
fidget_core::point_tests!(HipFunction);
