//! `qip-hip` — run circuits built with `qip` on an AMD MI355X.
//!
//! The reference has no FFI seam and forbids `unsafe` in both library crates (`qip/src/lib.rs:1`,
//! `qip-iterators/src/lib.rs:1`), so the binding is this separate crate:
//!
//! * [`sys`]     raw `extern "C"` declarations, one per entry of `include/qip_hip.h`;
//! * [`op`]      `MatrixOp<Complex<f64>>` -> `struct qip_op` (borrowing the op's own buffers);
//! * [`iterators`] `apply_op` / `apply_op_overwrite` / `apply_op_row` with the reference's signatures for any element type
//!               (`Complex<f64 / f32>`, `f64`, `f32`, `i64`, `i32`): qip-iterators' kernel is generic over `P`;
//! * [`state`]   `HipState`: the device-resident amplitude vector (RAII over `qip_hip_state_*`);
//! * [`builder`] `HipBuilder<P>`: implements `CircuitBuilder` and the 12 extension traits `LocalBuilder<P>` implements
//!               (`qip/src/builder.rs:325…969`) by delegation to an inner `LocalBuilder<P>`, replacing only
//!               `calculate_state_with_init` (`builder.rs:400-519`) with device launches; `P = f64` or `f32`;
//! * [`replay`]  writer for the flat circuit-replay text format (`rustqip_amd/replay.py`), for setups
//!               where the Rust program and the GPU are not in the same process.
//!
//! Status: **uncompiled** (no Rust toolchain in the build image).  The C ABI underneath is built and tested
//! through the same entry points from Python (`tests/`) and C++ (`rustqip_amd/host/qip_hip.hpp`).
pub mod builder;
pub mod iterators;
pub mod op;
pub mod replay;
pub mod state;
pub mod sys;

pub use builder::{HipBuilder, HipMeasurementHandle, HipMeasurements, HipStochasticMeasurementHandle};
pub use iterators::HipElement;
pub use op::HipPrecision;
pub use state::{HipError, HipState};
