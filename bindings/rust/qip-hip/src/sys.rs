//! Raw bindings: one declaration per entry point of `include/qip_hip.h` (ABI version 8: 66 entry points; `tests/test_host_ops.py` compares the two sets).
//! The host-only test hooks of `include/qip_hip_debug.h` are not part of the binding contract and are not mirrored here.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_double, c_int, c_void};

pub const QIP_C64: c_int = 0; // Complex<f64>
pub const QIP_C32: c_int = 1; // Complex<f32>
pub const QIP_F64: c_int = 2; // f64  (real / integer P: slice-level calls only)
pub const QIP_F32: c_int = 3; // f32
pub const QIP_I64: c_int = 4; // i64 (wrapping)
pub const QIP_I32: c_int = 5; // i32 (wrapping)

pub const QIP_OP_MATRIX: i32 = 0;
pub const QIP_OP_SPARSE: i32 = 1;
pub const QIP_OP_SWAP: i32 = 2;
pub const QIP_OP_CONTROL: i32 = 3;

pub const QIP_OK: c_int = 0;
pub const QIP_ERR_INVALID: c_int = 1;
pub const QIP_ERR_DEVICE: c_int = 2;
pub const QIP_ERR_NO_DEVICE: c_int = 3;
pub const QIP_ERR_UNSUPPORTED: c_int = 4;

/// `struct qip_op`: the flat image of `MatrixOp<P>` (`qip-iterators/src/iterators/ops.rs:11-20`).
#[repr(C)]
pub struct qip_op {
    pub kind: i32,
    pub n_indices: u32,
    /// Control: control indices then op indices.  Swap: A half then B half.
    pub indices: *const u64,
    pub n_controls: u32,
    /// Matrix: 4^k `Complex<P>`, row-major.
    pub dense: *const c_void,
    /// SparseMatrix as CSR: `Vec<Vec<(usize, P)>>` flattened, order preserved.
    pub sparse_rowptr: *const u64,
    pub sparse_cols: *const u64,
    pub sparse_vals: *const c_void,
    pub inner: *const qip_op,
}

#[repr(C)]
pub struct qip_hip_state {
    _private: [u8; 0],
}
#[repr(C)]
pub struct qip_hip_program {
    _private: [u8; 0],
}

extern "C" {
    pub fn qip_hip_last_error() -> *const c_char;
    pub fn qip_hip_device_count() -> c_int;
    pub fn qip_hip_abi_version() -> c_int;
    pub fn qip_hip_set_global_option(key: *const c_char, value: i64) -> c_int;

    pub fn qip_hip_validate_op(n: u32, op: *const qip_op) -> c_int;
    pub fn qip_hip_op_algorithmic_bytes(dtype: c_int, n: u32, op: *const qip_op, bytes: *mut c_double) -> c_int;

    /// Twin of `apply_op` (`accumulate = 1`, matrix_ops.rs:98-123) and `apply_op_overwrite`
    /// (`accumulate = 0`, :127-152) on host buffers, windows included.
    pub fn qip_hip_apply_op_host(
        dtype: c_int, n: u32, op: *const qip_op,
        input: *const c_void, in_len: u64, output: *mut c_void, out_len: u64,
        in_off: u64, out_off: u64, accumulate: c_int,
    ) -> c_int;

    /// The same on device slices, for any `P` of `qip_dtype` (matrix_ops.rs:98-107 is generic over `P`).
    pub fn qip_hip_apply_op_device(
        dtype: c_int, device: c_int, stream: *mut c_void, n: u32, op: *const qip_op,
        d_in: *const c_void, in_len: u64, d_out: *mut c_void, out_len: u64,
        in_off: u64, out_off: u64, accumulate: c_int,
    ) -> c_int;

    pub fn qip_hip_state_create(n: u32, dtype: c_int, device: c_int, out: *mut *mut qip_hip_state) -> c_int;
    pub fn qip_hip_state_wrap(
        n: u32, dtype: c_int, device: c_int, amps: *mut c_void, scratch: *mut c_void,
        stream: *mut c_void, out: *mut *mut qip_hip_state,
    ) -> c_int;
    pub fn qip_hip_state_destroy(s: *mut qip_hip_state) -> c_int;
    pub fn qip_hip_state_init_basis(s: *mut qip_hip_state, index: u64) -> c_int;
    pub fn qip_hip_state_upload(s: *mut qip_hip_state, src: *const c_void, offset: u64, len: u64) -> c_int;
    pub fn qip_hip_state_download(s: *mut qip_hip_state, dst: *mut c_void, offset: u64, len: u64) -> c_int;
    pub fn qip_hip_state_device_ptr(s: *mut qip_hip_state, amps: *mut *mut c_void) -> c_int;
    pub fn qip_hip_state_scratch_ptr(s: *mut qip_hip_state, scratch: *mut *mut c_void) -> c_int;
    pub fn qip_hip_state_swap_buffers(s: *mut qip_hip_state) -> c_int;
    pub fn qip_hip_state_sync(s: *mut qip_hip_state) -> c_int;
    pub fn qip_hip_state_permute_bits(s: *mut qip_hip_state, pi: *const u32) -> c_int;

    pub fn qip_hip_state_apply_op(s: *mut qip_hip_state, op: *const qip_op) -> c_int;
    pub fn qip_hip_state_apply_ops(s: *mut qip_hip_state, ops: *const qip_op, count: u64) -> c_int;

    pub fn qip_hip_program_create(
        s: *mut qip_hip_state, ops: *const qip_op, count: u64, out: *mut *mut qip_hip_program,
    ) -> c_int;
    pub fn qip_hip_program_run(p: *mut qip_hip_program) -> c_int;
    pub fn qip_hip_program_is_graph(p: *const qip_hip_program) -> c_int;
    pub fn qip_hip_program_destroy(p: *mut qip_hip_program) -> c_int;

    pub fn qip_hip_plan_tiles(
        dtype: c_int, n: u32, ops: *const qip_op, count: u64, mode: c_int,
        step_of_op: *mut i64, n_steps: *mut u64,
    ) -> c_int;

    pub fn qip_hip_state_set_option(s: *mut qip_hip_state, key: *const c_char, value: i64) -> c_int;
    pub fn qip_hip_kernel_class_count() -> c_int;
    pub fn qip_hip_kernel_class_name(cls: c_int) -> *const c_char;
    pub fn qip_hip_state_profile_get(
        s: *mut qip_hip_state, cls: c_int, launches: *mut u64, total_ms: *mut c_double,
        algorithmic_bytes: *mut c_double,
    ) -> c_int;
    pub fn qip_hip_state_profile_reset(s: *mut qip_hip_state) -> c_int;

    pub fn qip_hip_state_copy_from(dst: *mut qip_hip_state, src: *mut qip_hip_state) -> c_int;
    pub fn qip_hip_state_max_abs_diff(a: *mut qip_hip_state, b: *mut qip_hip_state, max_abs: *mut c_double, n_differ: *mut u64) -> c_int;
    pub fn qip_hip_state_download_indices(s: *mut qip_hip_state, indices: *const u64, count: u64, dst: *mut c_void) -> c_int;
    pub fn qip_hip_state_norm_sqr(s: *mut qip_hip_state, out: *mut c_double) -> c_int;
    pub fn qip_hip_state_measure_probs(
        s: *mut qip_hip_state, indices: *const u64, k: u32, out: *mut c_double,
    ) -> c_int;
    pub fn qip_hip_state_measure_prob(
        s: *mut qip_hip_state, measured: u64, indices: *const u64, k: u32, out: *mut c_double,
    ) -> c_int;
    pub fn qip_hip_state_soft_measure(
        s: *mut qip_hip_state, indices: *const u64, k: u32, rand_u01: c_double, measured: *mut u64,
    ) -> c_int;
    pub fn qip_hip_state_measure(
        s: *mut qip_hip_state, indices: *const u64, k: u32, forced: i64, rand_u01: c_double,
        measured: *mut u64, prob: *mut c_double,
    ) -> c_int;
    pub fn qip_hip_state_measure_state(
        s: *mut qip_hip_state, indices: *const u64, k: u32, measured: u64, prob: c_double,
    ) -> c_int;

    /// `apply_op_row` (matrix_ops.rs:38-59) and the windowed `measure_probs` / `measure_prob`
    /// (measurement_ops.rs:44-58,115-127: `input_offset`) on host buffers.
    pub fn qip_hip_apply_op_row_host(
        dtype: c_int, n: u32, op: *const qip_op, input: *const c_void, in_len: u64,
        outputrow: u64, in_off: u64, out_off: u64, out_value: *mut c_void,
    ) -> c_int;
    pub fn qip_hip_measure_probs_host(
        dtype: c_int, n: u32, indices: *const u64, k: u32, input: *const c_void, in_len: u64, in_off: u64,
        out: *mut c_double,
    ) -> c_int;
    pub fn qip_hip_measure_prob_host(
        dtype: c_int, n: u32, measured: u64, indices: *const u64, k: u32, input: *const c_void, in_len: u64,
        in_off: u64, out: *mut c_double,
    ) -> c_int;

    pub fn qip_hip_tile_bits() -> c_int;
    pub fn qip_hip_jit_cache_info(resident: *mut u64, evicted: *mut u64, cap: *mut u64) -> c_int;
    /// (ABI 6) where run-time-compiled segments come from: disk cache, helper processes, this process.
    pub fn qip_hip_jit_stats2(out: *mut qip_hip_jit_counters) -> c_int;
    pub fn qip_hip_jit_set_cache_dir(dir: *const c_char) -> c_int;
    pub fn qip_hip_jit_cache_dir() -> *const c_char;
    pub fn qip_hip_jit_compile_file(src_path: *const c_char, fma: c_int, out_path: *const c_char) -> c_int;

    // ---- the state sharded over several GPUs (one process per GPU) ------------------------------------------
    pub fn qip_hip_dist_unique_id(id_out: *mut c_void) -> c_int; // QIP_HIP_UNIQUE_ID_BYTES = 128
    pub fn qip_hip_dist_create(
        n: u32, dtype: c_int, device: c_int, rank: c_int, world: c_int, unique_id: *const c_void,
        transport: *const qip_hip_transport, out: *mut *mut qip_hip_dist,
    ) -> c_int;
    pub fn qip_hip_dist_destroy(d: *mut qip_hip_dist) -> c_int;
    /// (ABI 6) a caller-supplied transport's slice entry point (the exchange overlapped with the neighbouring tile sweeps)
    pub fn qip_hip_dist_set_slice_transport(d: *mut qip_hip_dist, f: qip_hip_all_to_all_slice_fn) -> c_int;
    pub fn qip_hip_dist_init_basis(d: *mut qip_hip_dist, logical_index: u64) -> c_int;
    pub fn qip_hip_dist_apply_op(d: *mut qip_hip_dist, op: *const qip_op) -> c_int;
    pub fn qip_hip_dist_apply_ops(d: *mut qip_hip_dist, ops: *const qip_op, count: u64) -> c_int;
    pub fn qip_hip_dist_sync(d: *mut qip_hip_dist) -> c_int;
    pub fn qip_hip_dist_set_option(d: *mut qip_hip_dist, key: *const c_char, value: i64) -> c_int;
    pub fn qip_hip_dist_norm_sqr(d: *mut qip_hip_dist, out: *mut c_double) -> c_int;
    pub fn qip_hip_dist_measure_probs(d: *mut qip_hip_dist, indices: *const u64, k: u32, out: *mut c_double) -> c_int;
    pub fn qip_hip_dist_measure(
        d: *mut qip_hip_dist, indices: *const u64, k: u32, forced: i64, rand_u01: c_double,
        measured: *mut u64, prob: *mut c_double,
    ) -> c_int;
    pub fn qip_hip_dist_soft_measure(d: *mut qip_hip_dist, indices: *const u64, k: u32, rand_u01: c_double, measured: *mut u64) -> c_int;
    pub fn qip_hip_dist_local_state(d: *mut qip_hip_dist, shard: *mut *mut qip_hip_state) -> c_int;
    pub fn qip_hip_dist_layout(d: *mut qip_hip_dist, phys: *mut u32) -> c_int;
    pub fn qip_hip_dist_rank_flip(d: *mut qip_hip_dist, mask: *mut u32) -> c_int;
    pub fn qip_hip_dist_take_stats(d: *mut qip_hip_dist, out: *mut qip_hip_dist_stats) -> c_int;
}

pub const QIP_HIP_UNIQUE_ID_BYTES: usize = 128;
pub type qip_hip_all_to_all_slice_fn =
    Option<unsafe extern "C" fn(*mut c_void, *const c_void, *mut c_void, u64, u64, u64, *mut c_void) -> c_int>;

/// `struct qip_hip_jit_counters` (ABI 7)
#[repr(C)]
#[derive(Default, Debug, Clone, Copy)]
pub struct qip_hip_jit_counters {
    pub kernels_resident_total: u64,
    pub compiled: u64,
    pub compiled_by_helpers: u64,
    pub helper_processes: u64,
    pub disk_hits: u64,
    pub disk_stores: u64,
    pub compile_ms: c_double,
    pub disk_load_ms: c_double,
    pub procs: i32,
    pub disk_cache: i32,
    pub background_segments: u64,
    pub disk_trimmed: u64,
}

#[repr(C)]
pub struct qip_hip_dist {
    _private: [u8; 0],
}
/// `struct qip_hip_transport`: NULL selects the built-in RCCL transport.
#[repr(C)]
pub struct qip_hip_transport {
    pub ctx: *mut c_void,
    pub all_to_all: Option<unsafe extern "C" fn(*mut c_void, *const c_void, *mut c_void, u64, *mut c_void) -> c_int>,
    pub all_reduce_sum: Option<unsafe extern "C" fn(*mut c_void, *mut c_double, u64) -> c_int>,
}
#[repr(C)]
#[derive(Default, Debug, Clone, Copy)]
pub struct qip_hip_dist_stats {
    pub remaps: u64,
    pub pack_sweeps: u64,
    pub bytes_sent: u64,
    pub exchange_ms: c_double,
    pub pack_ms: c_double,
    /// read back from the communicator (ncclCommCount / ncclCommUserRank); 0 / -1 with caller-supplied callbacks
    pub rccl_ranks: i32,
    pub rccl_rank: i32,
    pub pieces_sent: u64,
    pub piece_bytes: u64,
    /// (ABI 5) pack sweeps that took the LDS-tiled bit-permutation sweep; remaps whose gather rode in the preceding tile sweep
    pub packs_via_permute: u64,
    pub packs_folded: u64,
    /// (ABI 6) remaps whose exchange ran in slices overlapped with the sweep before / also after them; slices issued
    pub remaps_overlapped: u64,
    pub remaps_overlapped_after: u64,
    pub slices_overlapped: u64,
}
