/*
 * qip_hip.h — C ABI of the MI355X (gfx950) state-vector backend for RustQIP's
 * gate-application hot path.
 *
 * Everything here is plain C: pointers, sizes, integers.  No torch / HIP types
 * appear in any signature, so the library can be bound from Rust (`extern "C"`),
 * ctypes, cgo, ... unchanged.  Each entry point names the reference interface
 * (file:line under the RustQIP tree, qip 1.5.0) it replaces.
 *
 * Conventions
 *   - every function returns 0 on success and a non-zero QIP_ERR_* code on
 *     failure; the message is available from qip_hip_last_error() (thread-local).
 *     Nothing aborts or throws across the ABI.  (The reference has no error
 *     channel in the kernel — malformed ops panic, qip/src/builder.rs:517
 *     `.unwrap()`s — so a Rust shim turns non-zero into CircuitError / panic.)
 *   - amplitudes are interleaved {re, im}: bit-compatible with
 *     num_complex::Complex<f64> / Complex<f32> (#[repr(C)]).
 *   - qubit q of an n-qubit state is bit (n-1-q) of the amplitude index
 *     (qip-iterators/src/matrix_ops.rs:18,28).
 *   - a state handle is driven by one host thread; calls are asynchronous on
 *     the handle's HIP stream; download / measure* / sync synchronise.
 */
#ifndef QIP_HIP_H
#define QIP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- value types ------------------------------------------------------ */

typedef struct { double re, im; } qip_c64; /* Complex<f64>  (qip/src/types.rs:6-13) */
typedef struct { float re, im; } qip_c32;  /* Complex<f32> */

/* Element type `P` of a call.  A device-resident STATE is always complex (QIP_C64 / QIP_C32: the `qip` crate's states are
 * Complex<P>, qip/src/types.rs:6-13).  The real and integer types are the other instances of the generic `P` of
 * qip-iterators' kernel (matrix_ops.rs:98-107 `apply_op<P>`: its own tests run it on i32, matrix_ops.rs:271-374, its benches
 * on f64, qip-iterators/benches/matmul_bench.rs:19-33,163-177): they are accepted by the slice-level calls
 * qip_hip_apply_op_host, qip_hip_apply_op_row_host and qip_hip_apply_op_device, where op payloads (`dense`, `sparse_vals`)
 * and both vectors hold plain `P` values.  Integer arithmetic wraps (two's complement), as Rust's does in release builds. */
enum qip_dtype { QIP_C64 = 0, QIP_C32 = 1, QIP_F64 = 2, QIP_F32 = 3, QIP_I64 = 4, QIP_I32 = 5 };

/* MatrixOp<P> variants (qip-iterators/src/iterators/ops.rs:11-20). */
enum qip_op_kind {
  QIP_OP_MATRIX = 0,  /* Matrix(indices, 4^k row-major data)                  */
  QIP_OP_SPARSE = 1,  /* SparseMatrix(indices, rows of (col, val)) as CSR      */
  QIP_OP_SWAP = 2,    /* Swap(h, A-indices ++ B-indices), n_indices = 2h       */
  QIP_OP_CONTROL = 3  /* Control(nc, control-indices ++ op-indices, inner)     */
};

/*
 * Flat mirror of the recursive Rust enum MatrixOp<P>.
 *   indices      n_indices qubit indices.  CONTROL: the n_controls control
 *                indices first, then the inner op's indices (ops.rs:19,84-91).
 *                SWAP: the A half then the B half (ops.rs:66-79).
 *   dense        MATRIX only: 4^k entries, row-major, element type chosen by the
 *                `dtype` argument of the call (qip_c64, qip_c32, or plain P).
 *   sparse_*     SPARSE only: CSR flattening of Vec<Vec<(usize, P)>>; row r is
 *                entries [rowptr[r], rowptr[r+1]); stored order is preserved and
 *                nothing is filtered (qubit_iterators.rs:87-101).
 *   inner        CONTROL only.  inner->indices are ignored at apply time, exactly
 *                as in the reference (matrix_ops.rs:108, ops.rs:44); nested
 *                CONTROLs accumulate their controls (ops.rs:150-154).
 */
typedef struct qip_op {
  int32_t kind;
  uint32_t n_indices;
  const uint64_t* indices;
  uint32_t n_controls;
  const void* dense;
  const uint64_t* sparse_rowptr; /* 2^k + 1 entries */
  const uint64_t* sparse_cols;
  const void* sparse_vals;
  const struct qip_op* inner;
} qip_op;

/* error codes */
enum qip_status {
  QIP_OK = 0,
  QIP_ERR_INVALID = 1,     /* what make_*_op would reject, or a malformed descriptor */
  QIP_ERR_DEVICE = 2,      /* HIP runtime error (message carries hipGetErrorString) */
  QIP_ERR_NO_DEVICE = 3,   /* no gfx950 device visible: there is no CPU fallback    */
  QIP_ERR_UNSUPPORTED = 4
};

typedef struct qip_hip_state qip_hip_state; /* opaque device-resident state */

/* ---- library ----------------------------------------------------------- */

/* Message of the last failing call on this thread ("" if none). */
const char* qip_hip_last_error(void);
/* Number of visible HIP devices (0 when none; never fails). */
int qip_hip_device_count(void);
/* ABI version of this header (bumped when entry points or options are added or changed). */
int qip_hip_abi_version(void);
/* Process-wide options (every key the product build accepts; unknown keys are QIP_ERR_INVALID).
 *   "force_generic"        1: every op (the host twins below included) runs through the literal out-of-place gather kernel —
 *                          the path the parity tests hold the specialised kernels against.  Default 0.
 *   "single_via_tile"      which single ops run as a ONE-op tile sweep (whole wave rows on both global sides whatever the
 *                          target bits; the arithmetic of the dedicated VALU kernels, IEEE-equal): 0 none, 1 dense k = 2, 3 and
 *                          Swap ops with a bit inside a 1-KiB row, 2 every dense k = 2, 3, 3 (default) also uncontrolled
 *                          single-qubit gates on a position >= 6.
 *   "tile_sched"           the tile scheduler's host-side search: 1 (default) orders the gates of a segment for the fewest LDS
 *                          passes ("tile" = 1: only across gates that commute exactly) and, for "tile" = 2 from n = 24, picks
 *                          the shortest of three position-claiming plans; 0 first come, circuit order; 2 search at every size.
 *   "jit_cache_cap"        bound of the in-process cache of run-time-compiled kernels (default 512; qip_hip_jit_cache_info).
 *   "jit_disk_cache"       1 (default) / 0: keep code objects on disk (qip_hip_jit_stats2).
 *   "jit_disk_cap_mb"      bound of that directory in MiB; oldest files go first (default -1 = $QIP_HIP_CACHE_MAX_MB or 1024;
 *                          0 = unbounded).
 *   "jit_procs"            helper processes a plan's new segments are compiled in: 0 (default) automatic, 1 in this process
 *                          only, up to 64.
 *   "dist_fold_pack"       1 (default): the gather of a sharded state's remap rides in the store phase of the tile sweep
 *                          before it where it can (qip_hip_dist_stats.packs_folded); 0 always a sweep of its own.
 *   "dist_plan_cost"       1 (default): at a remap the leaving qubits are chosen by modelled cost (count-optimal set unless
 *                          keeping the gather out of the wave rows is cheaper over the rest of the circuit); 0 by count alone.
 *   "collective_timeout_s" seconds a rank of a sharded state waits for queued work that contains an exchange or an all-reduce
 *                          before the call FAILS (and the handle is unusable) instead of hanging: a peer died or issued other
 *                          calls.  Default 120; 0 waits for ever.
 * Every rank plans for itself: "dist_plan_cost", "dist_fold_pack" and the per-handle options "tile" / "dist_overlap" must agree
 * across the processes of a sharded state.  A batch that contains an exchange compares a fingerprint of its communication
 * steps and of those options across the ranks first (one 16-byte all-reduce) and is refused on all of them when they differ.
 * (The measured alternatives of earlier rounds — row layouts, unroll factors, one-pass variants — are fixed at the values that
 * won; a -DQIP_HIP_TUNING build of the library makes them options again for tools/bench_*.py.  profiles/ holds the runs.) */
int qip_hip_set_global_option(const char* key, int64_t value);

/* ---- op validation ------------------------------------------------------
 * Re-validates what the reference constructors validate
 * (qip/src/state_ops/matrix_ops.rs:12-27 make_matrix_op, :32-81
 * make_sparse_matrix_op, :84-100 make_swap_op, :103-122 make_control_op) plus
 * what would make the reference kernel panic (index >= n, duplicate index,
 * sparse column >= 2^k).  Pure host code: works without a GPU. */
int qip_hip_validate_op(uint32_t n, const qip_op* op);

/* Algorithmic bytes one application of `op` moves on an n-qubit state of the
 * given dtype: every amplitude that can change is read once and written once
 * (SURVEY.md §8(d)).  Pure host code. */
int qip_hip_op_algorithmic_bytes(int dtype, uint32_t n, const qip_op* op, double* bytes);

/* ---- inner seam: host-pointer twin of apply_op / apply_op_overwrite -----
 * Replaces qip_iterators::matrix_ops::apply_op (matrix_ops.rs:98-123,
 * accumulate != 0: out[r] += ...) and apply_op_overwrite (:127-152,
 * accumulate == 0: out[r] = ...), including the input/output window offsets
 * (:74-90: a column outside [in_off, in_off+in_len) contributes zero).
 * `in` and `out` are HOST pointers and must not alias; the call uploads,
 * runs the HIP kernels, downloads and synchronises.  It exists for parity
 * tests and small states, not for speed (see the device-resident API below). */
int qip_hip_apply_op_host(int dtype, uint32_t n, const qip_op* op,
                          const void* in, uint64_t in_len,
                          void* out, uint64_t out_len,
                          uint64_t in_off, uint64_t out_off, int accumulate);

/* The same two functions on DEVICE slices, for any `P` of enum qip_dtype: `d_in` / `d_out` are device pointers to in_len /
 * out_len elements (not aliasing), `stream` a hipStream_t (NULL: the null stream) on `device`.  Every output row is the
 * reference's literal fold (one lane per row, columns in iterator order), so the result is bit-equal to the reference's for
 * every op kind, window and `P`.  Real / integer `P`: a dense op on k <= 4 qubits (with or without controls) and Swap travel
 * in the kernel arguments — the call only launches (it can be recorded into a hipGraph) and returns without synchronising;
 * over the whole vector such an op on distinct qubits reads every input once (16-byte accesses); larger dense tables and
 * SparseMatrix rows are uploaded per call and the call synchronises `stream` before it returns.  Complex `P`: the state
 * path's literal kernel through a temporary handle, always synchronising — a complex amplitude vector that takes many ops
 * belongs in a state (qip_hip_state_wrap adopts device memory): that is where the specialised kernels are. */
int qip_hip_apply_op_device(int dtype, int device, void* stream, uint32_t n, const qip_op* op,
                            const void* d_in, uint64_t in_len, void* d_out, uint64_t out_len,
                            uint64_t in_off, uint64_t out_off, int accumulate);

/* apply_op_row (matrix_ops.rs:38-59): the single value (op . input)[output_offset + outputrow], same window rules.
 * `out_value` points at one element of `dtype`.  Host pointers; for parity tests. */
int qip_hip_apply_op_row_host(int dtype, uint32_t n, const qip_op* op, const void* in, uint64_t in_len,
                              uint64_t outputrow, uint64_t in_off, uint64_t out_off, void* out_value);

/* Host-pointer twins of measure_probs / measure_prob WITH the input_offset window every reference measurement
 * function takes (measurement_ops.rs:44-58,115-127: `input` holds amplitudes [in_off, in_off + in_len) of the 2^n
 * vector; what lies outside the window contributes nothing).  out: 2^k doubles / one double.  (On a device-resident
 * state the window is always the whole vector; a sharded state resolves its rank bits itself, qip_hip_dist_*.) */
int qip_hip_measure_probs_host(int dtype, uint32_t n, const uint64_t* indices, uint32_t k, const void* in,
                               uint64_t in_len, uint64_t in_off, double* out);
int qip_hip_measure_prob_host(int dtype, uint32_t n, uint64_t measured, const uint64_t* indices, uint32_t k,
                              const void* in, uint64_t in_len, uint64_t in_off, double* out);

/* ---- outer seam: device-resident state ----------------------------------
 * Replaces the two host Vecs `state` / `arena` that
 * LocalBuilder::calculate_state_with_init allocates and ping-pongs
 * (qip/src/builder.rs:406-407,514).  Amplitudes stay in HBM between gates. */

/* Allocate 2^n amplitudes on `device` (all zero).  A second buffer of the
 * same size is allocated lazily, only if an op needs the out-of-place path. */
int qip_hip_state_create(uint32_t n, int dtype, int device, qip_hip_state** out);
/* Wrap caller-owned device memory (e.g. a torch tensor): `amps` holds 2^n
 * amplitudes, `scratch` (may be NULL) a second buffer of the same size,
 * `stream` is the caller's hipStream_t, used as is (NULL = the HIP null / legacy
 * default stream, which is torch's default stream on ROCm), so kernels are ordered
 * with the caller's other work such as RCCL collectives.  The handle never frees
 * wrapped memory and never creates a stream of its own. */
int qip_hip_state_wrap(uint32_t n, int dtype, int device, void* amps, void* scratch,
                       void* stream, qip_hip_state** out);
int qip_hip_state_destroy(qip_hip_state* s);

/* state[i] = (i == index) ? 1 : 0      (builder.rs:406-421) */
int qip_hip_state_init_basis(qip_hip_state* s, uint64_t index);
/* Copy amplitudes [offset, offset+len) from / to host memory. */
int qip_hip_state_upload(qip_hip_state* s, const void* src, uint64_t offset, uint64_t len);
int qip_hip_state_download(qip_hip_state* s, void* dst, uint64_t offset, uint64_t len);
/* Device pointer of the current amplitude buffer (may change after apply_op
 * when the out-of-place path swapped buffers). */
int qip_hip_state_device_ptr(qip_hip_state* s, void** amps);
/* Device pointer of the second (scratch / arena) buffer, allocating it if needed, and the
 * exchange of the two buffers.  Together they let an external out-of-place step — the RCCL
 * all-to-all of the multi-GPU qubit remap — write the scratch buffer and then make it current,
 * the way the reference swaps `state` and `arena` (builder.rs:514). */
int qip_hip_state_scratch_ptr(qip_hip_state* s, void** scratch);
int qip_hip_state_swap_buffers(qip_hip_state* s);
int qip_hip_state_sync(qip_hip_state* s);

/* Any permutation of the index bits in ONE out-of-place sweep: new[j] = old[src(j)] where bit pi[d] of src(j) is
 * bit d of j (pi has n entries).  This is what a run of Swap ops composes to (SwapOpIterator,
 * qip-iterators/src/iterators/qubit_iterators.rs:208-218: every Swap is a product of bit transpositions and moves
 * amplitudes without arithmetic, so the composition is bit-identical to applying them one by one); the tile scheduler
 * uses it for runs of swaps (QFT's closing bit reversal) and the multi-GPU remap for its gather.  Uses the scratch buffer. */
int qip_hip_state_permute_bits(qip_hip_state* s, const uint32_t* pi);

/* state <- op · state.  Equivalent to apply_op_overwrite(n, op, state, arena, 0, 0)
 * followed by the buffer swap (builder.rs:499,514). */
int qip_hip_state_apply_op(qip_hip_state* s, const qip_op* op);
/* Apply `count` ops in order (one FFI crossing per circuit instead of per gate). */
int qip_hip_state_apply_ops(qip_hip_state* s, const qip_op* ops, uint64_t count);

/* A circuit recorded once and replayed as ONE hipGraph launch (launch-bound regime: for small states a
 * kernel takes less time than its launch, ~3.7 us per gate eagerly on MI355X).  The graph is captured from
 * the same kernel launches qip_hip_state_apply_ops issues, bound to the state's current buffer; `ops` must
 * stay valid for the program's lifetime.  Programs containing ops that take the out-of-place path, or whose
 * state buffer has changed since capture, transparently fall back to eager application. */
typedef struct qip_hip_program qip_hip_program;
int qip_hip_program_create(qip_hip_state* s, const qip_op* ops, uint64_t count, qip_hip_program** out);
int qip_hip_program_run(qip_hip_program* p);
/* 1 if the last run replayed a hipGraph, 0 if it applied the ops eagerly */
int qip_hip_program_is_graph(const qip_hip_program* p);
int qip_hip_program_destroy(qip_hip_program* p);

/* The schedule option "tile" would use for this circuit, without touching a device: step_of_op[i] = index of
 * the step op i belongs to; a step holding one op is an ordinary launch, a step holding several is one
 * LDS-resident sweep (or, for a run of uncontrolled Swap ops no segment can hold, one bit-permutation sweep).
 * mode bits 0-1: 1 = circuit order, 2 = with commuting reorder; bit 2 (+4): option "tile_relabel" (an uncontrolled Swap
 * op that became a label exchange gets step -1; the closing permutation is a step of its own); bit 3 (+8): keep the
 * relabelled plan even when it is not shorter.  Pure host code. */
int qip_hip_plan_tiles(int dtype, uint32_t n, const qip_op* ops, uint64_t count, int mode,
                       int64_t* step_of_op, uint64_t* n_steps);




/* Number of index bits of a tile of the LDS-resident multi-gate sweeps (low 6 bits + the free positions). */
int qip_hip_tile_bits(void);

/* Segment-specialised tile sweeps (option "tile_jit"): where the segment kernels of this process come from.  A segment that is not resident in this process is looked up on disk first
 * ($QIP_HIP_CACHE_DIR, default $XDG_CACHE_HOME/qip_hip or ~/.cache/qip_hip; "off" / "" disables; global option "jit_disk_cache"
 * 0 / 1): code objects are stored under a 128-bit hash of the compiler + flags + the embedded kernel header + the source
 * text, so a second process LOADS (~1 ms per segment) instead of compiling (~0.45 s per 11-bit segment, ~1.3 s per wide one).
 * "The compiler" is the ROCm installation the helper processes use ($ROCM_PATH or /opt/rocm: its release and the sizes of its
 * libhiprtc / libamd_comgr), not whatever copy of those libraries the host program happens to have loaded: processes with and
 * without e.g. PyTorch's bundled ROCm share one cache, and the former hand even a single new segment to a helper.
 * The segments of a plan that are new are compiled side by side in up to "jit_procs" helper PROCESSES (global option; 0 =
 * automatic: the CPUs this process may use, at most 16, divided by the ranks of a sharded state; 1 = in this process only) —
 * hiprtc serialises the compilations of one process, separate processes scale.  The helper is `qip_jitc` next to the library
 * ($QIP_HIP_JITC overrides; absent = compile in process).  Results are bit-identical whichever way a kernel arrived. */
typedef struct qip_hip_jit_counters {
  uint64_t kernels_resident_total; /* kernels made resident in this process so far (cache misses only: a segment met again costs nothing) */
  uint64_t compiled;               /* hiprtc compilations on behalf of this process (here + in helpers)             */
  uint64_t compiled_by_helpers;    /* ... of which in helper processes                                               */
  uint64_t helper_processes;       /* helper processes spawned so far                                                */
  uint64_t disk_hits, disk_stores; /* code objects loaded from / written to the disk cache                           */
  double compile_ms;               /* wall time of the compilations (helpers run side by side: wall, not CPU time)   */
  double disk_load_ms;             /* wall time spent reading code objects                                           */
  int32_t procs;                   /* helper processes a plan may use right now                                      */
  int32_t disk_cache;              /* 1 = a cache directory is in use                                                */
  uint64_t background_segments;    /* (ABI 7) segments handed to background helpers by one-shot apply_ops ("tile_auto") */
  uint64_t disk_trimmed;           /* (ABI 7) code objects removed to keep the cache under "jit_disk_cap_mb"            */
} qip_hip_jit_counters;
int qip_hip_jit_stats2(qip_hip_jit_counters* out);
/* Cache directory: NULL = back to the default rule, "" = none.  qip_hip_jit_cache_dir: the directory in use ("" = none; the
 * string is owned by the library until the calling thread's next call). */
int qip_hip_jit_set_cache_dir(const char* dir);
const char* qip_hip_jit_cache_dir(void);
/* Host only (no device needed): compile the tile-segment source in `src_path` (contraction allowed when fma != 0) and leave the
 * code object at `out_path` in the disk cache's file format.  What the helper processes run (tools/qip_jitc.c). */
int qip_hip_jit_compile_file(const char* src_path, int fma, const char* out_path);
/* The cache of those kernels is process-wide, guarded by a mutex (handles driven by different threads stay independent)
 * and bounded: beyond `cap` entries (qip_hip_set_global_option("jit_cache_cap", n), default 512) the least recently used
 * kernels are unloaded; programs recorded into a hipGraph re-record themselves when an eviction happened since.
 * Any output may be NULL. */
int qip_hip_jit_cache_info(uint64_t* resident, uint64_t* evicted, uint64_t* cap);


/* Per-handle options (unknown keys are QIP_ERR_INVALID).  Defaults in brackets.
 *   "force_generic" [0]  1: this handle's ops run through the literal gather kernel.
 *   "profile"       [0]  1: every kernel is bracketed with HIP events (qip_hip_state_profile_get); programs run eagerly.
 *   "mfma"          [1]  matrix-core kernels for dense k = 3..5 where they win, k = 6..10 always; 0: VALU register kernels
 *                        (k <= 4, the bit-faithful fold) / the literal kernel.
 *   "fuse"          [0]  K = 2..5: qip_hip_state_apply_ops merges consecutive gates into dense gates on <= K qubits, one sweep
 *                        each; results match the gate-by-gate path to the 1e-12 bar, not bit for bit.
 *   "tile"          [0]  1: apply_ops cuts the circuit into segments (1-qubit gates with any controls, dense 2- / 3-qubit
 *                        gates, bit swaps) whose exchanging bits fit index bits 0..5 plus five free higher positions, and applies
 *                        each segment in ONE sweep through an LDS-resident tile, in circuit order up to exact commutations:
 *                        IEEE-equal to the gate-by-gate path (a dense 3-qubit gate rides as the unfused register fold, i.e.
 *                        equal to its "mfma" = 0 form).  2: also hoists gates over gates they commute with (1e-12 bar).
 *   "tile_jit"      [0]  1: every segment runs as a kernel compiled at run time (hiprtc) for the segment's STRUCTURE — op codes,
 *                        positions, masks, zero / unit / real shapes are constants; matrix entries other than 0 / +-1 are kernel
 *                        data, so new angles reuse the kernel.  Bit-identical to the interpreter.  ~0.4 - 1.3 s per new
 *                        structure, compiled side by side in helper processes and kept on disk (qip_hip_jit_stats2).
 *   "tile_wide"     [0]  1 (needs "tile_jit"): segments over a 13-bit tile held in registers, seven free positions per sweep;
 *                        same helpers and gate order as the 11-bit tile (configs[1]: 18 -> 13 sweeps).
 *   "tile_relabel"  [0]  1: the scheduler keeps a logical -> physical qubit map: in-tile swaps at the end of a segment put the
 *                        soonest-needed qubits on bits 0..5, uncontrolled Swap ops become label exchanges, one permutation sweep
 *                        at the end restores the order; used when the plan gets shorter; only moves are added (bit-identical
 *                        for "tile" = 1).  2: always.  3: as 1 and the layout PERSISTS across apply_ops calls — the first call
 *                        that needs the caller's order (download, measurement, device_ptr, a program) restores it with one sweep.
 *                        A relabelled batch that fails half way leaves an order nobody can name: the handle refuses every call
 *                        that reads amplitudes until init_basis / a full upload / copy_from overwrites them.
 *   "tile_fma"      [0]  1: compiled segments of "tile" = 2 may contract products into sums (1e-12 bar; ignored for "tile" = 1).
 *   "tile_merge"    [0]  1: compiled segments of "tile" = 2 apply a run of diagonal gates as products of their factors.
 *   "tile_auto"     [1]  who pays for compilation when "tile" >= 1 and "tile_jit" = 0: a PROGRAM (n >= 22) compiles its
 *                        segments (wide tiles) at creation; apply_ops (n >= 24: below, looking the plan up costs more host
 *                        time than the compiled sweeps save) uses compiled wide sweeps only when EVERY segment of the plan
 *                        is already resident or in the disk cache, otherwise the interpreter now and the missing segments
 *                        in background helper processes for the next call or process.  0: the state's options as set.
 *   "pair_floor"    [1]  gate-by-gate apply_ops (n >= 22): a gate whose selectors sit inside a 1-KiB wave row (a sweep of the
 *                        whole vector for half / a quarter of the bytes) goes with its neighbour as ONE two-item tile sweep when
 *                        both fit a tile (IEEE-equal).  0: one launch per gate.
 */
int qip_hip_state_set_option(qip_hip_state* s, const char* key, int64_t value);

/* Per-kernel-class timing, collected when option "profile" = 1.
 * classes: see qip_hip_kernel_class_name() — enumerate them, the list grows at the end (r4: "k_sparse_ell", "k_sparse_tile";
 * r5: "tile_sweep_parts" — not a kernel: its launch count is the number of PARTS of tile sweeps that ran in slices, no time, no bytes;
 * r6: "k_dense_small").
 * Resets with *_profile_reset. */
int qip_hip_kernel_class_count(void);
const char* qip_hip_kernel_class_name(int cls);
int qip_hip_state_profile_get(qip_hip_state* s, int cls, uint64_t* launches,
                              double* total_ms, double* algorithmic_bytes);
int qip_hip_state_profile_reset(qip_hip_state* s);

/* ---- two states side by side (validation support; no reference counterpart: the reference compares host Vecs) ---------
 * copy_from: dst <- src (same n, precision and device; ordered after src's queued work, and COMPLETE when the call returns:
 * the two handles own separate streams, so the next gate queued on src must not overtake the copy's reads).
 * max_abs_diff: max_i |a_i - b_i| over the WHOLE vector and the number of amplitudes whose components are not IEEE-equal
 * (one coalesced pass over both states).  The parity checks use it to hold a state that went through a fast path against a
 * twin that went gate by gate through the literal kernel, so that a stray write anywhere in the 2^n amplitudes is seen. */
int qip_hip_state_copy_from(qip_hip_state* dst, qip_hip_state* src);
int qip_hip_state_max_abs_diff(qip_hip_state* a, qip_hip_state* b, double* max_abs, uint64_t* n_differ);
/* (ABI 5) dst[i] = state[indices[i]], i < count: amplitudes picked by an explicit index list in one small gather kernel.  The
 * sub-cubes the parity checks compare are CONTIGUOUS windows only in the caller's own index order; on a shard of a sharded
 * state (whose logical -> physical map moves with every exchange, qip_hip_dist_layout) a logical window is scattered, and
 * this is how the checker reads it at shard sizes of 2^28 and more without downloading the shard.  Synchronises. */
int qip_hip_state_download_indices(qip_hip_state* s, const uint64_t* indices, uint64_t count, void* dst);

/* ---- measurement (qip/src/state_ops/measurement_ops.rs) ----------------- */

/* Σ|amp|²  (prob_magnitude, measurement_ops.rs:11-13) */
int qip_hip_state_norm_sqr(qip_hip_state* s, double* out);
/* out[m] for m in [0, 2^k): probability of reading m from `indices`
 * (bit i of m ↔ indices[i]; measure_probs :115-127, measure_prob :44-112). */
int qip_hip_state_measure_probs(qip_hip_state* s, const uint64_t* indices, uint32_t k,
                                double* out);
int qip_hip_state_measure_prob(qip_hip_state* s, uint64_t measured, const uint64_t* indices,
                               uint32_t k, double* out);
/* soft_measure (:153-176) with the uniform sample supplied by the caller
 * (`rand_u01` in [0,1)), so the Rust side keeps using `rand`.  f64 states reproduce the reference's sample ->
 * outcome map (the chunk that crosses zero is replayed sequentially); for f32 states the chunk sums are accumulated
 * in double while the reference subtracts sequentially in f32, so only the DISTRIBUTION of outcomes matches, not
 * necessarily the exact index for a given sample. */
int qip_hip_state_soft_measure(qip_hip_state* s, const uint64_t* indices, uint32_t k,
                               double rand_u01, uint64_t* measured);
/* measure (:190-214): forced >= 0 plays MeasuredCondition.measured;
 * forced < 0 samples with rand_u01.  Collapses and renormalises in place
 * (measure_state :220-269; no-op when the probability is 0). */
int qip_hip_state_measure(qip_hip_state* s, const uint64_t* indices, uint32_t k,
                          int64_t forced, double rand_u01, uint64_t* measured, double* prob);

/* measure_state (:220-269) with the outcome and its probability supplied by the caller
 * (MeasuredCondition{measured, prob: Some(p)}, :181-186,199-203): zero what disagrees with
 * `measured`, scale the rest by 1/sqrt(prob); no-op when prob == 0.  k may be 0 (pure rescale).
 * A sharded state collapses each shard with the GLOBAL probability through this entry. */
int qip_hip_state_measure_state(qip_hip_state* s, const uint64_t* indices, uint32_t k,
                                uint64_t measured, double prob);

/* ---- the state sharded over several GPUs (SURVEY.md §8 row e) ------------------------------------------
 * The reference is single-process; its only provision for distribution is the input / output offset windows
 * of apply_op (qip-iterators/src/matrix_ops.rs:96-97) and of the measurement functions
 * (qip/src/state_ops/measurement_ops.rs:17-19).  A qip_hip_dist is the outer seam
 * (builder.rs:406-407,499,514) for a 2^n state whose index is split by its top g = log2(world) PHYSICAL bits:
 * one process per GPU, rank r holds the 2^(n-g) amplitudes whose top physical bits read r.  A host-side
 * logical -> physical bit permutation decides which qubits are "global" at any moment:
 *   - an op whose amplitude-exchanging targets are all local runs on the shard as an ordinary local op; controls
 *     and diagonal targets on rank bits are resolved per rank (skip / restrict the matrix) and never communicate;
 *   - an op with an exchanging target on a rank bit first triggers a REMAP: the g qubits whose next use is farthest
 *     (when the circuit is known: qip_hip_dist_apply_ops) are gathered into the top g local bit positions by ONE
 *     out-of-place bit-permutation sweep, then ONE all-to-all exchanges those g bits with the g rank bits
 *     (ncclGroupStart; ncclSend / ncclRecv to every peer; ncclGroupEnd: all xGMI links busy at once).
 * Every rank must issue the same calls in the same order (SPMD). */
typedef struct qip_hip_dist qip_hip_dist;

/* How amplitudes move between ranks.  NULL selects the built-in RCCL transport (librccl is loaded on first use).
 * A caller-supplied transport exists for tests (several ranks on one GPU, exchange staged through the host) and for
 * hosts that already own a communicator.  Both functions are called by every rank collectively, return 0 on success.
 *   all_to_all      `send` / `recv` are DEVICE buffers of world * chunk_bytes; chunk p of `send` goes to rank p and
 *                   lands as chunk <sender's rank> of that rank's `recv`; `stream` is the handle's hipStream_t: the
 *                   transfer must be ordered after the work already queued on it, and complete (or be ordered on it)
 *                   before the function's effects are relied on by later work on that stream.
 *   all_reduce_sum  in-place sum over ranks of `count` doubles in HOST memory. */
typedef struct qip_hip_transport {
  void* ctx;
  int (*all_to_all)(void* ctx, const void* send, void* recv, uint64_t chunk_bytes, void* stream);
  int (*all_reduce_sum)(void* ctx, double* values, uint64_t count);
} qip_hip_transport;

/* r5 (ABI 6), optional second transport entry point: the all-to-all restricted to bytes [slice_off, slice_off + slice_bytes) of
 * every chunk, ordered on `stream` (the handle's communication stream, not the shard's).  With it (the built-in RCCL transport
 * has one) and option "dist_overlap" = 2 / 4 / 8 a remap's exchange is cut into that many slices and overlapped with the tile
 * sweeps either side of it (qip_hip_dist_stats.remaps_overlapped); without it every exchange is one all_to_all call as before. */
typedef int (*qip_hip_all_to_all_slice_fn)(void* ctx, const void* send, void* recv, uint64_t chunk_bytes, uint64_t slice_off,
                                           uint64_t slice_bytes, void* stream);
/* 128 opaque bytes that identify one RCCL communicator (ncclGetUniqueId): rank 0 calls this and hands the bytes to
 * the other ranks by whatever channel the host has (the Rust side: the launcher's environment / a file / MPI). */
#define QIP_HIP_UNIQUE_ID_BYTES 128
int qip_hip_dist_unique_id(void* id_out);

/* SURVEY.md §8(b) outer seam "qip_hip_state_create(n, dtype, n_gpus, &h)": the n-qubit state over `world` ranks
 * (a power of two; world = 1 is allowed), this process being `rank` and driving `device`.  unique_id: the bytes from
 * qip_hip_dist_unique_id (ignored when `transport` is given).  All amplitudes start at zero. */
int qip_hip_dist_create(uint32_t n, int dtype, int device, int rank, int world, const void* unique_id,
                        const qip_hip_transport* transport, qip_hip_dist** out);
int qip_hip_dist_destroy(qip_hip_dist* d);
/* (ABI 6) hand a caller-supplied transport's slice entry point to the handle (same ctx as its qip_hip_transport); NULL removes it */
int qip_hip_dist_set_slice_transport(qip_hip_dist* d, qip_hip_all_to_all_slice_fn fn);

/* state[i] = (i == logical_index) ? 1 : 0 over the whole sharded vector (builder.rs:406-421) */
int qip_hip_dist_init_basis(qip_hip_dist* d, uint64_t logical_index);
/* state <- op · state (apply_op_overwrite + swap, builder.rs:499,514); remaps when it has to */
int qip_hip_dist_apply_op(qip_hip_dist* d, const qip_op* op);
/* a whole circuit: the remap choices look ahead (farthest next use), and the runs of local ops between two remaps
 * go to the shard as one qip_hip_state_apply_ops batch (so option "tile" applies to them) */
int qip_hip_dist_apply_ops(qip_hip_dist* d, const qip_op* ops, uint64_t count);
int qip_hip_dist_sync(qip_hip_dist* d);
/* Per-handle options of a sharded state: the keys of qip_hip_state_set_option are forwarded to the shard, plus
 *   "piece_bytes"   largest single ncclSend / ncclRecv of the built-in RCCL transport (default 1 GiB; QIP_ERR_UNSUPPORTED with a
 *                   caller-supplied transport).
 *   "dist_overlap"  EXPERIMENTAL, default 0: 2 / 4 / 8 cut a remap's exchange into that many slices on a second stream,
 *                   overlapped with the tile sweeps either side of it ("tile" >= 1; same amplitudes bit for bit; a third
 *                   shard-sized buffer when the gather rides in the sweep before).  Has only ever run over the host-staged test
 *                   transport: no multi-GPU machine was available. */
int qip_hip_dist_set_option(qip_hip_dist* d, const char* key, int64_t value);

/* measurement over the whole vector (measurement_ops.rs:11-13, 115-127, 190-269): local reductions + one all-reduce;
 * `measure` collapses every shard with the GLOBAL probability.  Sampling (forced < 0) is soft_measure (:153-176) over
 * the whole vector in logical index order with rank 0's rand_u01: the index where the running subtraction crosses zero
 * is found by descending the index bit by bit (one masked norm + one all-reduce per bit, ~2 sweeps in all), i.e. the
 * reference's sample -> outcome map up to the rounding of block sums. */
int qip_hip_dist_norm_sqr(qip_hip_dist* d, double* out);
int qip_hip_dist_measure_probs(qip_hip_dist* d, const uint64_t* indices, uint32_t k, double* out);
int qip_hip_dist_measure(qip_hip_dist* d, const uint64_t* indices, uint32_t k, int64_t forced, double rand_u01,
                         uint64_t* measured, double* prob);
/* (ABI 4) the sampling step alone, without the collapse: soft_measure (:153-176) of the sharded state.  How often the
 * descent and the reference's sequential scan disagree is measured, not assumed: tests/dist_worker_gpu.py sweeps 2000
 * samples per layout against the oracle (0 disagreements; a disagreement needs the sample within ~1e-16 of a boundary). */
int qip_hip_dist_soft_measure(qip_hip_dist* d, const uint64_t* indices, uint32_t k, double rand_u01, uint64_t* measured);

/* The shard's own handle (upload / download / profile of this rank's 2^(n-g) amplitudes; owned by `d`), and the
 * current layout: phys[p] = physical bit position of logical bit position p (= n-1-qubit), n entries; physical
 * positions >= n-g are rank bits.  Together they let the host scatter / gather a vector in logical order.  The layout
 * changes with every exchange and with every uncontrolled Swap (SwapOpIterator, qubit_iterators.rs:176-219, only permutes
 * index bits: on a sharded state the qubits trade entries of this map and no amplitude moves — also when world = 1: the
 * shard of a qip_hip_dist is ALWAYS to be read through qip_hip_dist_layout, never as if it were in the caller's order;
 * behaviour since ABI 4, where such a Swap used to run as a sweep). */
int qip_hip_dist_local_state(qip_hip_dist* d, qip_hip_state** shard);
int qip_hip_dist_layout(qip_hip_dist* d, uint32_t* phys);
/* Pending rank renamings: rank bit j (physical position n-g+j) reads as (bit j of the rank) XOR (bit j of *mask).  An
 * uncontrolled anti-diagonal 1-qubit gate (X, Y, ...) on a qubit that lives on a rank bit moves nothing: the ranks trade
 * names and scale their shards; the renaming is settled by local X sweeps at the next exchange.  The amplitudes rank r
 * holds are those whose rank-bit values are r ^ *mask. */
int qip_hip_dist_rank_flip(qip_hip_dist* d, uint32_t* mask);

typedef struct qip_hip_dist_stats {
  uint64_t remaps;            /* all-to-all exchanges */
  uint64_t pack_sweeps;       /* bit-permutation sweeps that gathered the outgoing qubits (0 when already on top) */
  uint64_t bytes_sent;        /* by this rank, over all remaps */
  double exchange_ms;         /* HIP-event time of the all-to-alls on the handle's stream */
  double pack_ms;
  /* (ABI 4) what the transport itself reports, read back from the communicator — NOT what the caller passed in:
   * ncclCommCount / ncclCommUserRank for the built-in RCCL transport (0 / -1 with caller-supplied callbacks).  A bench
   * line that prints rccl_ranks = N proves RCCL saw N ranks. */
  int32_t rccl_ranks, rccl_rank;
  uint64_t pieces_sent;       /* ncclSend calls issued (chunks above `piece_bytes` go in several) */
  uint64_t piece_bytes;       /* the piece size in force (option "piece_bytes", default 1 GiB) */
  /* (ABI 5) how many of `pack_sweeps` took the LDS-tiled bit-permutation sweep (k_permute_bits) because a gathered position
   * lies inside a 1-KiB row (position < 6), and how many remaps needed no sweep of their own because the gather rode in the
   * store phase of the tile sweep before them (`packs_folded`: counted in neither pack_sweeps nor pack_ms) */
  uint64_t packs_via_permute;
  uint64_t packs_folded;
  /* (ABI 6) remaps whose exchange ran in slices on the communication stream, overlapped with the sweep before them, and how many
   * of those were also overlapped with the sweep after them; slices issued in all (option "dist_overlap" per overlapped remap) */
  uint64_t remaps_overlapped, remaps_overlapped_after, slices_overlapped;
} qip_hip_dist_stats;
/* counters since the previous call (they reset; rccl_ranks / rccl_rank / piece_bytes are properties, not counters) */
int qip_hip_dist_take_stats(qip_hip_dist* d, qip_hip_dist_stats* out);




#ifdef __cplusplus
}
#endif
#endif /* QIP_HIP_H */
