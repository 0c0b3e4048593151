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

enum qip_dtype { QIP_C64 = 0, QIP_C32 = 1 };

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
 *                `dtype` argument of the call (qip_c64 or qip_c32).
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
/* Process-wide options.
 *   "force_generic"    1 routes every op (including the host twin below) through the literal gather kernel; used by the
 *                      parity tests to check both the specialised kernels and the fallback against the oracle.
 *   "single_via_tile"  which single ops run as a ONE-op tile sweep (whole rows on both global sides whatever the target bits;
 *                      the arithmetic of the dedicated VALU kernels, IEEE-equal): 0 = none, 1 = dense k = 2, 3 and Swap ops with
 *                      a bit inside a 1-KiB row, 2 = every dense k = 2, 3, 3 (default) = uncontrolled single-qubit gates on a
 *                      position >= 6 as well.  "single_via_tile_f32": the same switch for Complex<f32> states (default 3).
 *   "jit_cache_cap"    bound of the run-time compiler's kernel cache (default 512, see qip_hip_jit_cache_info).
 *   "tile_sched"       the tile scheduler's host-side decisions: 1 (default) = gates inside a segment are ordered for the fewest
 *                      LDS passes (tile = 1: only across gates that commute exactly, the result stays IEEE-equal to circuit
 *                      order) and, with tile = 2 from n = 24, the five positions of a segment are claimed by what they buy
 *                      (shortest of three plans); 0 = first come, circuit order inside segments; 2 = search at every size (tests).
 *   "tile_row_split"   11 (default): every wave-level access of a tile sweep is two 512-byte halves 32 KiB apart (the tile's six
 *                      low index positions are 0..4 and 11; Complex<f64>, n >= 12), 5: one contiguous 1-KiB row (rounds 1-3).
 *                      Same results bit for bit; which address bits travel together is what sets a sweep's HBM rate
 *                      (profiles/r04_tile_rows.md).
 *   "dist_fold_pack"   1 (default): the gather of a sharded state's remap rides in the store phase of the tile sweep before
 *                      it where it can (qip_hip_dist_stats.packs_folded); 0 = always a sweep of its own.
 *   "dist_plan_cost"   1 (default): at a remap of a sharded state the leaving qubits are chosen by modelled cost — the count-optimal set
 *                      (farthest next use) unless the set that keeps the gather out of the wave rows is cheaper over the rest of the
 *                      circuit (exchange = shard / world bytes per link, free-standing gather = one copy of the shard); 0 = by count alone.
 *                      EVERY RANK plans for itself: "dist_plan_cost", "dist_fold_pack" and "tile_row_split" must have the same value in
 *                      every process of a sharded state.  A batch that contains an exchange compares a fingerprint of its communication
 *                      steps across the ranks first (one 16-byte all-reduce) and is refused on all of them when they differ.
 *   "tile_diag_runs"   1 (default, r5): the interpreter kernel of the tile sweeps (k_tile_passes) walks every run of >= 2 consecutive diagonal
 *                      gates of a pass as ONE loop over 64-byte steps (factor, lane condition, host-resolved element mask, outside condition)
 *                      instead of decoding each gate — the same products in the same order: bit-identical; QFT at n = 30 through the
 *                      interpreter 125.3 -> 85.1 ms.  0 = every gate through its own code path (rounds 1-4).
 *   "jit_disk_cache"   1 (default) / 0, "jit_procs" 0 (automatic) .. 64: see qip_hip_jit_stats2 below.
 *   "soft_measure_one_pass"  0 (default): soft_measure = chunk sums, host walk, crossing search in one chunk (two launches); 1 = one launch
 *                      whose last block does the walk and the search.  The same function of the sample; measured slower.
 *   "tile_wide_pin"    1 (default) / 0: wide segments pass their 32 amplitudes through an empty register constraint after every gate
 *                      applied under a block-uniform branch — no semantics (bit-identical results), fewer spills in branch-heavy
 *                      segments (Clifford+T: 175 -> 0 VGPR spills, a 72-gate prefix at n = 30 25.9 -> 23.9 ms; configs[1] unchanged).
 *   "tile_wide_dense3_inline"  1 (default since r5) / 0: dense 3-qubit gates of wide segments written out group by group instead of through
 *                      pass_dense3w's loop (whose run-time indexing put the lane's 32 amplitudes in a 528-byte stack object).  The same fold:
 *                      bit-identical (tests); dense-k3 Grover at n = 30 on wide tiles 109.9 -> 77.5 ms (11-bit tiles: 88.2).
 *   "sparse_tile"      1 (default): a SparseMatrix on k >= 6 qubits with <= 4 entries per row and 3..7 of its positions outside
 *                      the wave row is applied IN PLACE with its group staged in LDS (k_sparse_tile); 0 = always the out-of-place
 *                      gather (k_sparse_ell).  Same results bit for bit.
 *   "tile_row_split_f32"  5 (default) / 12: the same choice for Complex<f32> states (measured: no gain, profiles/r04_summary.md).
 *   "debug_slice_sweeps"  0 (default) / 2 / 4 / 8 (r5, measuring aid): every multi-gate tile sweep is launched in that many parts, cut at the
 *                      highest index positions its tile leaves alone — the sliced launches of the overlapped exchange ("dist_overlap") without
 *                      any exchange: what cutting a sweep costs by itself.  Same results bit for bit.
 *   tuning aids        "perm_rows" (0 / 5 / 6), "line_bits" (0..3), "tile_pad_from" (11), "tile_wave_rule" (1), "tile_remap" (0; 4 = XCD-aware
 *                      block -> tile order in run-time-compiled segments), "k4_direct" (0): measured alternatives kept switchable
 *                      (profiles/r02_*.md, r03_tile_skeleton.md). */
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

/* apply_op_row (matrix_ops.rs:38-59): the single value (op . input)[output_offset + outputrow], same window rules.
 * `out_value` points at one qip_c64 / qip_c32.  Host pointers; for parity tests. */
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
/* Host-only test hook: the descriptor of that sweep as JSON (tile bit positions, LDS swizzle), NULL on error. */
const char* qip_hip_debug_permute_plan(uint32_t n, const uint32_t* pi, uint32_t row_bits, uint32_t fold_bits);

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

/* Host-only: how one pass of a tile sweep lays the thread id over the tile.  pass_bits = the pass's three exchange
 * bits (tile-index space 0..10, ascending); *lanepos gets nibble k = tile-index bit filled by bit k of the 8-bit
 * thread id.  The tile is stored in LDS at slot(t) = t ^ ((t >> S) & (2^S - 1)), S = 4 (QIP_C64) / 5 (QIP_C32);
 * together the two make every pass free of LDS bank conflicts unless it holds both bits of a pair (j, j+S).  Exposed
 * so the claim can be checked without a GPU (tests/test_host_ops.py). */
int qip_hip_tile_lane_assignment(int dtype, const uint32_t* pass_bits, uint64_t* lanepos);

/* Host-only test hook: the complete tile plan of a circuit as a JSON string (owned by the library, valid until
 * the calling thread's next call; NULL on error): the schedule of qip_hip_plan_tiles and, for every multi-gate
 * step, the free bit positions, the passes (exchange bits, lane-bit assignment) and the gate descriptors exactly
 * as they are shipped to k_tile_passes.  tests/test_tile_plan_cpu.py replays it with a numpy model of the kernel
 * and checks the result against the CPU oracle, so the host half of the tile path is covered without a GPU. */
const char* qip_hip_debug_tile_plan(int dtype, uint32_t n, const qip_op* ops, uint64_t count, int mode);

/* Host-only test hook (r4): what the host decides about applying ONE SparseMatrix op in place through the LDS-staged tile
 * kernel (k_sparse_tile) on a state of n qubits — the tile's positions, the block-base descriptor, the row table (entries per
 * row, each stored column's place in the tile, the values) — as JSON, or {"applies":0} when the op takes another kernel.
 * tests/test_tile_plan_cpu.py replays it with a numpy model of the kernel against the CPU oracle.  NULL on error. */
const char* qip_hip_debug_sparse_tile(int dtype, uint32_t n, const qip_op* op);

/* Number of index bits of a tile of the LDS-resident multi-gate sweeps (low 6 bits + the free positions). */
int qip_hip_tile_bits(void);

/* Segment-specialised tile sweeps (option "tile_jit"): how many segment kernels this process has compiled with hiprtc
 * so far and the time that took (cache misses only; a segment met again costs nothing). */
int qip_hip_jit_stats(uint64_t* kernels_compiled, double* compile_ms);

/* r5 (ABI 6): where those kernels come from.  A segment that is not resident in this process is looked up on disk first
 * ($QIP_HIP_CACHE_DIR, default $XDG_CACHE_HOME/qip_hip or ~/.cache/qip_hip; "off" / "" disables; global option "jit_disk_cache"
 * 0 / 1): code objects are stored under a 128-bit hash of compiler version + flags + the embedded kernel header + the source
 * text, so a second process LOADS (~1 ms per segment) instead of compiling (~0.45 s per 11-bit segment, ~1.3 s per wide one).
 * The segments of a plan that are new are compiled side by side in up to "jit_procs" helper PROCESSES (global option; 0 =
 * automatic: the CPUs this process may use, at most 16, divided by the ranks of a sharded state; 1 = in this process only) —
 * hiprtc serialises the compilations of one process, separate processes scale.  The helper is `qip_jitc` next to the library
 * ($QIP_HIP_JITC overrides; absent = compile in process).  Results are bit-identical whichever way a kernel arrived. */
typedef struct qip_hip_jit_counters {
  uint64_t kernels_resident_total; /* kernels made resident in this process so far (= qip_hip_jit_stats' count)    */
  uint64_t compiled;               /* hiprtc compilations on behalf of this process (here + in helpers)             */
  uint64_t compiled_by_helpers;    /* ... of which in helper processes                                               */
  uint64_t helper_processes;       /* helper processes spawned so far                                                */
  uint64_t disk_hits, disk_stores; /* code objects loaded from / written to the disk cache                           */
  double compile_ms;               /* wall time of the compilations (helpers run side by side: wall, not CPU time)   */
  double disk_load_ms;             /* wall time spent reading code objects                                           */
  int32_t procs;                   /* helper processes a plan may use right now                                      */
  int32_t disk_cache;              /* 1 = a cache directory is in use                                                */
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

/* Host-only test hook: generate AND compile (hiprtc cross-compiles for gfx950 without a device) the run-time source
 * of every multi-gate step of the circuit's tile schedule.  *first_source (may be NULL) points at the first segment's
 * source text, owned by the library until the calling thread's next call. */
int qip_hip_debug_tile_jit(int dtype, uint32_t n, const qip_op* ops, uint64_t count, int mode, uint64_t* segments,
                           uint64_t* source_bytes, uint64_t* code_bytes, const char** first_source);

/* Options: key is one of
 *   "force_generic"  1 = route every op through the literal gather kernel
 *   "profile"        1 = bracket every kernel with HIP events (see *_profile_*)
 *   "lowbit_shuffle" 1 = cross-lane variant of the 1-qubit kernel for low bit positions (default 1)
 *   "mfma"           1 = matrix-core kernels: dense k = 3..5 where they win (f64 and f32 forms), k = 6..8 (f64,
 *                    A operand streamed through LDS); 0 = VALU register kernels (k <= 4) / the literal kernel (default 1)
 *   "fuse"           K >= 2: qip_hip_state_apply_ops merges consecutive gates into dense gates on
 *                    <= K qubits (K <= 5) and applies each in one sweep; results
 *                    then match the gate-by-gate path to rounding (1e-12 bar), not bit for bit.
 *                    0 (default) = one sweep per gate, bit-faithful to the reference's fold order.
 *   "tile"           1: qip_hip_state_apply_ops cuts the circuit into segments of gates (1-qubit gates with any
 *                    controls, dense 2-qubit gates, bit swaps) whose exchanging bits live on index bits 0..5
 *                    plus five free higher bits and applies each segment in ONE sweep
 *                    through an LDS-resident tile, in circuit order up to exact commutations of rounding-free gates
 *                    (IEEE-equal to the gate-by-gate path; a dense 3-qubit gate rides along as the unfused register fold,
 *                    i.e. equal to its gate-by-gate form under "mfma" = 0 — on the matrix cores it is an fma chain);
 *                    2: additionally hoists gates over skipped gates they commute with (1e-12 bar). 0 = off.
 *   "tile_jit"       1: every tile segment runs as a kernel compiled at run time for that segment's STRUCTURE (hiprtc; op
 *                    codes, bit positions, control masks, zero / unit / real shapes become constants of the code: no descriptor
 *                    fetch, no dispatch left), cached per process by its source.  The segment's NUMBERS — every matrix component
 *                    that is not exactly 0 or +-1 — are kernel data (scalar loads from the arena), so new rotation angles reuse
 *                    the compiled kernel: a variational loop compiles once.  Bit-identical to the interpreter (same helpers,
 *                    same order).  Pays ~0.3-1 s per NEW structure: for circuits that are replayed (programs, loops), not for
 *                    one-shot runs.  2: same as 1 (kept for round-3 callers).  3 (tuning aid): numbers as literals in the
 *                    source (every new angle is a new kernel).  0 (default) = the interpreter kernel.
 *   "tile_relabel"   1: the tile scheduler keeps a logical -> physical map of the qubits: at the end of every segment
 *                    in-tile bit swaps (riding along in the same sweep) put the qubits whose next amplitude-exchanging use
 *                    comes soonest on index bits 0..5, so the next segment spends its five free positions on five OTHER
 *                    qubits; uncontrolled Swap ops become label exchanges (no sweep at all); one bit-permutation sweep at
 *                    the end restores the order.  Only moves are added and no gate changes its place in the plain schedule's
 *                    order: bit-identical to "tile" = 1 without it (for "tile" = 2 the hoists depend on which gates share a
 *                    physical tile, so the two plans agree to the 1e-12 bar of that mode only; a dense 3-qubit gate that rides
 *                    in a segment in one plan and runs alone on the matrix cores in the other likewise differs by rounding).
 *                    The plan is only used when it is shorter than the plain
 *                    one (random circuits: 19 -> 14 sweeps for configs[1]; layered ones like QFT / Grover keep the plain
 *                    plan).  2 = use it unconditionally (tests).  3 = as 1, and the layout PERSISTS across apply_ops calls: a
 *                    batch starts from the layout the previous one left and does not pay the restoring sweep; the caller's
 *                    order is restored (one sweep) by the first call that needs it — download / upload / measurement /
 *                    device_ptr / a batch without relabelling / a program capture.  A circuit applied in chunks (a
 *                    variational loop, a host that streams its ops) then costs what it costs in one piece.
 *                    0 (default) = off.  Needs the scratch buffer (a state too large for it keeps the plain plan).  A relabelled
 *                    batch that fails half way (a launch, an allocation, the run-time compiler) leaves the buffer in an order
 *                    nobody can name: the handle then refuses every call that reads or computes from the amplitudes, with the
 *                    original message, until qip_hip_state_init_basis / a full upload / copy_from overwrites them (ABI 5).
 *   "tile_fma"       1: run-time-compiled segments of "tile" = 2 are compiled with multiply-add contraction (v_fma_f64: a complex
 *                    product is 4 instead of 6 vector instructions; QFT at n = 30: 57 -> 50 ms).  Ignored for "tile" = 1, which
 *                    promises IEEE equality with the gate-by-gate path.  0 (default) = off.
 *   "tile_merge"     1: in run-time-compiled segments of "tile" = 2, a run of consecutive diagonal gates (they all commute) is
 *                    applied as PRODUCTS: each gate's factor joins the running product of the set of a lane's elements it acts on,
 *                    each element then takes the product of its sets (QFT: ~33 complex products per lane after each H instead of
 *                    116).  Rounding differs from the sequential products (1e-12 bar).  Ignored for "tile" = 1.  0 (default) = off.
 *   "tile_wide"      1 (ABI 5; needs "tile_jit"): the segments run over a WIDE tile — 2^13 amplitudes per block held in registers
 *                    (32 per lane = five register bits; 256 lanes), LDS only as a transposition buffer — so a segment claims
 *                    SEVEN free positions instead of five and the circuit needs fewer sweeps (configs[1] in circuit order:
 *                    18 -> 13, relabelled 14 -> 10; a light sweep costs the same 5.2 - 5.8 ms).  Gates on the five register
 *                    bits of the moment cost no LDS traffic; a transposition (four quarters through the buffer) brings in up to
 *                    three new register bits.  Same helpers, same gate order: "tile" = 1 stays IEEE-equal to the gate-by-gate
 *                    path.  "tile_fma" / "tile_merge" apply to wide segments of "tile" = 2 as they do to narrow ones.
 *                    0 (default) = the 11-bit LDS-resident tile.  r5: also inside hipGraph programs.
 *   "tile_auto"      1 (default, ABI 6): who compiles.  apply_ops on a state with "tile" >= 1 and "tile_jit" = 0 runs the interpreter
 *                    kernel (a circuit that runs once does not repay seconds of compilation); a PROGRAM (qip_hip_program_create)
 *                    created on such a state with n >= 22 is made to be replayed, so its own launches use run-time-compiled
 *                    segments over wide tiles ("tile_wide"), compiled once at creation
 *                    (helper processes + disk cache, qip_hip_jit_stats2).  Same helpers, same order of operations: bit-identical
 *                    to the interpreter for "tile" = 1.  0 = programs use the state's options as they are.
 *   "pair_floor"     1 (default, ABI 6): in the gate-by-gate path of apply_ops (n >= 22), a gate whose selectors — controls, the target of
 *                    a phase-type diagonal — sit inside a 1-KiB wave row costs a sweep of the WHOLE vector for half / a quarter of the
 *                    algorithmic bytes (the memory system moves whole lines: T on bit 0 41 %, CNOT with the control in a row 40 %); when it
 *                    and the NEXT gate fit one tile, the two go as ONE two-item tile sweep (interpreter kernel, circuit order, the same
 *                    unfused arithmetic: IEEE-equal) and the neighbour rides for free.  0 = always one launch per gate.
 *   "swap_single"    1 = one sweep per transposition of a Swap (tuning aid; default: groups of transpositions per sweep)
 *   "tile_passes"    1 (default): tile sweeps keep each lane's 8-element group in registers across a pass of
 *                    gates (one LDS round trip per pass); 0: one LDS round trip per gate (tuning aid)
 *   "packed_f32"     1 (default): f32 states are swept as 16-B elements of two amplitudes where possible
 *   "unroll"         1 = one item per iteration in the matrix-core kernel (tuning aid)
 */
int qip_hip_state_set_option(qip_hip_state* s, const char* key, int64_t value);

/* Per-kernel-class timing, collected when option "profile" = 1.
 * classes: see qip_hip_kernel_class_name() — enumerate them, the list grows at the end (r4: "k_sparse_ell", "k_sparse_tile";
 * r5: "tile_sweep_parts" — not a kernel: its launch count is the number of PARTS of tile sweeps that ran in slices, no time, no bytes).
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
/* options: "tile", "fuse", "mfma", "profile", ... are forwarded to the shard (qip_hip_state_set_option); "piece_bytes"
 * (largest single ncclSend / ncclRecv) belongs to the built-in RCCL transport: QIP_ERR_UNSUPPORTED with a caller-supplied one;
 * "dist_overlap" (r5): 0 (default) / 1 = every exchange is one all-to-all on the shard's stream; 2 / 4 / 8 = a remap whose
 * neighbouring local batches run as tile sweeps ("tile" >= 1) has its exchange cut into that many slices (the index positions right
 * below the chunk-selecting ones), issued on a separate stream as soon as the LAST sweep before the remap — launched in as many
 * parts — has stored them, and the FIRST sweep after it starts on each slice as soon as it has landed.  Same amplitudes bit for
 * bit.  Costs a third shard-sized buffer when the remap's gather rides in that last sweep.  Default off for the built-in RCCL
 * transport: no multi-GPU machine has been available to run it on. */
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

/* Host-only: how one rank's all-to-all of `chunk_bytes` per peer is cut into sends of at most `piece_bytes` — the list the
 * built-in RCCL transport walks inside ONE ncclGroupStart / ncclGroupEnd (peer, byte offset inside the chunk, length; the
 * matching receive has the same three numbers).  Returns the number of pieces; fills at most `cap` entries of each array
 * (any may be NULL).  Test transports use the same list, so the loop is exercised without a second GPU. */
int64_t qip_hip_dist_debug_pieces(int rank, int world, uint64_t chunk_bytes, uint64_t piece_bytes, uint64_t cap,
                                  int32_t* peer, uint64_t* offset, uint64_t* length);

/* Host-only (r5): which of that plan's remaps the overlapped exchange (option "dist_overlap" = `slices`) serves when the local
 * batches run as tile sweeps in scheduler mode `tile_mode` (1 / 2 = "tile", + 16 = wide tiles): JSON
 * {"remaps":[{"pack":0|1,"before":0|1,"after":0|1,"batch_sweeps_before":k}, ...]} — "before": the batch's last sweep is cut into
 * slices and the exchange starts beside it, "after": the next batch's first sweep awaits the slices one by one.  The predicate
 * the executor itself applies; tools/model_scaling.py prices the overlap with it.  NULL on error. */
const char* qip_hip_dist_debug_overlap(uint32_t n, int dtype, int rank, int world, const qip_op* ops, uint64_t count, int tile_mode,
                                       int slices);

/* Host-only test hook: what rank `rank` of `world` would do for this circuit on a fresh state, as a JSON string
 * (owned by the library, valid until the calling thread's next call; NULL on error): the steps
 *   {"t":"local","op":{...}}   the op this rank applies to its shard, in LOCAL qubit indices
 *   {"t":"pack","sel":[...]}   gather these local bit positions into the top g positions (in this order)
 *   {"t":"exchange"}           all-to-all of the top g local bits with the g rank bits
 * and the final layout.  tests/test_distributed_cpu.py replays it with the CPU oracle as the shard and gloo as the
 * transport, so the planner and the per-rank localisation are covered without a GPU. */
const char* qip_hip_dist_debug_plan(uint32_t n, int dtype, int rank, int world, const qip_op* ops, uint64_t count);

#ifdef __cplusplus
}
#endif
#endif /* QIP_HIP_H */
