// qip_internal.h — what the translation units of libqip_hip.so share (not part of the C ABI).
//   qip_core.hip      errors, options, op flattening / validation, kernel choice (make_plan), state handles, profiling
//   qip_launch.hip    one launcher per kernel class, apply_op
//   qip_tile_sched.hip the host-only tile scheduler (segments, passes, relabelling, plan export)
//   qip_circuit.hip   apply_ops: fusion, tile sweeps (interpreter + run-time-compiled segments), hipGraph programs
//   qip_host.hip      host-pointer twins of the reference functions
//   qip_measure.hip   measurement
//   qip_dist.hip      the sharded state (planner, pack sweep, exchange)
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <complex>
#include <cstdarg>
#include <functional>
#include <cstdio>
#include <cstring>
#include <deque>
#include <dlfcn.h>
#include <map>
#include <string>
#include <exception>
#include <new>
#include <type_traits>
#include <utility>
#include <vector>

#include "../../include/qip_hip.h"
#include "../../include/qip_hip_debug.h"
#include "qip_kernels.h"

using namespace qipk;

extern thread_local std::string g_last_error;
int fail(int code, const char* fmt, ...);

#define HIPCHK(expr)                                                                         \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      return fail(QIP_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),     \
                  __FILE__, __LINE__);                                                       \
  } while (0)

#define QCHK(expr)             \
  do {                         \
    int rc_ = (expr);          \
    if (rc_ != QIP_OK) return rc_; \
  } while (0)

// No C++ exception may cross the C ABI (include/qip_hip.h: "never aborts or throws"): every entry point that can
// allocate is a function-try-block ending in this handler, which turns std::bad_alloc / std::length_error / ...
// into a status + message like any other failure.
#define QIP_CATCH_ALL                                                                       \
  catch (const std::bad_alloc&) { return fail(QIP_ERR_DEVICE, "out of host memory"); }      \
  catch (const std::exception& e) { return fail(QIP_ERR_INVALID, "internal error: %s", e.what()); } \
  catch (...) { return fail(QIP_ERR_INVALID, "internal error: unknown C++ exception"); }

extern int64_t g_force_generic;
extern uint32_t g_line_bits;
extern int64_t g_tile_pad_from, g_tile_wave_rule, g_tile_remap, g_tile_sched;
extern int64_t g_single_via_tile, g_single_via_tile_f32;
extern int64_t g_dist_fold_pack, g_dist_plan_cost;  // qip_dist.hip
extern int64_t g_collective_timeout_s;             // qip_core.hip: seconds a rank waits for an exchange before it fails (0 = for ever)
extern int64_t g_sparse_tile;  // qip_launch.hip
extern int64_t g_soft_measure_one_pass;  // qip_measure.hip
extern int64_t g_tile_wide_pin, g_tile_wide_dense3_inline;  // qip_circuit.hip
extern int64_t g_jit_disk, g_jit_procs, g_jit_world;  // qip_circuit.hip: code objects on disk, helper processes, ranks sharing the host
extern int64_t g_debug_slice_sweeps;  // qip_circuit.hip
extern int64_t g_jit_threads;     // qip_circuit.hip: host threads that compile a plan's new segments side by side
extern int64_t g_force_k4_direct;  // tuning aid: dense k = 4 on the matrix cores reads its operands straight from HBM (k_gate_kq_mfma)  // 0 = never, 1 = single dense k = 2, 3 / Swap ops with a bit inside a row go as a one-item tile sweep, 2 = every dense k = 2, 3  // tuning aids of the tile sweeps (qip_hip_set_global_option)

struct FlatOp {
  const qip_op* outer = nullptr;
  const qip_op* inner = nullptr;  // innermost non-Control op
  uint32_t k_all = 0;             // outer->n_indices
  uint32_t n_control = 0;         // flattened (ops.rs:150-154)
  uint32_t n_op = 0;              // indices the inner iterator is built with
  bool distinct = true;           // all outer indices distinct
};
int flatten_op(uint32_t n, const qip_op* op, bool strict, FlatOp* f);

enum KernelClass {
  KC_GATE1Q_PAIR = 0,
  KC_GATE1Q_XLANE,
  KC_PHASE,
  KC_DIAG,
  KC_DIAG1Q,
  KC_SWAP_BITS,
  KC_GATE_KQ,
  KC_GATE_KQ_MFMA,
  KC_TILE_GATES,
  KC_GATHER_GENERIC,
  KC_NOOP,
  KC_SPARSE_KQ,
  KC_GATE_KQ_BIG,
  KC_PERMUTE,
  KC_SPARSE_ELL,
  KC_SPARSE_TILE,
  KC_TILE_PARTS,  // (r5) not a kernel: `launches` counts the PARTS of tile sweeps that were launched in slices (TileSlicing); no time, no bytes
  KC_DENSE_SMALL, // (r6) dense k = 5..10 on a state with fewer than 16 groups (k_dense_small)
  KC_COUNT
};

template <typename T> struct HostAmp { T re, im; };  // host view of one downloaded amplitude

struct Plan {
  int cls = KC_GATHER_GENERIC;
  double alg_bytes = 0;  // algorithmic bytes
  // shared
  std::vector<uint32_t> cpos;  // control bit positions
  std::vector<uint32_t> opos;  // op bit positions, opos[0] = MSB of the sub-index
  // 1q
  double m[8] = {0};  // 2x2 as re,im pairs (converted to T at launch)
  uint32_t nz = 0;
  // phase
  uint64_t phase_ones = 0;  // among opos: bits that must be 1 (others 0)
  double phase[2] = {0};
  // diag / kq: host copy of matrix data to ship to the device arena (as doubles re,im)
  std::vector<double> table;
};

static inline bool is_zero2(double re, double im) { return re == 0.0 && im == 0.0; }
static inline bool is_one2(double re, double im) { return re == 1.0 && im == 0.0; }
static constexpr uint32_t kMaxRegK = 4;     // dense gates held in registers (VALU form)
static constexpr uint32_t kMaxMfmaK = 5;    // dense gates on the f64 matrix cores: k = 3..5 (A operand in registers)
static constexpr uint32_t kMaxBigK = 8;     // ... k = 6..8 with the A operand streamed through LDS (k_gate_big_mfma)
static constexpr uint32_t kMaxHugeK = 10;   // ... k = 9, 10 with X in LDS and the A operand streamed from L2 (k_gate_huge_mfma)
static constexpr uint32_t kMaxSparseK = 5;  // SparseMatrix ops applied in place (one 2^k group per lane, staged in LDS)
static constexpr uint32_t kMaxDiagK = 12;   // largest Matrix op inspected for structure (4^k entries are read)
int make_plan(int dtype, uint32_t n, const FlatOp& f, bool force_generic, Plan* p);
struct qip_hip_state;
bool sparse_tile_applies(const qip_hip_state* s, const Plan& p, const FlatOp& f);  // qip_launch.hip: the op would run in place through k_sparse_tile

struct ProfRec {
  int cls;
  hipEvent_t e0, e1;
  double bytes;
};

struct qip_hip_state {
  uint32_t n = 0;
  int dtype = QIP_C64;
  int device = 0;
  uint64_t namps = 0;
  size_t amp_bytes = 16;
  void* cur = nullptr;
  void* alt = nullptr;
  bool owns_cur = false, owns_alt = false;
  hipStream_t stream = nullptr;
  bool owns_stream = false;
  // device arena for op payloads (matrices, CSR)
  void* arena = nullptr;
  size_t arena_cap = 0;
  uint64_t arena_gen = 0;  // bumped whenever the arena is re-allocated: captured graphs hold its address
  // reduction scratch
  double* d_partial = nullptr;
  void* d_ticket = nullptr;  // soft_measure in one launch: ticket counter + result words (qip_measure.hip)
  size_t partial_cap = 0;
  // options
  int64_t force_generic = 0;
  int64_t profile = 0;
  int64_t lowbit_shuffle = 1;
  int64_t mfma = 1;
  int64_t fuse = 0;
  int64_t tile_passes = 1;  // tile sweeps: group gates into register passes (k_tile_passes) vs one LDS pass per gate
  int64_t tile = 0;  // 0 off, 1 = LDS-resident multi-gate sweeps in circuit order, 2 = with commuting reorder
  int64_t packed_f32 = 1;
  int64_t tile_relabel = 0;  // tile sweeps: the scheduler relabels the qubits (schedule_tiles_relabel); 3 = the layout persists
  // Persistent relabelling (tile_relabel = 3): layout[p] = physical position of logical index bit p, in force BETWEEN
  // apply_ops calls (empty = the caller's order).  Everything that reads, writes or addresses amplitudes other than a
  // relabelling apply_ops first restores the caller's order with one bit-permutation sweep (state_settle, part of STATE_ENTER).
  std::vector<uint32_t> layout;
  // A relabelling plan that failed half way left the buffer in an order nobody can name: every call that reads or computes
  // from the amplitudes fails with the original message until init_basis / a full upload / copy_from overwrites them.
  bool poisoned = false;
  std::string poison_msg;
  // The sharded state's remap (qip_dist.hip) asks the batch it hands to this shard to leave its result PACKED in the second
  // buffer (TileStorePerm: the leaving qubits' positions gathered on top): the last sweep of the batch — if it is a tile
  // sweep — stores its tiles there and the remap needs no gather sweep of its own.  `fold_now` is raised while the batch's
  // last step runs, `fold_done` reports that a sweep took the request.
  const TileStorePerm* fold_request = nullptr;
  bool fold_now = false, fold_done = false;
  // r5: the FIRST / LAST tile sweep of a batch launched in 2^nbits parts, part k = the blocks whose base index reads k at `pos`
  // (amplitude-index positions that are not tile positions of that sweep) — the sharded state's exchange is cut into the same
  // slices and overlaps with these sweeps (qip_dist.hip): `after(k)` runs once part k is enqueued (the slice can be sent as soon
  // as it is stored), `before(k)` before part k is enqueued (it may start as soon as the slice has landed).  `fallback` runs
  // instead when the step turns out not to be sliceable.  See TileSlicing.
  struct TileSlicing* slice_first = nullptr;
  struct TileSlicing* slice_last = nullptr;
  struct TileSlicing* slice_now = nullptr;
  int64_t swap_single = 0;  // 1 = one sweep per transposition (tuning aid; default groups them, k_swapn)
  int64_t tile_jit = 0;     // 1 (= 2) = tile segments run as kernels compiled at run time for that segment's STRUCTURE (hiprtc,
                            // cached), its numbers are kernel data (angles can change without recompiling); 3 = numbers as literals
  int64_t tile_fma = 0;     // run-time-compiled segments of tile = 2: products may fuse into sums (1e-12 bar, not IEEE equality)
  int64_t tile_merge = 0;   // ... and runs of diagonal gates are applied as products of their factors
  int64_t tile_wide = 0;    // r4: run-time-compiled segments over a 13-bit register-resident tile (seven free positions per sweep)
  int64_t tile_auto = 1;    // r5: programs (qip_hip_program_create) on a state with tile >= 1, tile_jit = 0 and n >= 22 compile their segments
                            // (wide ones unless the circuit holds dense 3-qubit gates); apply_ops keeps the interpreter.  0 = never
  int64_t pair_floor = 1;   // r5: gate-by-gate apply_ops pairs a gate whose selectors sit inside a wave row (a full sweep for half the bytes)
                            // with its neighbour into one two-item tile sweep when both fit a tile (IEEE-equal); 0 = one launch per gate, always
  int num_cus = 256;        // compute units of the device
  bool jit_prepare = false; // compile the segments' kernels but launch nothing (before a graph capture; the parallel pre-compilation)
  bool jit_for_capture = false;  // ... on behalf of a graph capture: the plan must be the one the capture will record
  bool jit_lookup_only = false;  // r6 (option tile_auto, one-shot callers): the plan's segments are only LOOKED UP (memory, disk cache); a miss
                                 // hands them to background helpers and the batch returns kJitMiss before anything has run
  // r4: apply_ops collects the sources of a plan's segments that are not in the kernel cache yet (source text, contraction flag)
  // and compiles them on several host threads before the first launch (hiprtc: ~0.35 s per 11-bit segment, ~1.4 s per wide one)
  std::vector<std::pair<std::string, bool>>* jit_collect = nullptr;
  // program capture (hipGraph): non-null while a program records its launches.  Op payloads then go to the PROGRAM's own
  // device pool instead of the arena (ProgPool below): the graph holds kernel nodes only and replays nothing from the host.
  struct ProgPool* capture_pool = nullptr;
  std::vector<struct qip_hip_program*> programs;  // graphs recorded against this state's buffers  // f32: sweep two amplitudes per 16-B element when bit 0 is not involved  // 0 = gate by gate; K >= 2 = fuse into dense gates on <= K qubits
  int64_t unroll = 0;  // 0 = default per kernel
  // profiling
  std::vector<ProfRec> pending;
  std::vector<hipEvent_t> free_events;
  uint64_t prof_launches[KC_COUNT] = {0};
  double prof_ms[KC_COUNT] = {0};
  double prof_bytes[KC_COUNT] = {0};
};

// Device memory a recorded program owns for the payloads of its ops (dense tables, matrix-core fragments, CSR / ELL rows, tile
// descriptors, the numbers of run-time-compiled segments): packed ONCE while the program is recorded, one region per launch
// group, uploaded in one copy when the recording has succeeded.  While a program records (qip_hip_state::capture_pool),
// ensure_arena hands out regions of this pool and arena_upload writes the host image instead of enqueueing a copy.
struct ProgPool {
  void* base = nullptr;
  size_t cap = 0, used = 0;
  bool overflow = false;        // a sizing pass: `used` keeps counting, nothing is stored, the recording is thrown away
  std::vector<char> image;      // host image of [0, used)
};
struct TileSlicing {
  uint32_t nbits = 0;
  uint32_t pos[3] = {0, 0, 0};  // slice k has bit j of k at amplitude-index position pos[j] (the state's order when the step runs)
  bool need_fold = false;       // only valid together with a packed store (the slices are slices of the PACKED buffer)
  bool in_place_only = false;   // never together with a packed store (the second buffer may still be read by the exchange)
  std::function<int(uint32_t, bool)> before, after;  // (k, folding)
  std::function<int()> fallback;                     // the step ran unsliced
  uint32_t parts_done = 0;
  bool folded = false;
};
int ensure_arena(qip_hip_state* s, size_t bytes);
// start of a launch group (one op, one tile segment): while a program records, its payload gets a region of its own
static inline void arena_begin_group(qip_hip_state* s) {
  if (s->capture_pool) {
    s->arena = nullptr;
    s->arena_cap = 0;
  }
}
int ensure_partial(qip_hip_state* s, size_t count);
int ensure_alt(qip_hip_state* s);
void programs_orphan(qip_hip_state* s);  // qip_circuit.hip
int jit_set_cache_cap(int64_t cap);      // qip_circuit.hip (global option "jit_cache_cap")
int jit_set_disk_cap_mb(int64_t mb);     // qip_circuit.hip (global option "jit_disk_cap_mb")
uint64_t jit_cache_generation();
// qip_circuit.hip: `op` as a one-item tile sweep; *done = false when it is not a tile item (nothing launched)
template <typename T> int tile_apply_single(qip_hip_state* s, const qip_op* op, bool* done, double alg_bytes = 0);
int prof_begin(qip_hip_state* s, int cls, double bytes, ProfRec* r);
int prof_end(qip_hip_state* s, ProfRec* r);
int state_settle(qip_hip_state* s);  // qip_launch.hip: a relabelled state back to the caller's order (one permutation sweep)

#define STATE_ENTER_NOCHECK(s)                                 \
  if (!(s)) return fail(QIP_ERR_INVALID, "null state handle"); \
  HIPCHK(hipSetDevice((s)->device))
#define STATE_ENTER_RAW(s)                                                                                                    \
  STATE_ENTER_NOCHECK(s);                                                                                                     \
  if ((s)->poisoned)                                                                                                          \
  return fail(QIP_ERR_DEVICE, "state unusable after a relabelled batch failed half way (re-initialise it): %s", (s)->poison_msg.c_str())
#define STATE_ENTER(s) \
  STATE_ENTER_RAW(s);  \
  if (!(s)->layout.empty()) QCHK(state_settle(s))

static inline unsigned grid_for(uint64_t items, uint64_t per_block) {
  uint64_t g = (items + per_block - 1) / per_block;
  if (g == 0) g = 1;
  return (unsigned)std::min<uint64_t>(g, 0x7fffffffull);
}
// block count -> grid; a second dimension keeps every launch under HIP's 2^32-thread limit
static inline dim3 grid2d(uint64_t items, uint64_t per_block) {
  uint64_t g = (items + per_block - 1) / per_block;
  if (g == 0) g = 1;
  const uint64_t gx = std::min<uint64_t>(g, 1ull << 22);
  return dim3((unsigned)gx, (unsigned)((g + gx - 1) / gx));
}
static inline unsigned grid_stride(uint64_t items) {
  // memory-bound grid-stride kernels: enough workgroups to fill 256 CUs x 8
  return (unsigned)std::min<uint64_t>(std::max<uint64_t>((items + kBlock - 1) / kBlock, 1), 256 * 16);
}

// qip_launch.hip
Ins make_ins(std::vector<uint32_t> positions, uint64_t ormask);
int arena_upload(qip_hip_state* s, const void* src, size_t bytes, size_t arena_off);
int launch_permute(qip_hip_state* s, const uint32_t* pi_in);
template <typename T> int apply_op_t(qip_hip_state* s, const qip_op* op);
template <typename T>
int launch_gather(qip_hip_state* s, const FlatOp& f, const amp_t<T>* in, uint64_t in_len, amp_t<T>* out, uint64_t out_len,
                  uint64_t in_off, uint64_t out_off, int accumulate);

template <typename T> static amp_t<T> mk(double re, double im) {
  amp_t<T> a;
  a.x = (T)re;
  a.y = (T)im;
  return a;
}
static inline bool use_nt(const qip_hip_state* s) { return s->namps * s->amp_bytes >= (1ull << 30); }
