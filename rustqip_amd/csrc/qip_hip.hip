// qip_hip.hip — C ABI (include/qip_hip.h) over the gfx950 kernels in qip_kernels.h.
//
// Host side of the drop-in boundary: validates op descriptors the way the reference's
// constructors do (qip/src/state_ops/matrix_ops.rs:12-122), classifies each op into the
// cheapest kernel that is result-identical to the reference's gather formulation
// (qip-iterators/src/matrix_ops.rs:62-152), and launches it on the handle's HIP stream.
// There is NO CPU fallback: without a HIP device every compute entry point fails with
// QIP_ERR_NO_DEVICE.
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
#include <vector>

#include "../../include/qip_hip.h"
#include "qip_kernels.h"

using namespace qipk;

// ---------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

static int fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

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

extern "C" const char* qip_hip_last_error(void) { return g_last_error.c_str(); }
extern "C" int qip_hip_abi_version(void) { return 3; }  // 3: + permute_bits, dist_rank_flip, options tile_relabel / perm_rows / line_bits
extern "C" int qip_hip_device_count(void) try {
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return c;
} QIP_CATCH_ALL

static int64_t g_force_generic = 0;
// Selector bits below this position stay in the grid as a per-lane predicate (whole lines are swept); see kLineBits.
static uint32_t g_line_bits = qipk::kLineBits;
static int64_t g_perm_rows = 0;  // row bits of k_permute_bits for 16-byte elements: 0 = by the permutation, 5 / 6 = forced (tuning aid)
extern "C" int qip_hip_set_global_option(const char* key, int64_t value) try {
  if (key && !strcmp(key, "force_generic")) {
    g_force_generic = value;
    return QIP_OK;
  }
  if (key && !strcmp(key, "perm_rows")) {
    if (value != 0 && value != 5 && value != 6) return fail(QIP_ERR_INVALID, "perm_rows must be 0 (automatic), 5 or 6");
    g_perm_rows = value;
    return QIP_OK;
  }
  if (key && !strcmp(key, "line_bits")) {  // tuning aid (tools/bench_ops.py): 0..3
    if (value < 0 || value > 3) return fail(QIP_ERR_INVALID, "line_bits must be 0..3");
    g_line_bits = (uint32_t)value;
    return QIP_OK;
  }
  return fail(QIP_ERR_INVALID, "unknown global option '%s'", key ? key : "(null)");
} QIP_CATCH_ALL

// ---------------------------------------------------------------------------------------
// op flattening + validation
// ---------------------------------------------------------------------------------------
struct FlatOp {
  const qip_op* outer = nullptr;
  const qip_op* inner = nullptr;  // innermost non-Control op
  uint32_t k_all = 0;             // outer->n_indices
  uint32_t n_control = 0;         // flattened (ops.rs:150-154)
  uint32_t n_op = 0;              // indices the inner iterator is built with
  bool distinct = true;           // all outer indices distinct
};

// strict = what make_*_op rejects; always = what would panic / read out of bounds in the
// reference kernel.
static int flatten_op(uint32_t n, const qip_op* op, bool strict, FlatOp* f) {
  if (!op) return fail(QIP_ERR_INVALID, "null op");
  if (n == 0 || n > 62) return fail(QIP_ERR_INVALID, "n = %u out of range [1, 62]", n);
  if (op->kind < QIP_OP_MATRIX || op->kind > QIP_OP_CONTROL)
    return fail(QIP_ERR_INVALID, "unknown op kind %d", op->kind);
  if (op->n_indices == 0)
    return fail(QIP_ERR_INVALID, "Must supply at least one op index");  // matrix_ops.rs:15-16
  if (op->n_indices > (uint32_t)kMaxIns || !op->indices)
    return fail(QIP_ERR_INVALID, "op has %u indices (max %d) or a null index list", op->n_indices,
                kMaxIns);
  f->outer = op;
  f->k_all = op->n_indices;
  uint64_t seen = 0;
  for (uint32_t j = 0; j < op->n_indices; ++j) {
    if (op->indices[j] >= n)
      return fail(QIP_ERR_INVALID, "qubit index %llu out of range for n = %u",
                  (unsigned long long)op->indices[j], n);
    if (seen & (1ull << op->indices[j])) f->distinct = false;
    seen |= 1ull << op->indices[j];
  }
  const qip_op* inner = op;
  uint32_t n_control = 0, n_op = op->n_indices;
  if (op->kind == QIP_OP_CONTROL) {
    if (op->n_controls == 0)
      return fail(QIP_ERR_INVALID, "Must supply at least one control index");  // :107-108
    if (op->n_controls >= op->n_indices)
      return fail(QIP_ERR_INVALID, "Control op needs at least one op index after its %u controls",
                  op->n_controls);
    if (!op->inner) return fail(QIP_ERR_INVALID, "Control op without inner op");
    n_control = op->n_controls;
    n_op = op->n_indices - op->n_controls;
    inner = op->inner;
    int depth = 0;
    while (inner->kind == QIP_OP_CONTROL) {
      if (!inner->inner || inner->n_controls == 0 || inner->n_controls >= inner->n_indices ||
          ++depth > 64)
        return fail(QIP_ERR_INVALID, "malformed nested Control op");
      n_control += inner->n_controls;
      n_op = inner->n_indices - inner->n_controls;
      inner = inner->inner;
    }
    if (inner->kind < QIP_OP_MATRIX || inner->kind > QIP_OP_SWAP)
      return fail(QIP_ERR_INVALID, "unknown inner op kind %d", inner->kind);
    if (n_control + n_op != op->n_indices)
      return fail(QIP_ERR_INVALID,
                  "Control op lists %u indices but its controls (%u) + inner op indices (%u) differ",
                  op->n_indices, n_control, n_op);
  }
  f->inner = inner;
  f->n_control = n_control;
  f->n_op = n_op;
  if (n_op > 30) return fail(QIP_ERR_UNSUPPORTED, "inner op on %u qubits is too large", n_op);
  switch (inner->kind) {
    case QIP_OP_MATRIX:
      if (!inner->dense) return fail(QIP_ERR_INVALID, "Matrix op without data");
      // make_matrix_op :17-23 checks dat.len() == 4^k; here the length is implied by the
      // ABI (4^n_op entries are read).  inner->indices are ignored, as in the reference.
      break;
    case QIP_OP_SPARSE: {
      if (!inner->sparse_rowptr) return fail(QIP_ERR_INVALID, "Sparse op without row pointers");
      const uint64_t rows = 1ull << n_op;
      if (inner->sparse_rowptr[0] != 0) return fail(QIP_ERR_INVALID, "Sparse rowptr[0] != 0");
      for (uint64_t r = 0; r < rows; ++r) {
        if (inner->sparse_rowptr[r + 1] < inner->sparse_rowptr[r])
          return fail(QIP_ERR_INVALID, "Sparse rowptr not monotone at row %llu",
                      (unsigned long long)r);
        if (strict && inner->sparse_rowptr[r + 1] == inner->sparse_rowptr[r])
          return fail(QIP_ERR_INVALID, "All rows of sparse matrix must have data (%llu is empty)",
                      (unsigned long long)r);  // :49-58
      }
      const uint64_t nnz = inner->sparse_rowptr[rows];
      if (nnz && (!inner->sparse_cols || !inner->sparse_vals))
        return fail(QIP_ERR_INVALID, "Sparse op without column/value arrays");
      for (uint64_t p = 0; p < nnz; ++p)
        if (inner->sparse_cols[p] >= rows)
          return fail(QIP_ERR_INVALID, "Sparse column %llu out of range for %u qubits",
                      (unsigned long long)inner->sparse_cols[p], n_op);
      break;
    }
    case QIP_OP_SWAP:
      if (n_op % 2 != 0 || n_op == 0)
        return fail(QIP_ERR_INVALID,
                    "Swap must be performed on two sets of indices of equal length");  // :87-93
      break;
    default:
      break;
  }
  return QIP_OK;
}

extern "C" int qip_hip_validate_op(uint32_t n, const qip_op* op) try {
  FlatOp f;
  return flatten_op(n, op, /*strict=*/true, &f);
} QIP_CATCH_ALL

// ---------------------------------------------------------------------------------------
// kernel classes (profiling + algorithmic bytes)
// ---------------------------------------------------------------------------------------
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
  KC_COUNT
};
static const char* kKernelClassNames[KC_COUNT] = {
    "k_gate1q_pair", "k_gate1q_xlane", "k_phase",          "k_diag",           "k_diag1q",
    "k_swap_bits",   "k_gate_kq",      "k_gate_kq_mfma",   "k_tile_gates",     "k_gather_generic",
    "noop_identity", "k_sparse_kq", "k_gate_big_mfma", "k_permute_bits"};

extern "C" int qip_hip_kernel_class_count(void) { return KC_COUNT; }
extern "C" const char* qip_hip_kernel_class_name(int cls) {
  return (cls >= 0 && cls < KC_COUNT) ? kKernelClassNames[cls] : "";
}

// ---------------------------------------------------------------------------------------
// planning: which kernel applies an op
// ---------------------------------------------------------------------------------------
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

template <typename T>
static void read_dense(const void* dense, uint64_t count, std::vector<double>* out) {
  const T* p = static_cast<const T*>(dense);
  out->resize(count * 2);
  for (uint64_t i = 0; i < count * 2; ++i) (*out)[i] = (double)p[i];
}

static constexpr uint32_t kMaxRegK = 4;     // dense gates held in registers (VALU form)
static constexpr uint32_t kMaxMfmaK = 5;    // dense gates on the f64 matrix cores: k = 3..5 (A operand in registers)
static constexpr uint32_t kMaxBigK = 8;     // ... k = 6..8 with the A operand streamed through LDS (k_gate_big_mfma)
static constexpr uint32_t kMaxSparseK = 5;  // SparseMatrix ops applied in place (one 2^k group per lane, staged in LDS)
static constexpr uint32_t kMaxDiagK = 12;   // largest Matrix op inspected for structure (4^k entries are read)

static int make_plan(int dtype, uint32_t n, const FlatOp& f, bool force_generic, Plan* p) {
  const double amp_bytes = dtype == QIP_C64 ? 16.0 : 8.0;
  const double N = std::ldexp(1.0, (int)n);
  p->cls = KC_GATHER_GENERIC;
  p->alg_bytes = 2.0 * amp_bytes * std::ldexp(1.0, (int)(n - f.n_control));
  for (uint32_t j = 0; j < f.n_control; ++j) p->cpos.push_back(n - 1 - (uint32_t)f.outer->indices[j]);
  for (uint32_t j = f.n_control; j < f.k_all; ++j)
    p->opos.push_back(n - 1 - (uint32_t)f.outer->indices[j]);
  (void)N;
  if (force_generic || !f.distinct) return QIP_OK;

  const uint32_t k = f.n_op;
  if (f.inner->kind == QIP_OP_SWAP) {
    p->cls = KC_SWAP_BITS;
    return QIP_OK;
  }
  if (f.inner->kind == QIP_OP_SPARSE) {
    if (k <= kMaxSparseK) p->cls = KC_SPARSE_KQ;  // in place, stored order (qubit_iterators.rs:87-101)
    return QIP_OK;
  }
  if (f.inner->kind != QIP_OP_MATRIX) return QIP_OK;
  if (k > kMaxDiagK) return QIP_OK;

  const uint64_t side = 1ull << k;
  // entries are read in place (no 4^k host copy just to look at the structure); the off-diagonal scan exits at the
  // first non-zero, so a dense matrix costs O(1) here and only a truly diagonal one is walked completely
  auto re_of = [&](uint64_t e) { return dtype == QIP_C64 ? static_cast<const double*>(f.inner->dense)[2 * e] : (double)static_cast<const float*>(f.inner->dense)[2 * e]; };
  auto im_of = [&](uint64_t e) { return dtype == QIP_C64 ? static_cast<const double*>(f.inner->dense)[2 * e + 1] : (double)static_cast<const float*>(f.inner->dense)[2 * e + 1]; };
  bool diag = true;
  for (uint64_t r = 0; r < side && diag; ++r)
    for (uint64_t c = 0; c < side; ++c)
      if (r != c && !is_zero2(re_of(r * side + c), im_of(r * side + c))) {
        diag = false;
        break;
      }
  if (diag) {
    uint64_t non_one = 0, last = 0;
    for (uint64_t r = 0; r < side; ++r)
      if (!is_one2(re_of(r * side + r), im_of(r * side + r))) {
        ++non_one;
        last = r;
      }
    if (non_one == 0) {
      p->cls = KC_NOOP;
      p->alg_bytes = 0;
      return QIP_OK;
    }
    if (non_one == 1) {
      p->cls = KC_PHASE;
      p->phase_ones = last;
      p->phase[0] = re_of(last * side + last);
      p->phase[1] = im_of(last * side + last);
      p->alg_bytes = 2.0 * amp_bytes * std::ldexp(1.0, (int)(n - f.n_control - k));
      return QIP_OK;
    }
    p->cls = KC_DIAG;
    p->table.resize(side * 2);
    for (uint64_t r = 0; r < side; ++r) {
      p->table[2 * r] = re_of(r * side + r);
      p->table[2 * r + 1] = im_of(r * side + r);
    }
    p->alg_bytes = 2.0 * amp_bytes * std::ldexp(1.0, (int)(n - f.n_control - k)) * (double)non_one;
    return QIP_OK;
  }
  std::vector<double> d;
  if (k <= kMaxBigK) {
    if (dtype == QIP_C64)
      read_dense<double>(f.inner->dense, side * side, &d);
    else
      read_dense<float>(f.inner->dense, side * side, &d);
  }
  if (k == 1) {
    p->cls = KC_GATE1Q_PAIR;  // the launcher may pick the cross-lane variant
    p->nz = 0;
    for (int e = 0; e < 4; ++e) {
      p->m[2 * e] = d[2 * e];
      p->m[2 * e + 1] = d[2 * e + 1];
      if (!is_zero2(d[2 * e], d[2 * e + 1])) p->nz |= 1u << e;
    }
    return QIP_OK;
  }
  if (k <= kMaxBigK) {
    // the launcher picks the matrix-core forms (k in 3..5 / 6..8) when >= 16 groups exist
    p->cls = KC_GATE_KQ;
    p->table = d;
    return QIP_OK;
  }
  return QIP_OK;
}

extern "C" int qip_hip_op_algorithmic_bytes(int dtype, uint32_t n, const qip_op* op, double* bytes) try {
  if (!bytes) return fail(QIP_ERR_INVALID, "null output");
  if (dtype != QIP_C64 && dtype != QIP_C32) return fail(QIP_ERR_INVALID, "bad dtype %d", dtype);
  FlatOp f;
  QCHK(flatten_op(n, op, false, &f));
  Plan p;
  QCHK(make_plan(dtype, n, f, false, &p));
  // Swap(h): only amplitudes whose A and B halves differ can change, but SURVEY.md §8(d)
  // prices Swap at the full vector; keep that convention for the reported figure.
  *bytes = p.alg_bytes;
  return QIP_OK;
} QIP_CATCH_ALL

// ---------------------------------------------------------------------------------------
// state handle
// ---------------------------------------------------------------------------------------
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
  int64_t tile_relabel = 0;  // tile sweeps: the scheduler relabels the qubits (schedule_tiles_relabel)
  int64_t swap_single = 0;  // 1 = one sweep per transposition (tuning aid; default groups them, k_swapn)
  int64_t tile_jit = 0;     // 1 = tile segments run as kernels compiled at run time for that very segment (hiprtc, cached)
  bool jit_prepare = false; // compile the segments' kernels but launch nothing (before a graph capture)
  // program capture (hipGraph): payload staging that must outlive the graph, and arena growth request
  std::deque<std::vector<char>>* capture_staging = nullptr;
  size_t capture_arena_need = 0;
  std::vector<struct qip_hip_program*> programs;  // graphs recorded against this state's buffers  // f32: sweep two amplitudes per 16-B element when bit 0 is not involved  // 0 = gate by gate; K >= 2 = fuse into dense gates on <= K qubits
  int64_t unroll = 0;  // 0 = default per kernel
  // profiling
  std::vector<ProfRec> pending;
  std::vector<hipEvent_t> free_events;
  uint64_t prof_launches[KC_COUNT] = {0};
  double prof_ms[KC_COUNT] = {0};
  double prof_bytes[KC_COUNT] = {0};
};

static int ensure_arena(qip_hip_state* s, size_t bytes) {
  if (bytes <= s->arena_cap) return QIP_OK;
  if (s->capture_staging) {  // no malloc / sync inside a stream capture: ask the caller to grow and retry
    s->capture_arena_need = std::max(s->capture_arena_need, bytes);
    return fail(QIP_ERR_UNSUPPORTED, "arena too small during graph capture");
  }
  if (s->arena) {
    HIPCHK(hipStreamSynchronize(s->stream));
    HIPCHK(hipFree(s->arena));
    s->arena = nullptr;
    s->arena_cap = 0;
  }
  size_t cap = std::max<size_t>(bytes, 1 << 16);
  HIPCHK(hipMalloc(&s->arena, cap));
  s->arena_cap = cap;
  s->arena_gen += 1;
  return QIP_OK;
}

static int ensure_partial(qip_hip_state* s, size_t count) {
  if (count <= s->partial_cap) return QIP_OK;
  if (s->d_partial) {
    HIPCHK(hipStreamSynchronize(s->stream));
    HIPCHK(hipFree(s->d_partial));
    s->d_partial = nullptr;
    s->partial_cap = 0;
  }
  size_t cap = std::max<size_t>(count, 4096);
  HIPCHK(hipMalloc((void**)&s->d_partial, cap * sizeof(double)));
  s->partial_cap = cap;
  return QIP_OK;
}

static int ensure_alt(qip_hip_state* s) {
  if (s->alt) return QIP_OK;
  HIPCHK(hipMalloc(&s->alt, s->namps * s->amp_bytes));
  s->owns_alt = true;
  return QIP_OK;
}

static int state_new(uint32_t n, int dtype, int device, qip_hip_state** out) {
  if (!out) return fail(QIP_ERR_INVALID, "null output handle");
  *out = nullptr;
  if (dtype != QIP_C64 && dtype != QIP_C32) return fail(QIP_ERR_INVALID, "bad dtype %d", dtype);
  if (n == 0 || n > 40) return fail(QIP_ERR_INVALID, "n = %u out of range [1, 40]", n);
  int count = qip_hip_device_count();
  if (count <= 0)
    return fail(QIP_ERR_NO_DEVICE,
                "no HIP device visible: qip_hip has no CPU fallback (hipGetDeviceCount = 0)");
  if (device < 0 || device >= count)
    return fail(QIP_ERR_INVALID, "device %d out of range (have %d)", device, count);
  HIPCHK(hipSetDevice(device));
  qip_hip_state* s = new qip_hip_state();
  s->n = n;
  s->dtype = dtype;
  s->device = device;
  s->namps = 1ull << n;
  s->amp_bytes = dtype == QIP_C64 ? 16 : 8;
  *out = s;
  return QIP_OK;
}

extern "C" int qip_hip_state_create(uint32_t n, int dtype, int device, qip_hip_state** out) try {
  QCHK(state_new(n, dtype, device, out));
  qip_hip_state* s = *out;
  hipError_t e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
  if (e == hipSuccess) {
    s->owns_stream = true;
    e = hipMalloc(&s->cur, s->namps * s->amp_bytes);
  }
  if (e == hipSuccess) {
    s->owns_cur = true;
    e = hipMemsetAsync(s->cur, 0, s->namps * s->amp_bytes, s->stream);
  }
  if (e != hipSuccess) {
    qip_hip_state_destroy(s);
    *out = nullptr;
    return fail(QIP_ERR_DEVICE, "allocating a %u-qubit state failed: %s", n, hipGetErrorString(e));
  }
  return QIP_OK;
} QIP_CATCH_ALL

extern "C" int qip_hip_state_wrap(uint32_t n, int dtype, int device, void* amps, void* scratch,
                                  void* stream, qip_hip_state** out) try {
  if (!amps) return fail(QIP_ERR_INVALID, "null amplitude buffer");
  QCHK(state_new(n, dtype, device, out));
  qip_hip_state* s = *out;
  s->cur = amps;
  s->alt = scratch;
  // the caller's stream as is; NULL is the HIP null (legacy default) stream, which is what
  // torch.cuda.current_stream().cuda_stream reports for torch's default stream on ROCm
  s->stream = (hipStream_t)stream;
  s->owns_stream = false;
  return QIP_OK;
} QIP_CATCH_ALL

static void programs_orphan(qip_hip_state* s);

extern "C" int qip_hip_state_destroy(qip_hip_state* s) try {
  if (!s) return QIP_OK;
  (void)hipSetDevice(s->device);
  if (s->stream) (void)hipStreamSynchronize(s->stream);
  programs_orphan(s);  // programs outliving their state become inert instead of dangling
  for (auto& r : s->pending) {
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
  }
  for (auto e : s->free_events) (void)hipEventDestroy(e);
  if (s->owns_cur && s->cur) (void)hipFree(s->cur);
  if (s->owns_alt && s->alt) (void)hipFree(s->alt);
  if (s->arena) (void)hipFree(s->arena);
  if (s->d_partial) (void)hipFree(s->d_partial);
  if (s->owns_stream && s->stream) (void)hipStreamDestroy(s->stream);
  delete s;
  return QIP_OK;
} QIP_CATCH_ALL

#define STATE_ENTER(s)                                        \
  if (!(s)) return fail(QIP_ERR_INVALID, "null state handle"); \
  HIPCHK(hipSetDevice((s)->device))

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

extern "C" int qip_hip_state_init_basis(qip_hip_state* s, uint64_t index) try {
  STATE_ENTER(s);
  if (index >= s->namps) return fail(QIP_ERR_INVALID, "basis index out of range");
  HIPCHK(hipMemsetAsync(s->cur, 0, s->namps * s->amp_bytes, s->stream));
  if (s->dtype == QIP_C64) {
    const double one[2] = {1.0, 0.0};
    HIPCHK(hipMemcpyAsync((char*)s->cur + index * 16, one, 16, hipMemcpyHostToDevice, s->stream));
  } else {
    const float one[2] = {1.0f, 0.0f};
    HIPCHK(hipMemcpyAsync((char*)s->cur + index * 8, one, 8, hipMemcpyHostToDevice, s->stream));
  }
  HIPCHK(hipStreamSynchronize(s->stream));
  return QIP_OK;
} QIP_CATCH_ALL

extern "C" int qip_hip_state_upload(qip_hip_state* s, const void* src, uint64_t offset, uint64_t len) try {
  STATE_ENTER(s);
  if (offset > s->namps || len > s->namps - offset) return fail(QIP_ERR_INVALID, "upload range out of bounds");
  if (len == 0) return QIP_OK;
  if (!src) return fail(QIP_ERR_INVALID, "null source");
  HIPCHK(hipMemcpyAsync((char*)s->cur + offset * s->amp_bytes, src, len * s->amp_bytes,
                        hipMemcpyHostToDevice, s->stream));
  HIPCHK(hipStreamSynchronize(s->stream));
  return QIP_OK;
} QIP_CATCH_ALL

extern "C" int qip_hip_state_download(qip_hip_state* s, void* dst, uint64_t offset, uint64_t len) try {
  STATE_ENTER(s);
  if (offset > s->namps || len > s->namps - offset) return fail(QIP_ERR_INVALID, "download range out of bounds");
  if (len == 0) return QIP_OK;
  if (!dst) return fail(QIP_ERR_INVALID, "null destination");
  HIPCHK(hipMemcpyAsync(dst, (const char*)s->cur + offset * s->amp_bytes, len * s->amp_bytes,
                        hipMemcpyDeviceToHost, s->stream));
  HIPCHK(hipStreamSynchronize(s->stream));
  return QIP_OK;
} QIP_CATCH_ALL

extern "C" int qip_hip_state_device_ptr(qip_hip_state* s, void** amps) try {
  if (!s || !amps) return fail(QIP_ERR_INVALID, "null argument");
  *amps = s->cur;
  return QIP_OK;
} QIP_CATCH_ALL

extern "C" int qip_hip_state_scratch_ptr(qip_hip_state* s, void** scratch) try {
  STATE_ENTER(s);
  if (!scratch) return fail(QIP_ERR_INVALID, "null argument");
  QCHK(ensure_alt(s));
  *scratch = s->alt;
  return QIP_OK;
} QIP_CATCH_ALL

extern "C" int qip_hip_state_swap_buffers(qip_hip_state* s) try {
  STATE_ENTER(s);
  QCHK(ensure_alt(s));
  std::swap(s->cur, s->alt);
  std::swap(s->owns_cur, s->owns_alt);
  return QIP_OK;
} QIP_CATCH_ALL

extern "C" int qip_hip_state_sync(qip_hip_state* s) try {
  STATE_ENTER(s);
  HIPCHK(hipStreamSynchronize(s->stream));
  return QIP_OK;
} QIP_CATCH_ALL

extern "C" int qip_hip_state_set_option(qip_hip_state* s, const char* key, int64_t value) try {
  if (!s || !key) return fail(QIP_ERR_INVALID, "null argument");
  if (!strcmp(key, "force_generic")) s->force_generic = value;
  else if (!strcmp(key, "profile")) s->profile = value;
  else if (!strcmp(key, "lowbit_shuffle")) s->lowbit_shuffle = value;
  else if (!strcmp(key, "mfma")) s->mfma = value;
  else if (!strcmp(key, "fuse")) s->fuse = value;
  else if (!strcmp(key, "packed_f32")) s->packed_f32 = value;
  else if (!strcmp(key, "tile")) s->tile = value;
  else if (!strcmp(key, "tile_passes")) s->tile_passes = value;
  else if (!strcmp(key, "unroll")) s->unroll = value;
  else if (!strcmp(key, "swap_single")) s->swap_single = value;
  else if (!strcmp(key, "tile_jit")) s->tile_jit = value;
  else if (!strcmp(key, "tile_relabel")) s->tile_relabel = value;
  else return fail(QIP_ERR_INVALID, "unknown option '%s'", key);
  return QIP_OK;
} QIP_CATCH_ALL

// ---- profiling ---------------------------------------------------------------------------
static int prof_begin(qip_hip_state* s, int cls, double bytes, ProfRec* r) {
  r->cls = cls;
  r->bytes = bytes;
  for (hipEvent_t* e : {&r->e0, &r->e1}) {
    if (!s->free_events.empty()) {
      *e = s->free_events.back();
      s->free_events.pop_back();
    } else {
      HIPCHK(hipEventCreate(e));
    }
  }
  HIPCHK(hipEventRecord(r->e0, s->stream));
  return QIP_OK;
}
static int prof_end(qip_hip_state* s, ProfRec* r) {
  HIPCHK(hipEventRecord(r->e1, s->stream));
  s->pending.push_back(*r);
  return QIP_OK;
}
static int prof_drain(qip_hip_state* s) {
  if (s->pending.empty()) return QIP_OK;
  HIPCHK(hipStreamSynchronize(s->stream));
  for (auto& r : s->pending) {
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, r.e0, r.e1));
    s->prof_launches[r.cls] += 1;
    s->prof_ms[r.cls] += ms;
    s->prof_bytes[r.cls] += r.bytes;
    s->free_events.push_back(r.e0);
    s->free_events.push_back(r.e1);
  }
  s->pending.clear();
  return QIP_OK;
}

extern "C" int qip_hip_state_profile_get(qip_hip_state* s, int cls, uint64_t* launches,
                                         double* total_ms, double* algorithmic_bytes) try {
  STATE_ENTER(s);
  if (cls < 0 || cls >= KC_COUNT) return fail(QIP_ERR_INVALID, "bad kernel class %d", cls);
  QCHK(prof_drain(s));
  if (launches) *launches = s->prof_launches[cls];
  if (total_ms) *total_ms = s->prof_ms[cls];
  if (algorithmic_bytes) *algorithmic_bytes = s->prof_bytes[cls];
  return QIP_OK;
} QIP_CATCH_ALL
extern "C" int qip_hip_state_profile_reset(qip_hip_state* s) try {
  STATE_ENTER(s);
  QCHK(prof_drain(s));
  for (int c = 0; c < KC_COUNT; ++c) {
    s->prof_launches[c] = 0;
    s->prof_ms[c] = 0;
    s->prof_bytes[c] = 0;
  }
  return QIP_OK;
} QIP_CATCH_ALL

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
static Ins make_ins(std::vector<uint32_t> positions, uint64_t ormask) {
  std::sort(positions.begin(), positions.end());
  Ins ins;
  memset(&ins, 0, sizeof ins);
  ins.ormask = ormask;
  ins.npos = (uint32_t)positions.size();
  for (size_t j = 0; j < positions.size(); ++j) ins.pos[j] = positions[j];
  return ins;
}

static uint64_t mask_of(const std::vector<uint32_t>& pos) {
  uint64_t m = 0;
  for (uint32_t p : pos) m |= 1ull << p;
  return m;
}

template <typename T> static amp_t<T> mk(double re, double im) {
  amp_t<T> a;
  a.x = (T)re;
  a.y = (T)im;
  return a;
}

// Stream-ordered copy of an op payload into the device arena.  Eagerly the (pageable) source is staged by
// the runtime before the call returns; under graph capture the source must live as long as the graph, so
// it is first copied into storage owned by the program.
static int arena_upload(qip_hip_state* s, const void* src, size_t bytes, size_t arena_off) {
  if (bytes == 0) return QIP_OK;
  QCHK(ensure_arena(s, arena_off + bytes));
  if (s->capture_staging) {
    s->capture_staging->emplace_back((const char*)src, (const char*)src + bytes);
    src = s->capture_staging->back().data();
  }
  HIPCHK(hipMemcpyAsync((char*)s->arena + arena_off, src, bytes, hipMemcpyHostToDevice, s->stream));
  return QIP_OK;
}

// upload `count` complex values (host doubles re,im) to the device arena as amp_t<T>
template <typename T>
static int upload_table(qip_hip_state* s, const std::vector<double>& tab, size_t arena_off = 0) {
  const size_t count = tab.size() / 2;
  std::vector<amp_t<T>> tmp(count);
  for (size_t i = 0; i < count; ++i) tmp[i] = mk<T>(tab[2 * i], tab[2 * i + 1]);
  return arena_upload(s, tmp.data(), count * sizeof(amp_t<T>), arena_off);
}

// Compile-time position counts 0..4 cover every 1- and 2-qubit gate with up to two extra
// controls; anything longer takes the run-time loop (NP = -1).
template <typename F> static void dispatch_np(uint32_t npos, F&& f) {
  switch (npos) {
    case 0: f(std::integral_constant<int, 0>{}); break;
    case 1: f(std::integral_constant<int, 1>{}); break;
    case 2: f(std::integral_constant<int, 2>{}); break;
    case 3: f(std::integral_constant<int, 3>{}); break;
    case 4: f(std::integral_constant<int, 4>{}); break;
    default: f(std::integral_constant<int, -1>{}); break;
  }
}

// States that cannot stay in the 256-MiB Infinity Cache stream with non-temporal accesses.
static inline bool use_nt(const qip_hip_state* s) { return s->namps * s->amp_bytes >= (1ull << 30); }

// (E, the 16-B element type, must be in scope: amp_t<T>, or f32x4 for the packed f32 view.)
// LAUNCH_STREAMING(kernel, T, U, count, ins, args...): the unguarded <U> shape with the 32-KiB lane
// spacing when the power-of-two work-item count allows it, else the guarded single-item shape.
#define LAUNCH_STREAMING(KERNEL, T, UU, COUNT, INS, ...)                                           \
  dispatch_np((INS).npos, [&](auto np_) {                                                          \
    constexpr int NP = decltype(np_)::value;                                                       \
    if ((COUNT) >= ((uint64_t)(UU) << kStrideShift)) {                                             \
      if (use_nt(s))                                                                               \
        hipLaunchKernelGGL((KERNEL<T, UU, false, true, NP, E>), grid2d((COUNT), kBlock * (UU)),       \
                           dim3(kBlock), 0, s->stream, __VA_ARGS__);                               \
      else                                                                                         \
        hipLaunchKernelGGL((KERNEL<T, UU, false, false, NP, E>), grid2d((COUNT), kBlock * (UU)),      \
                           dim3(kBlock), 0, s->stream, __VA_ARGS__);                               \
    } else {                                                                                       \
      hipLaunchKernelGGL((KERNEL<T, 1, true, false, NP, E>), dim3(grid_for((COUNT), kBlock)),     \
                         dim3(kBlock), 0, s->stream, __VA_ARGS__);                                 \
    }                                                                                              \
  })

// independent accesses per stream per lane (tools/tune_gate1q.hip, MI355X, n = 30)
constexpr int kUPair = 8;   // two streams per item (the zero-skip branches need the longer load phase)
constexpr int kUSwap = 4;
constexpr int kUXlane = 4;
constexpr int kUPhase = 2;

// Split selector bit positions (controls / phase bits, all required to be 1 unless `ones` says
// otherwise) into the ones opened in the grid (>= kLineBits) and the in-line predicate (`Sel`).
struct Split {
  std::vector<uint32_t> hi;  // positions opened in the work index
  uint64_t hi_ones = 0;      // bits to set among them
  Sel low{0, 0};
};
static Split split_selectors(const std::vector<uint32_t>& pos, uint64_t ones_mask) {
  Split sp;
  for (uint32_t p : pos) {
    const uint64_t bit = 1ull << p;
    if (p < g_line_bits) {
      sp.low.mask |= bit;
      sp.low.val |= ones_mask & bit;
    } else {
      sp.hi.push_back(p);
      sp.hi_ones |= ones_mask & bit;
    }
  }
  return sp;
}
// work-index bit that amplitude-index bit `pos` maps to once the opened positions below it are removed
static uint32_t work_bit(uint32_t pos, const std::vector<uint32_t>& opened) {
  uint32_t below = 0;
  for (uint32_t o : opened)
    if (o < pos) ++below;
  return pos - below;
}

template <typename T, typename E>
static int launch_gate1q(qip_hip_state* s, uint32_t n, const Plan& p, E* st, int* actual_cls) {
  const uint32_t tpos = p.opos[0];
  Mat2<T> g;
  for (int e = 0; e < 4; ++e) g.m[e] = mk<T>(p.m[2 * e], p.m[2 * e + 1]);
  g.nz = p.nz;
  const Split sp = split_selectors(p.cpos, mask_of(p.cpos));
  const uint32_t tb = work_bit(tpos, sp.hi);
  const uint64_t namps_sub = 1ull << (n - (uint32_t)sp.hi.size());
  if (s->lowbit_shuffle && tb < 6 && namps_sub >= 64) {
    Ins ins = make_ins(sp.hi, sp.hi_ones);
    *actual_cls = KC_GATE1Q_XLANE;
    LAUNCH_STREAMING(k_gate1q_xlane, T, kUXlane, namps_sub, ins, st, namps_sub, ins, tb, sp.low, g);
  } else {
    std::vector<uint32_t> pos = sp.hi;
    pos.push_back(tpos);
    Ins ins = make_ins(pos, sp.hi_ones);
    const uint64_t npairs = namps_sub >> 1;
    const uint64_t tmask = 1ull << tpos;
    LAUNCH_STREAMING(k_gate1q_pair, T, kUPair, npairs, ins, st, npairs, ins, tmask, sp.low, g);
  }
  HIPCHK(hipGetLastError());
  return QIP_OK;
}

template <typename T, typename E>
static int launch_phase(qip_hip_state* s, uint32_t n, const Plan& p, E* st) {
  const uint32_t k = (uint32_t)p.opos.size();
  std::vector<uint32_t> pos = p.cpos;
  uint64_t ones = mask_of(p.cpos);
  for (uint32_t j = 0; j < k; ++j) {
    pos.push_back(p.opos[j]);
    if ((p.phase_ones >> (k - 1 - j)) & 1ull) ones |= 1ull << p.opos[j];
  }
  const Split sp = split_selectors(pos, ones);
  Ins ins = make_ins(sp.hi, sp.hi_ones);
  const uint64_t count = 1ull << (n - (uint32_t)sp.hi.size());
  const amp_t<T> value = mk<T>(p.phase[0], p.phase[1]);
  LAUNCH_STREAMING(k_phase, T, kUPhase, count, ins, st, count, ins, sp.low, value);
  HIPCHK(hipGetLastError());
  return QIP_OK;
}

static DiagDesc make_diagdesc(const Plan& p) {
  DiagDesc d;
  memset(&d, 0, sizeof d);
  d.k = (uint32_t)p.opos.size();
  for (uint32_t j = 0; j < d.k && j < 32; ++j) d.tpos[j] = p.opos[j];
  return d;
}

template <typename T, typename E>
static int launch_diag(qip_hip_state* s, uint32_t n, const Plan& p, E* st, int* actual_cls) {
  const Split sp = split_selectors(p.cpos, mask_of(p.cpos));
  Ins ins = make_ins(sp.hi, sp.hi_ones);
  const uint64_t count = 1ull << (n - (uint32_t)sp.hi.size());
  if (p.opos.size() == 1) {  // Rz-like: no table, factor picked by the target bit
    *actual_cls = KC_DIAG1Q;
    const uint64_t tmask = 1ull << p.opos[0];
    const amp_t<T> d0 = mk<T>(p.table[0], p.table[1]), d1 = mk<T>(p.table[2], p.table[3]);
    LAUNCH_STREAMING(k_diag1q, T, kUPhase, count, ins, st, count, ins, tmask, sp.low, d0, d1);
    HIPCHK(hipGetLastError());
    return QIP_OK;
  }
  QCHK(upload_table<T>(s, p.table));
  const DiagDesc dd = make_diagdesc(p);
  const amp_t<T>* table = (const amp_t<T>*)s->arena;
  LAUNCH_STREAMING(k_diag, T, kUPhase, count, ins, st, count, ins, sp.low, dd, table);
  HIPCHK(hipGetLastError());
  return QIP_OK;
}

// A group of >= 2 transpositions (pa pb), pa < pb, in ONE sweep (k_swapn).  `done` = false when the state is too
// small for the shape (fewer work items than lanes): the caller then applies them one at a time.
struct SwPair {
  uint32_t pa, pb;
};
static inline int swap_pair_regbits(const Split& sp, const SwPair& q) {  // HH: 2, HL: 1, LL: 0 (pa < pb)
  return (work_bit(q.pa, sp.hi) < 6 ? 0 : 1) + (work_bit(q.pb, sp.hi) < 6 ? 0 : 1);
}

template <typename T, typename E>
static int launch_swapn(qip_hip_state* s, uint32_t n, const Split& sp, const std::vector<SwPair>& grp, E* st, bool* done) {
  *done = false;
  SwapNDesc d;
  memset(&d, 0, sizeof d);
  std::vector<uint32_t> pos = sp.hi, regpos;
  // register bits: the HL bits first, then the HH pairs (bits 2j, 2j+1 of what follows) — the order k_swapn assumes
  for (const SwPair& q : grp)
    if (work_bit(q.pa, sp.hi) < 6 && work_bit(q.pb, sp.hi) >= 6) {
      d.hl_lane[regpos.size()] = work_bit(q.pa, sp.hi);
      regpos.push_back(q.pb);
    }
  const uint32_t NHL = (uint32_t)regpos.size();
  for (const SwPair& q : grp) {
    const bool la = work_bit(q.pa, sp.hi) < 6, lb = work_bit(q.pb, sp.hi) < 6;  // a lane-bit pb implies a lane-bit pa
    if (la && lb) {
      d.ll_a[d.n_ll] = work_bit(q.pa, sp.hi);
      d.ll_b[d.n_ll++] = work_bit(q.pb, sp.hi);
    } else if (!la) {
      regpos.push_back(q.pa);
      regpos.push_back(q.pb);
    }
  }
  const uint32_t NH = (uint32_t)regpos.size();
  if (NH > 4 || d.n_ll > 4) return fail(QIP_ERR_UNSUPPORTED, "swap group too large (internal error)");
  for (uint32_t p : regpos) pos.push_back(p);
  const uint64_t nsub = 1ull << (n - (uint32_t)sp.hi.size());
  const uint64_t nitems = nsub >> NH;
  if (nitems < 64) return QIP_OK;  // fewer items than lanes: the lane-bit classification does not hold
  for (uint32_t c = 0; c < (1u << NH); ++c)
    for (uint32_t r = 0; r < NH; ++r)
      if ((c >> r) & 1u) d.off_ld[c] |= 1ull << regpos[r];
  for (uint32_t c = 0; c < (1u << NH); ++c) {
    uint32_t pc = c;
    for (uint32_t r = NHL; r + 1 < NH; r += 2) {  // HH pair on register bits (r, r + 1)
      const uint32_t b0 = (pc >> r) & 1u, b1 = (pc >> (r + 1)) & 1u;
      pc = (pc & ~((1u << r) | (1u << (r + 1)))) | (b1 << r) | (b0 << (r + 1));
    }
    d.off_st[c] = d.off_ld[pc];
  }
  Ins ins = make_ins(pos, sp.hi_ones);
  const bool big = use_nt(s);  // >= 1 GiB states: unguarded, several groups per lane, non-temporal; else one guarded group
#define SWN(NHV, NHLV, LLV, UU)                                                                                             \
  do {                                                                                                                      \
    if (big && nitems >= ((uint64_t)(UU) << kStrideShift))                                                                  \
      hipLaunchKernelGGL((k_swapn<T, NHV, NHLV, LLV, UU, false, true, E>), grid2d(nitems, kBlock * (UU)), dim3(kBlock), 0, s->stream, st, nitems, ins, d, sp.low); \
    else                                                                                                                    \
      hipLaunchKernelGGL((k_swapn<T, NHV, NHLV, LLV, 1, true, false, E>), grid2d(nitems, kBlock), dim3(kBlock), 0, s->stream, st, nitems, ins, d, sp.low);         \
  } while (0)
  const bool ll = d.n_ll > 0;
  const uint32_t code = NH * 16 + NHL * 2 + (ll ? 1 : 0);
  switch (code) {
    case 0 * 16 + 0 * 2 + 1: SWN(0, 0, true, 4); break;
    case 1 * 16 + 1 * 2 + 0: SWN(1, 1, false, 4); break;
    case 1 * 16 + 1 * 2 + 1: SWN(1, 1, true, 4); break;
    case 2 * 16 + 0 * 2 + 1: SWN(2, 0, true, 2); break;
    case 2 * 16 + 2 * 2 + 0: SWN(2, 2, false, 2); break;
    case 2 * 16 + 2 * 2 + 1: SWN(2, 2, true, 2); break;
    case 3 * 16 + 1 * 2 + 0: if (s->unroll == 2) SWN(3, 1, false, 2); else SWN(3, 1, false, 1); break;
    case 3 * 16 + 1 * 2 + 1: SWN(3, 1, true, 1); break;
    case 3 * 16 + 3 * 2 + 0: SWN(3, 3, false, 1); break;
    case 3 * 16 + 3 * 2 + 1: SWN(3, 3, true, 1); break;
    case 4 * 16 + 0 * 2 + 0: SWN(4, 0, false, 1); break;
    case 4 * 16 + 0 * 2 + 1: SWN(4, 0, true, 1); break;
    case 4 * 16 + 2 * 2 + 0: SWN(4, 2, false, 1); break;
    case 4 * 16 + 2 * 2 + 1: SWN(4, 2, true, 1); break;
    case 4 * 16 + 4 * 2 + 0: SWN(4, 4, false, 1); break;
    case 4 * 16 + 4 * 2 + 1: SWN(4, 4, true, 1); break;
    default: return QIP_OK;  // (2, 0, no LL) is a single HH transposition: the caller's one-at-a-time kernels
  }
#undef SWN
  HIPCHK(hipGetLastError());
  *done = true;
  return QIP_OK;
}

template <typename T, typename E>
static int launch_swap(qip_hip_state* s, uint32_t n, const Plan& p, E* st) {
  // Swap(h, A ++ B) = product of the h disjoint transpositions (A[j] B[j]); moves are exact, so applying them in
  // groups is bit-identical to the single permutation.  Groups hold as many transpositions as fit 4 register bits
  // (16 amplitudes per lane) and go in ONE sweep each (k_swapn); a group of one uses the single-transposition kernels.
  const uint32_t h = (uint32_t)p.opos.size() / 2;
  const Split sp = split_selectors(p.cpos, mask_of(p.cpos));
  std::vector<std::vector<SwPair>> groups;
  {
    std::vector<SwPair> cur;
    int regs = 0, lls = 0;
    for (uint32_t j = 0; j < h; ++j) {
      SwPair q{p.opos[j], p.opos[h + j]};
      if (q.pa > q.pb) std::swap(q.pa, q.pb);
      const int need = swap_pair_regbits(sp, q);
      if (!cur.empty() && (s->swap_single || regs + need > 4 || (need == 0 && lls == 4))) {
        groups.push_back(cur);
        cur.clear();
        regs = lls = 0;
      }
      cur.push_back(q);
      regs += need;
      lls += need == 0;
    }
    if (!cur.empty()) groups.push_back(cur);
  }
  for (const std::vector<SwPair>& grp : groups) {
    if (grp.size() >= 2) {
      bool done = false;
      QCHK((launch_swapn<T, E>(s, n, sp, grp, st, &done)));
      if (done) continue;
    }
    for (const SwPair& q : grp) {
      const uint32_t pa = q.pa, pb = q.pb;
      const uint32_t wa = work_bit(pa, sp.hi), wb = work_bit(pb, sp.hi);
      const uint64_t nsub = 1ull << (n - (uint32_t)sp.hi.size());
      if (wa < 6 && wb < 6 && nsub >= 64) {  // both inside the lane index: one row, lane permutation
        Ins ins = make_ins(sp.hi, sp.hi_ones);
        LAUNCH_STREAMING(k_swap_xlane1, T, kUXlane, nsub, ins, st, nsub, ins, wa, wb, sp.low);
      } else if (wa < 6 && nsub >= 128) {  // low bit in the lane index, high bit picks the row
        std::vector<uint32_t> pos = sp.hi;
        pos.push_back(pb);
        Ins ins = make_ins(pos, sp.hi_ones);
        const uint64_t nitems = nsub >> 1;
        const uint64_t hmask = 1ull << pb;
        // wa is unchanged by opening pb (pb > pa)
        LAUNCH_STREAMING(k_swap_xlane2, T, kUSwap, nitems, ins, st, nitems, ins, wa, hmask, sp.low);
      } else {
        std::vector<uint32_t> pos = sp.hi;
        pos.push_back(pa);
        pos.push_back(pb);
        Ins ins = make_ins(pos, sp.hi_ones);
        const uint64_t npairs = nsub >> 2;
        const uint64_t amask = 1ull << pa, bmask = 1ull << pb;
        LAUNCH_STREAMING(k_swap_bits, T, kUSwap, npairs, ins, st, npairs, ins, amask, bmask, sp.low);
      }
      HIPCHK(hipGetLastError());
    }
  }
  return QIP_OK;
}

// ---- any permutation of the index bits in one out-of-place sweep (k_permute_bits) ---------------------------------
// pi[d] = source bit position that destination bit position d takes its value from: out[j] = in[src(j)], bit pi[d] of
// src(j) = bit d of j.  Pure host code; exported through qip_hip_debug_permute_plan for the CPU tests.
static int make_perm_desc(uint32_t n, const uint32_t* pi, uint32_t R, uint32_t fold_bits, PermDesc* out) {
  const uint32_t TB = 2 * R;
  if (n < TB || TB > (uint32_t)kPermMaxTile) return fail(QIP_ERR_INVALID, "internal: permutation tile does not fit n = %u", n);
  PermDesc& d = *out;
  memset(&d, 0, sizeof d);
  std::vector<char> in_tile(n, 0);
  for (uint32_t b = 0; b < R; ++b) in_tile[b] = 1;             // the destination's row bits
  for (uint32_t b = 0; b < n; ++b)
    if (pi[b] < R) in_tile[b] = 1;                              // destination bits fed by the source's row bits
  uint32_t cnt = 0;
  for (uint32_t b = 0; b < n; ++b) cnt += in_tile[b];
  for (uint32_t b = 0; b < n && cnt < TB; ++b)                  // pad with the lowest positions left
    if (!in_tile[b]) {
      in_tile[b] = 1;
      ++cnt;
    }
  std::vector<uint32_t> tb, sb;
  for (uint32_t b = 0; b < n; ++b)
    if (in_tile[b]) {
      tb.push_back(b);
      sb.push_back(pi[b]);
    }
  std::sort(sb.begin(), sb.end());
  for (uint32_t i = 0; i < TB; ++i) {
    d.tbits[i] = tb[i];
    d.sbits[i] = sb[i];
  }
  for (uint32_t i = 0; i < TB; ++i) {  // source coordinate bit i is source position sb[i] = pi[tb[k]] -> tile bit k
    uint32_t k = 0;
    while (k < TB && pi[tb[k]] != sb[i]) ++k;
    if (k == TB) return fail(QIP_ERR_INVALID, "internal: permutation tile is not closed");
    d.u2c[i] = k;
  }
  // LDS swizzle: the source-side lanes of one bank group vary tile bits u2c[0..FB-1], the destination-side lanes tile
  // bits 0..FB-1; fold every u2c[i] >= FB into a low bit no u2c[j] < FB occupies, so both sides spread over all banks
  std::vector<char> taken(fold_bits, 0);
  for (uint32_t i = 0; i < fold_bits; ++i)
    if (d.u2c[i] < fold_bits) taken[d.u2c[i]] = 1;
  uint32_t slot = 0;
  for (uint32_t i = 0; i < fold_bits; ++i) {
    if (d.u2c[i] < fold_bits) continue;
    while (taken[slot]) ++slot;
    taken[slot] = 1;
    d.fold_from[d.nfold] = d.u2c[i];
    d.fold_to[d.nfold] = slot;
    d.nfold += 1;
  }
  for (uint32_t b = 0; b < n; ++b)
    if (!in_tile[b]) {
      d.outer_dst[d.n_outer] = (unsigned char)b;
      d.outer_src[d.n_outer] = (unsigned char)pi[b];
      d.n_outer += 1;
    }
  return QIP_OK;
}

static int check_bit_permutation(uint32_t n, const uint32_t* pi, bool* identity) {
  uint64_t seen = 0;
  *identity = true;
  for (uint32_t b = 0; b < n; ++b) {
    if (pi[b] >= n || ((seen >> pi[b]) & 1ull)) return fail(QIP_ERR_INVALID, "not a permutation of the %u index bits", n);
    seen |= 1ull << pi[b];
    if (pi[b] != b) *identity = false;
  }
  return QIP_OK;
}

// cur -> alt through the permutation, then alt becomes the current buffer (builder.rs:514 analogue)
static int launch_permute(qip_hip_state* s, const uint32_t* pi_in) {
  bool identity = true;
  QCHK(check_bit_permutation(s->n, pi_in, &identity));
  if (identity) return QIP_OK;
  QCHK(ensure_alt(s));
  std::vector<uint32_t> pi(pi_in, pi_in + s->n);
  uint32_t n = s->n;
  // Complex<f32>: two amplitudes per 16-byte element when index bit 0 stays where it is
  const bool packed = s->dtype == QIP_C32 && pi[0] == 0 && n >= 2 && s->packed_f32;
  if (packed) {
    for (uint32_t b = 0; b + 1 < n; ++b) pi[b] = pi[b + 1] - 1;
    n -= 1;
  }
  const bool wide = s->dtype == QIP_C64 || packed;   // 16-byte elements
  // 16-byte elements: 512-B rows in 16-KiB tiles (8 blocks per CU) unless two or more of the source's row bits feed
  // destination bits far above the rows — then the tile's rows are scattered on the source side and 1-KiB rows in
  // 64-KiB tiles pay (measured at n = 30: random permutation 7.2 -> 6.5 ms, three transpositions 7.3 -> 5.9, while a
  // single transposition is better off with the small tile: 5.9 vs 6.2 ms; profiles/r02_permute.md)
  uint32_t scattered = 0;
  for (uint32_t b = 10; b < n; ++b) scattered += pi[b] < 5u;
  const uint32_t R = !wide ? 6u : (g_perm_rows ? (uint32_t)g_perm_rows : (scattered >= 2 && n >= 12 ? 6u : 5u)), TB = 2 * R;
  ProfRec rec;
  if (s->profile) QCHK(prof_begin(s, KC_PERMUTE, 2.0 * (double)s->amp_bytes * (double)s->namps, &rec));
  const bool nt = use_nt(s);
  if (n < TB) {
    PermSmall ps;
    memset(&ps, 0, sizeof ps);
    ps.n = n;
    for (uint32_t b = 0; b < n; ++b) ps.pi[b] = (unsigned char)pi[b];
    const uint64_t count = 1ull << n;
    const dim3 grid(grid_for(count, kBlock)), block(kBlock);
    if (s->dtype == QIP_C64)
      hipLaunchKernelGGL((k_permute_bits_small<amp_t<double>>), grid, block, 0, s->stream, (const amp_t<double>*)s->cur, (amp_t<double>*)s->alt, count, ps);
    else if (packed)
      hipLaunchKernelGGL((k_permute_bits_small<f32x4>), grid, block, 0, s->stream, (const f32x4*)s->cur, (f32x4*)s->alt, count, ps);
    else
      hipLaunchKernelGGL((k_permute_bits_small<amp_t<float>>), grid, block, 0, s->stream, (const amp_t<float>*)s->cur, (amp_t<float>*)s->alt, count, ps);
  } else {
    PermDesc d;
    QCHK(make_perm_desc(n, pi.data(), R, wide ? 3u : 4u, &d));  // (fold width follows the element size, not R)
    const dim3 grid = grid2d(1ull << (n - TB), 1), block(kBlock);
#define PB(A, RR)                                                                                                        \
  do {                                                                                                                   \
    if (nt) hipLaunchKernelGGL((k_permute_bits<A, RR, true>), grid, block, 0, s->stream, (const A*)s->cur, (A*)s->alt, d);  \
    else hipLaunchKernelGGL((k_permute_bits<A, RR, false>), grid, block, 0, s->stream, (const A*)s->cur, (A*)s->alt, d);    \
  } while (0)
    if (s->dtype == QIP_C64 && R == 5) PB(amp_t<double>, 5);
    else if (s->dtype == QIP_C64) PB(amp_t<double>, 6);
    else if (packed && R == 5) PB(f32x4, 5);
    else if (packed) PB(f32x4, 6);
    else PB(amp_t<float>, 6);
#undef PB
  }
  HIPCHK(hipGetLastError());
  if (s->profile) QCHK(prof_end(s, &rec));
  std::swap(s->cur, s->alt);
  std::swap(s->owns_cur, s->owns_alt);
  return QIP_OK;
}

extern "C" int qip_hip_state_permute_bits(qip_hip_state* s, const uint32_t* pi) try {
  STATE_ENTER(s);
  if (!pi) return fail(QIP_ERR_INVALID, "null permutation");
  return launch_permute(s, pi);
} QIP_CATCH_ALL

// Host-only: the descriptor k_permute_bits would get, as JSON (tests/test_permute_plan_cpu.py replays the kernel's index
// arithmetic with numpy: every element lands where out[j] = in[src(j)] says, the LDS slots of a tile are a bijection,
// and the lanes of a bank group hit distinct banks).
extern "C" const char* qip_hip_debug_permute_plan(uint32_t n, const uint32_t* pi, uint32_t row_bits, uint32_t fold_bits) {
  static thread_local std::string json;
  try {
    bool identity = true;
    if (!pi || n == 0 || n > 62) return fail(QIP_ERR_INVALID, "bad argument"), nullptr;
    if (check_bit_permutation(n, pi, &identity) != QIP_OK) return nullptr;
    PermDesc d;
    if (make_perm_desc(n, pi, row_bits, fold_bits, &d) != QIP_OK) return nullptr;
    const uint32_t TB = 2 * row_bits;
    auto arr = [&](const char* key, const uint32_t* v, uint32_t cnt) {
      std::string a = std::string("\"") + key + "\":[";
      for (uint32_t i = 0; i < cnt; ++i) a += (i ? "," : "") + std::to_string(v[i]);
      return a + "]";
    };
    uint32_t od[64], os[64];
    for (uint32_t i = 0; i < d.n_outer; ++i) {
      od[i] = d.outer_dst[i];
      os[i] = d.outer_src[i];
    }
    json = "{\"n\":" + std::to_string(n) + ",\"row_bits\":" + std::to_string(row_bits) + "," + arr("tbits", d.tbits, TB) + "," +
           arr("sbits", d.sbits, TB) + "," + arr("u2c", d.u2c, TB) + "," + arr("fold_from", d.fold_from, d.nfold) + "," +
           arr("fold_to", d.fold_to, d.nfold) + "," + arr("outer_dst", od, d.n_outer) + "," + arr("outer_src", os, d.n_outer) + "}";
    return json.c_str();
  } catch (const std::exception& e) {
    fail(QIP_ERR_INVALID, "internal error: %s", e.what());
    return nullptr;
  }
}

template <typename T>
static int launch_gather(qip_hip_state* s, const FlatOp& f, const amp_t<T>* in, uint64_t in_len,
                         amp_t<T>* out, uint64_t out_len, uint64_t in_off, uint64_t out_off,
                         int accumulate);

// A operand of k_gate_kq_mfma, one double per (tile row block, K-step, lane); see the kernel header.
// f32_layout: the C/D rows of v_mfma_f32_16x16x4_f32 are 4 * (lane >> 4) + reg, those of the f64 form (lane >> 4) + 4 * reg.
static void build_afrag(const Plan& p, const std::vector<uint32_t>& tau, std::vector<double>* out, bool f32_layout = false) {
  const uint32_t k = (uint32_t)p.opos.size();
  const uint32_t S = 1u << k, TT = S / 8, KS = S / 2;
  uint32_t perm_bit[8];  // c~ bit b (b-th lowest target position) -> sub-index bit of the reference
  for (uint32_t b = 0; b < k; ++b)
    for (uint32_t j = 0; j < k; ++j)
      if (p.opos[j] == tau[b]) perm_bit[b] = k - 1 - j;
  auto c_of = [&](uint32_t ct) {
    uint32_t c = 0;
    for (uint32_t b = 0; b < k; ++b) c |= ((ct >> b) & 1u) << perm_bit[b];
    return c;
  };
  out->assign((size_t)TT * KS * 64, 0.0);
  for (uint32_t rb = 0; rb < TT; ++rb)
    for (uint32_t s = 0; s < KS; ++s)
      for (uint32_t l = 0; l < 64; ++l) {
        const uint32_t i = l & 15, kk = l >> 4;
        const uint32_t qp = f32_layout ? i >> 2 : i & 3, reg = f32_layout ? i & 3 : i >> 2, partp = reg & 1, t = reg >> 1;
        const uint32_t ctp = 4 * (2 * rb + t) + qp, ct = 4 * (s >> 1) + kk, part = s & 1;
        const size_t e = (size_t)c_of(ctp) * S + c_of(ct);
        const double re = p.table[2 * e], im = p.table[2 * e + 1];
        (*out)[((size_t)rb * KS + s) * 64 + l] = partp == 0 ? (part == 0 ? re : -im) : (part == 0 ? im : re);
      }
}

template <typename T>
static int launch_kq_mfma(qip_hip_state* s, const Plan& p, amp_t<T>* st) {
  const uint32_t k = (uint32_t)p.opos.size();
  std::vector<uint32_t> tau = p.opos;
  std::sort(tau.begin(), tau.end());
  std::vector<double> afrag;
  build_afrag(p, tau, &afrag, std::is_same<T, float>::value);
  std::vector<T> af_t(afrag.begin(), afrag.end());
  QCHK(arena_upload(s, af_t.data(), af_t.size() * sizeof(T), 0));
  std::vector<uint32_t> pos = p.cpos;
  for (uint32_t t : p.opos) pos.push_back(t);
  Ins ins = make_ins(pos, mask_of(p.cpos));
  MfmaDesc d;
  memset(&d, 0, sizeof d);
  for (uint32_t b = 0; b < k; ++b) d.tau[b] = tau[b];
  const uint64_t nitems = 1ull << (s->n - (uint32_t)pos.size() - 4);  // waves' worth of 16 groups
  const unsigned blocks = (unsigned)std::min<uint64_t>((nitems + 3) / 4, 256ull * 8);  // waves loop over items
  const dim3 grid(blocks), block(kBlock);
  const T* af = (const T*)s->arena;
  const bool nt = use_nt(s);
#define MF(K, WU)                                                                                            \
  do {                                                                                                       \
    if (nt) hipLaunchKernelGGL((k_gate_kq_mfma<T, K, WU, true>), grid, block, 0, s->stream, st, nitems, ins, d, af);  \
    else hipLaunchKernelGGL((k_gate_kq_mfma<T, K, WU, false>), grid, block, 0, s->stream, st, nitems, ins, d, af);    \
  } while (0)
  // WU items per iteration so that WU * 2^k / 4 = 8 loads are in flight per lane (nitems is a power of two)
  switch (k) {
    case 3: if (nitems >= 4 && s->unroll != 1) MF(3, 4); else MF(3, 1); break;
    case 4: if (nitems >= 2 && s->unroll == 2) MF(4, 2); else MF(4, 1); break;
    case 5: if (nitems >= 2 && s->unroll == 2) MF(5, 2); else MF(5, 1); break;
    default: return fail(QIP_ERR_UNSUPPORTED, "matrix-core kernel for k = %u", k);
  }
#undef MF
  HIPCHK(hipGetLastError());
  return QIP_OK;
}

// dense k = 6..8 on the matrix cores (f64 and f32 forms), A operand streamed through LDS (k_gate_big_mfma)
template <typename T>
static int launch_big_mfma(qip_hip_state* s, const Plan& p, amp_t<T>* st) {
  const uint32_t k = (uint32_t)p.opos.size();
  std::vector<uint32_t> tau = p.opos;
  std::sort(tau.begin(), tau.end());
  std::vector<double> afrag;
  build_afrag(p, tau, &afrag, std::is_same<T, float>::value);
  std::vector<T> af_t(afrag.begin(), afrag.end());
  QCHK(arena_upload(s, af_t.data(), af_t.size() * sizeof(T), 0));
  std::vector<uint32_t> pos = p.cpos;
  for (uint32_t t : p.opos) pos.push_back(t);
  Ins ins = make_ins(pos, mask_of(p.cpos));
  MfmaDesc d;
  memset(&d, 0, sizeof d);
  for (uint32_t b = 0; b < k; ++b) d.tau[b] = tau[b];
  const uint64_t nitems = 1ull << (s->n - (uint32_t)pos.size() - 4);  // waves' worth of 16 groups
  const unsigned per_cu = k <= 7 ? 2u : 1u;  // resident blocks per CU (registers: 2 waves per SIMD up to k = 7; LDS: 128 KiB at k = 8 in f64)
  const unsigned blocks = (unsigned)std::min<uint64_t>((nitems + 3) / 4, 256ull * per_cu);
  const dim3 grid(blocks), block(kBlock);
  const T* af = (const T*)s->arena;
  const bool nt = use_nt(s);
#define BM(K)                                                                                                        \
  do {                                                                                                               \
    if (nt) hipLaunchKernelGGL((k_gate_big_mfma<T, K, true>), grid, block, 0, s->stream, st, nitems, ins, d, af);    \
    else hipLaunchKernelGGL((k_gate_big_mfma<T, K, false>), grid, block, 0, s->stream, st, nitems, ins, d, af);      \
  } while (0)
  switch (k) {
    case 6: BM(6); break;
    case 7: BM(7); break;
    case 8: BM(8); break;
    default: return fail(QIP_ERR_UNSUPPORTED, "streamed matrix-core kernel for k = %u", k);
  }
#undef BM
  HIPCHK(hipGetLastError());
  return QIP_OK;
}

template <typename T>
static int launch_kq(qip_hip_state* s, const Plan& p, amp_t<T>* st, int* actual_cls, const FlatOp& f) {
  const uint32_t k = (uint32_t)p.opos.size();
  const uint32_t used = (uint32_t)(p.opos.size() + p.cpos.size());
  // how many target bits sit inside the lane index (a lane's 2^k accesses then share 1-KiB rows
  // with its neighbours only partially)
  uint32_t low_targets = 0, min_target = 64;
  for (uint32_t t : p.opos) {
    if (t < 6) ++low_targets;
    min_target = std::min(min_target, t);
  }
  {
    // matrix cores (f64 and f32 forms): always for k = 5 (no register form), and for k = 3, 4 when two or more targets
    // are low bit positions, where the MFMA mapping keeps 64-B+ runs per lane group and the per-lane
    // register form does not (measured at n = 30: profiles/r01_ops_table*.md)
    const bool want_mfma = k == 5 || (k >= 3 && low_targets >= 2) || s->mfma == 2;  // 2 = force (tuning aid)
    if (s->mfma && want_mfma && k >= 3 && k <= kMaxMfmaK && s->n >= used + 4) {
      *actual_cls = KC_GATE_KQ_MFMA;
      return launch_kq_mfma<T>(s, p, st);
    }
  }
  if (s->mfma && k > kMaxMfmaK && k <= kMaxBigK && s->n >= used + 4) {
    *actual_cls = KC_GATE_KQ_BIG;
    return launch_big_mfma<T>(s, p, st);
  }
  if (k > kMaxRegK) {  // no register form: literal kernel, out of place
    *actual_cls = KC_GATHER_GENERIC;
    QCHK(ensure_alt(s));
    QCHK(launch_gather<T>(s, f, (const amp_t<T>*)s->cur, s->namps, (amp_t<T>*)s->alt, s->namps, 0, 0, 0));
    std::swap(s->cur, s->alt);
    std::swap(s->owns_cur, s->owns_alt);
    return QIP_OK;
  }
  QCHK(upload_table<T>(s, p.table));
  std::vector<uint32_t> pos = p.cpos;
  for (uint32_t t : p.opos) pos.push_back(t);
  Ins ins = make_ins(pos, mask_of(p.cpos));
  const uint64_t groups = 1ull << (s->n - (uint32_t)pos.size());
  const DiagDesc d = make_diagdesc(p);
  const amp_t<T>* mat = (const amp_t<T>*)s->arena;
  const bool nt = use_nt(s) && min_target >= 6;
  // One group (2^k amplitudes) per lane.  Several groups per lane with the 32-KiB spacing that pays off
  // for the 1-qubit kernels were measured at n = 30 and do NOT help here (k = 2: 5.82 vs 5.85 TB/s, k = 3:
  // 5.52 vs 5.61): option unroll = 2 still selects them for experiments.
#define KQ(K, UU)                                                                                       \
  do {                                                                                                  \
    if (groups >= ((uint64_t)(UU) << kStrideShift) && s->unroll == 2 && (UU) > 1) {                     \
      const dim3 grid = grid2d(groups, kBlock * (UU));                                                  \
      if (nt) hipLaunchKernelGGL((k_gate_kq<T, K, UU, false, true>), grid, dim3(kBlock), 0, s->stream, st, groups, ins, d, mat);  \
      else hipLaunchKernelGGL((k_gate_kq<T, K, UU, false, false>), grid, dim3(kBlock), 0, s->stream, st, groups, ins, d, mat);    \
    } else {                                                                                            \
      const dim3 grid = grid2d(groups, kBlock);                                                         \
      if (nt) hipLaunchKernelGGL((k_gate_kq<T, K, 1, true, true>), grid, dim3(kBlock), 0, s->stream, st, groups, ins, d, mat);    \
      else hipLaunchKernelGGL((k_gate_kq<T, K, 1, true, false>), grid, dim3(kBlock), 0, s->stream, st, groups, ins, d, mat);      \
    }                                                                                                   \
  } while (0)
  switch (k) {
    case 2: KQ(2, 4); break;
    case 3: KQ(3, 2); break;
    case 4: KQ(4, 1); break;
    default: return fail(QIP_ERR_UNSUPPORTED, "register kernel for k = %u", k);
  }
#undef KQ
  HIPCHK(hipGetLastError());
  return QIP_OK;
}

// Ship the inner op's payload to the arena and run the literal gather kernel in -> out.
template <typename T>
static int launch_gather(qip_hip_state* s, const FlatOp& f, const amp_t<T>* in, uint64_t in_len,
                         amp_t<T>* out, uint64_t out_len, uint64_t in_off, uint64_t out_off,
                         int accumulate) {
  GatherDesc d;
  memset(&d, 0, sizeof d);
  d.n = s->n;
  d.k_all = f.k_all;
  d.n_control = f.n_control;
  d.n_op = f.n_op;
  d.inner_kind = f.inner->kind;
  d.accumulate = accumulate;
  d.in_len = in_len;
  d.out_len = out_len;
  d.in_off = in_off;
  d.out_off = out_off;
  for (uint32_t j = 0; j < f.k_all; ++j) d.pos[j] = (uint32_t)(s->n - 1 - f.outer->indices[j]);
  const amp_t<T>* dense = nullptr;
  const uint64_t* rowptr = nullptr;
  const uint64_t* cols = nullptr;
  const amp_t<T>* vals = nullptr;
  if (f.inner->kind == QIP_OP_MATRIX) {
    const size_t bytes = (sizeof(amp_t<T>) << (2 * f.n_op));
    QCHK(arena_upload(s, f.inner->dense, bytes, 0));
    dense = (const amp_t<T>*)s->arena;
  } else if (f.inner->kind == QIP_OP_SPARSE) {
    const uint64_t rows = 1ull << f.n_op;
    const uint64_t nnz = f.inner->sparse_rowptr[rows];
    const size_t b_rp = (rows + 1) * 8, b_cols = nnz * 8, b_vals = nnz * sizeof(amp_t<T>);
    const size_t o_cols = (b_rp + 15) & ~(size_t)15, o_vals = (o_cols + b_cols + 15) & ~(size_t)15;
    QCHK(ensure_arena(s, o_vals + b_vals + 16));
    QCHK(arena_upload(s, f.inner->sparse_rowptr, b_rp, 0));
    if (nnz) {
      QCHK(arena_upload(s, f.inner->sparse_cols, b_cols, o_cols));
      QCHK(arena_upload(s, f.inner->sparse_vals, b_vals, o_vals));
    }
    rowptr = (const uint64_t*)s->arena;
    cols = (const uint64_t*)((char*)s->arena + o_cols);
    vals = (const amp_t<T>*)((char*)s->arena + o_vals);
  }
  hipLaunchKernelGGL((k_gather_generic<T>), dim3(grid_stride(out_len)), dim3(kBlock), 0, s->stream, in,
                     out, d, dense, rowptr, cols, vals);
  HIPCHK(hipGetLastError());
  return QIP_OK;
}

// SparseMatrix (optionally controlled) on k <= 5 distinct qubits, in place (k_sparse_kq)
template <typename T>
static int launch_sparse_kq(qip_hip_state* s, const Plan& p, const FlatOp& f, amp_t<T>* st) {
  const uint32_t k = f.n_op;
  const uint64_t rows = 1ull << k;
  const uint64_t nnz = f.inner->sparse_rowptr[rows];
  const size_t b_rp = (rows + 1) * 8, b_cols = nnz * 8, b_vals = nnz * sizeof(amp_t<T>);
  const size_t o_cols = (b_rp + 15) & ~(size_t)15, o_vals = (o_cols + b_cols + 15) & ~(size_t)15;
  QCHK(ensure_arena(s, o_vals + b_vals + 16));  // one allocation: growing frees the old arena
  QCHK(arena_upload(s, f.inner->sparse_rowptr, b_rp, 0));
  if (nnz) {
    QCHK(arena_upload(s, f.inner->sparse_cols, b_cols, o_cols));
    QCHK(arena_upload(s, f.inner->sparse_vals, b_vals, o_vals));
  }
  const uint64_t* rowptr = (const uint64_t*)s->arena;
  const uint64_t* cols = (const uint64_t*)((char*)s->arena + o_cols);
  const amp_t<T>* vals = (const amp_t<T>*)((char*)s->arena + o_vals);
  std::vector<uint32_t> pos = p.cpos;
  uint32_t min_target = 64;
  for (uint32_t t : p.opos) {
    pos.push_back(t);
    min_target = std::min(min_target, t);
  }
  Ins ins = make_ins(pos, mask_of(p.cpos));
  const uint64_t groups = 1ull << (s->n - (uint32_t)pos.size());
  const DiagDesc d = make_diagdesc(p);
  const bool nt = use_nt(s) && min_target >= 6;
  const unsigned threads = k <= 3 ? 256u : (k == 4 ? 128u : 64u);
  const size_t lds = (sizeof(amp_t<T>) << k) * threads;
  const dim3 grid = grid2d(groups, threads);
#define SPK(K)                                                                                                          \
  do {                                                                                                                  \
    if (nt) hipLaunchKernelGGL((k_sparse_kq<T, K, true>), grid, dim3(threads), lds, s->stream, st, groups, ins, d, rowptr, cols, vals);  \
    else hipLaunchKernelGGL((k_sparse_kq<T, K, false>), grid, dim3(threads), lds, s->stream, st, groups, ins, d, rowptr, cols, vals);    \
  } while (0)
  switch (k) {
    case 1: SPK(1); break;
    case 2: SPK(2); break;
    case 3: SPK(3); break;
    case 4: SPK(4); break;
    case 5: SPK(5); break;
    default: return fail(QIP_ERR_UNSUPPORTED, "in-place sparse kernel for k = %u", k);
  }
#undef SPK
  HIPCHK(hipGetLastError());
  return QIP_OK;
}

template <typename T>
static int apply_op_t(qip_hip_state* s, const qip_op* op) {
  if (s->jit_prepare) return QIP_OK;  // compiling a program's segment kernels: single ops have nothing to prepare
  (void)hipGetLastError();  // a stale error of an unrelated earlier call must not be blamed on this launch
  FlatOp f;
  QCHK(flatten_op(s->n, op, false, &f));
  Plan p;
  QCHK(make_plan(s->dtype, s->n, f, s->force_generic || g_force_generic, &p));
  if (p.cls == KC_NOOP) {
    if (s->profile) s->prof_launches[KC_NOOP] += 1;
    return QIP_OK;
  }
  ProfRec rec;
  rec.cls = p.cls;
  if (s->profile) QCHK(prof_begin(s, p.cls, p.alg_bytes, &rec));
  amp_t<T>* st = (amp_t<T>*)s->cur;
  int rc = QIP_OK;
  if constexpr (std::is_same<T, float>::value) {
    // packed f32 view: 2^(n-1) elements of two amplitudes each, bit positions shifted down by one
    const bool streaming = p.cls == KC_GATE1Q_PAIR || p.cls == KC_PHASE || p.cls == KC_DIAG || p.cls == KC_SWAP_BITS;
    bool bit0_selector = false, bit0_target = false;
    for (uint32_t c : p.cpos) bit0_selector |= c == 0;
    for (uint32_t t : p.opos) bit0_target |= t == 0;
    if (streaming && s->packed_f32 && s->n >= 2 && !bit0_selector &&
        (!bit0_target || p.cls == KC_GATE1Q_PAIR)) {
      Plan q = p;
      for (auto& c : q.cpos) c -= 1;
      f32x4* pst = (f32x4*)s->cur;
      const uint32_t ne = s->n - 1;
      using E = f32x4;
      if (bit0_target) {  // the pair is the two halves of one element
        Mat2<float> g;
        for (int e = 0; e < 4; ++e) g.m[e] = mk<float>(p.m[2 * e], p.m[2 * e + 1]);
        g.nz = p.nz;
        const Split sp = split_selectors(q.cpos, mask_of(q.cpos));
        Ins ins = make_ins(sp.hi, sp.hi_ones);
        const uint64_t count = 1ull << (ne - (uint32_t)sp.hi.size());
        rec.cls = KC_GATE1Q_XLANE;
        dispatch_np(ins.npos, [&](auto np_) {
          constexpr int NP = decltype(np_)::value;
          if (count >= ((uint64_t)kUXlane << kStrideShift)) {
            if (use_nt(s))
              hipLaunchKernelGGL((k_gate1q_inelem<kUXlane, false, true, NP>), dim3(grid_for(count, kBlock * kUXlane)),
                                 dim3(kBlock), 0, s->stream, pst, count, ins, sp.low, g);
            else
              hipLaunchKernelGGL((k_gate1q_inelem<kUXlane, false, false, NP>), dim3(grid_for(count, kBlock * kUXlane)),
                                 dim3(kBlock), 0, s->stream, pst, count, ins, sp.low, g);
          } else {
            hipLaunchKernelGGL((k_gate1q_inelem<1, true, false, NP>), dim3(grid_for(count, kBlock)), dim3(kBlock), 0,
                               s->stream, pst, count, ins, sp.low, g);
          }
        });
        HIPCHK(hipGetLastError());
      } else {
        for (auto& t : q.opos) t -= 1;
        switch (p.cls) {
          case KC_GATE1Q_PAIR: rc = launch_gate1q<float, E>(s, ne, q, pst, &rec.cls); break;
          case KC_PHASE: rc = launch_phase<float, E>(s, ne, q, pst); break;
          case KC_DIAG: rc = launch_diag<float, E>(s, ne, q, pst, &rec.cls); break;
          default: rc = launch_swap<float, E>(s, ne, q, pst); break;
        }
      }
      if (rc != QIP_OK) return rc;
      if (s->profile) QCHK(prof_end(s, &rec));
      return QIP_OK;
    }
  }
  switch (p.cls) {
    case KC_GATE1Q_PAIR: rc = launch_gate1q<T, amp_t<T>>(s, s->n, p, st, &rec.cls); break;
    case KC_PHASE: rc = launch_phase<T, amp_t<T>>(s, s->n, p, st); break;
    case KC_DIAG: rc = launch_diag<T, amp_t<T>>(s, s->n, p, st, &rec.cls); break;
    case KC_SWAP_BITS: rc = launch_swap<T, amp_t<T>>(s, s->n, p, st); break;
    case KC_GATE_KQ: rc = launch_kq<T>(s, p, st, &rec.cls, f); break;
    case KC_SPARSE_KQ: rc = launch_sparse_kq<T>(s, p, f, st); break;
    default: {
      QCHK(ensure_alt(s));
      rc = launch_gather<T>(s, f, (const amp_t<T>*)s->cur, s->namps, (amp_t<T>*)s->alt, s->namps, 0,
                            0, 0);
      if (rc == QIP_OK) {
        std::swap(s->cur, s->alt);  // builder.rs:514
        std::swap(s->owns_cur, s->owns_alt);
      }
      break;
    }
  }
  if (rc != QIP_OK) return rc;
  if (s->profile) QCHK(prof_end(s, &rec));
  return QIP_OK;
}

extern "C" int qip_hip_state_apply_op(qip_hip_state* s, const qip_op* op) try {
  STATE_ENTER(s);
  return s->dtype == QIP_C64 ? apply_op_t<double>(s, op) : apply_op_t<float>(s, op);
} QIP_CATCH_ALL

// ---------------------------------------------------------------------------------------
// gate fusion: option "fuse" = K merges consecutive small gates into dense
// gates on <= K qubits, applied in ONE sweep each.  The reference has no analogue (its apply_ops
// multi-op path is unused and inconsistent); this is pure host bookkeeping on
// 2^K x 2^K matrices.  Results equal the gate-by-gate path up to the rounding of the matrix
// products (|delta| ~ 1e-15 per fused gate), so fused runs are held to the 1e-12 bar, never to
// bit equality.  Open clusters always act on pairwise disjoint qubit sets, so they commute and
// may be flushed in any order; an op is merged only into clusters it overlaps, or into a
// disjoint one (which also commutes with every other open cluster).
// ---------------------------------------------------------------------------------------
typedef std::complex<double> cd;
struct Cluster {
  std::vector<uint32_t> pos;  // bit positions, descending: pos[0] is the MSB of the sub-index
  std::vector<cd> m;          // 2^q x 2^q, row-major
};

// dense matrix of a flattened op over its own index list (controls first), MSB-first order
template <typename T>
static bool op_to_dense(uint32_t n, const FlatOp& f, uint32_t max_k, std::vector<uint32_t>* pos,
                        std::vector<cd>* mat) {
  if (!f.distinct || f.k_all > max_k) return false;
  const uint32_t kt = f.k_all, k = f.n_op;
  const size_t S = (size_t)1 << kt, Si = (size_t)1 << k, thr = S - Si;
  pos->clear();
  for (uint32_t j = 0; j < kt; ++j) pos->push_back(n - 1 - (uint32_t)f.outer->indices[j]);
  mat->assign(S * S, cd(0, 0));
  for (size_t r = 0; r < thr; ++r) (*mat)[r * S + r] = cd(1, 0);
  const T* dense = static_cast<const T*>(f.inner->dense);
  const T* vals = static_cast<const T*>(f.inner->sparse_vals);
  for (size_t r = 0; r < Si; ++r) {
    switch (f.inner->kind) {
      case QIP_OP_MATRIX:
        for (size_t c = 0; c < Si; ++c)
          (*mat)[(thr + r) * S + thr + c] = cd(dense[2 * (r * Si + c)], dense[2 * (r * Si + c) + 1]);
        break;
      case QIP_OP_SPARSE:
        for (uint64_t e = f.inner->sparse_rowptr[r]; e < f.inner->sparse_rowptr[r + 1]; ++e)
          (*mat)[(thr + r) * S + thr + f.inner->sparse_cols[e]] += cd(vals[2 * e], vals[2 * e + 1]);
        break;
      default: {  // SWAP
        const uint32_t h = k >> 1;
        const size_t c = ((r & (((size_t)1 << h) - 1)) << h) + (r >> h);
        (*mat)[(thr + r) * S + thr + c] = cd(1, 0);
      }
    }
  }
  return true;
}

// matrix of `m` (over positions `pos`, MSB first) embedded into the space of `upos` (descending)
static std::vector<cd> embed(const std::vector<cd>& m, const std::vector<uint32_t>& pos,
                             const std::vector<uint32_t>& upos) {
  const uint32_t u = (uint32_t)upos.size(), k = (uint32_t)pos.size();
  const size_t U = (size_t)1 << u, S = (size_t)1 << k;
  std::vector<uint32_t> bit_in_u(k);
  size_t opmask = 0;
  for (uint32_t i = 0; i < k; ++i)
    for (uint32_t j = 0; j < u; ++j)
      if (upos[j] == pos[i]) {
        bit_in_u[i] = u - 1 - j;
        opmask |= (size_t)1 << (u - 1 - j);
      }
  auto sub = [&](size_t r) {
    size_t s_ = 0;
    for (uint32_t i = 0; i < k; ++i) s_ |= ((r >> bit_in_u[i]) & 1) << (k - 1 - i);
    return s_;
  };
  std::vector<cd> e(U * U, cd(0, 0));
  for (size_t r = 0; r < U; ++r)
    for (size_t c = 0; c < U; ++c)
      if ((r & ~opmask) == (c & ~opmask)) e[r * U + c] = m[sub(r) * S + sub(c)];
  return e;
}

static void cluster_apply(Cluster* cl, const std::vector<uint32_t>& pos, const std::vector<cd>& m) {
  std::vector<uint32_t> upos = cl->pos;
  for (uint32_t p : pos)
    if (std::find(upos.begin(), upos.end(), p) == upos.end()) upos.push_back(p);
  std::sort(upos.begin(), upos.end(), std::greater<uint32_t>());
  const size_t U = (size_t)1 << upos.size();
  const std::vector<cd> a = embed(m, pos, upos);              // the new gate, applied after
  const std::vector<cd> b = embed(cl->m, cl->pos, upos);      // what the cluster already holds
  std::vector<cd> out(U * U, cd(0, 0));
  for (size_t r = 0; r < U; ++r)
    for (size_t k = 0; k < U; ++k) {
      const cd v = a[r * U + k];
      if (v == cd(0, 0)) continue;
      for (size_t c = 0; c < U; ++c) out[r * U + c] += v * b[k * U + c];
    }
  cl->pos = upos;
  cl->m = out;
}

template <typename T>
static int flush_cluster(qip_hip_state* s, const Cluster& cl) {
  const size_t S = (size_t)1 << cl.pos.size();
  std::vector<uint64_t> idx;
  for (uint32_t p : cl.pos) idx.push_back(s->n - 1 - p);
  std::vector<T> data(2 * S * S);
  for (size_t e = 0; e < S * S; ++e) {
    data[2 * e] = (T)cl.m[e].real();
    data[2 * e + 1] = (T)cl.m[e].imag();
  }
  qip_op op;
  memset(&op, 0, sizeof op);
  op.kind = QIP_OP_MATRIX;
  op.n_indices = (uint32_t)idx.size();
  op.indices = idx.data();
  op.dense = data.data();
  return apply_op_t<T>(s, &op);
}

template <typename T>
static int apply_ops_fused(qip_hip_state* s, const qip_op* ops, uint64_t count, uint32_t K) {
  std::vector<Cluster> open;
  auto overlaps = [](const Cluster& c, const std::vector<uint32_t>& pos) {
    for (uint32_t p : pos)
      if (std::find(c.pos.begin(), c.pos.end(), p) != c.pos.end()) return true;
    return false;
  };
  auto flush_overlapping = [&](const std::vector<uint32_t>& pos) -> int {
    for (size_t i = 0; i < open.size();) {
      if (overlaps(open[i], pos)) {
        QCHK(flush_cluster<T>(s, open[i]));
        open.erase(open.begin() + i);
      } else {
        ++i;
      }
    }
    return QIP_OK;
  };
  for (uint64_t i = 0; i < count; ++i) {
    FlatOp f;
    int rc = flatten_op(s->n, &ops[i], false, &f);
    if (rc != QIP_OK) {
      std::string msg = g_last_error;
      return fail(rc, "op %llu: %s", (unsigned long long)i, msg.c_str());
    }
    std::vector<uint32_t> pos;
    std::vector<cd> m;
    if (!op_to_dense<T>(s->n, f, K, &pos, &m)) {
      // not fusable (too many qubits / repeated indices): everything it touches goes first
      std::vector<uint32_t> all;
      for (uint32_t j = 0; j < f.k_all; ++j) all.push_back(s->n - 1 - (uint32_t)f.outer->indices[j]);
      QCHK(flush_overlapping(all));
      QCHK(apply_op_t<T>(s, &ops[i]));
      continue;
    }
    std::vector<size_t> hit;
    size_t union_size = pos.size();
    for (size_t c = 0; c < open.size(); ++c)
      if (overlaps(open[c], pos)) {
        hit.push_back(c);
        for (uint32_t p : open[c].pos)
          if (std::find(pos.begin(), pos.end(), p) == pos.end()) ++union_size;
      }
    if (hit.empty()) {
      // disjoint from every open cluster: join the fullest one that still has room
      size_t best = open.size();
      for (size_t c = 0; c < open.size(); ++c)
        if (open[c].pos.size() + pos.size() <= K && (best == open.size() || open[c].pos.size() > open[best].pos.size()))
          best = c;
      if (best == open.size()) {
        Cluster cl;
        cl.pos = {};
        cl.m = {cd(1, 0)};
        open.push_back(cl);
      }
      cluster_apply(&open[best], pos, m);
    } else if (union_size <= K) {
      // merge the overlapped clusters (mutually disjoint => their product is a Kronecker product)
      for (size_t h = 1; h < hit.size(); ++h) cluster_apply(&open[hit[0]], open[hit[h]].pos, open[hit[h]].m);
      for (size_t h = hit.size(); h-- > 1;) open.erase(open.begin() + hit[h]);
      cluster_apply(&open[hit[0]], pos, m);
    } else {
      QCHK(flush_overlapping(pos));
      Cluster cl;
      cl.pos = {};
      cl.m = {cd(1, 0)};
      cluster_apply(&cl, pos, m);
      open.push_back(cl);
    }
  }
  for (const Cluster& cl : open) QCHK(flush_cluster<T>(s, cl));
  return QIP_OK;
}

// ---------------------------------------------------------------------------------------
// LDS-resident multi-gate sweeps (option "tile"): the scheduler cuts the circuit into segments whose
// gates all live on index bits 0..5 plus five freely chosen higher bits, and k_tile_gates applies a whole
// segment with one read and one write of the vector.
//   tile = 1  circuit order up to EXACT commutations (a rounding-free gate — X, CNOT, SWAP, Z, S ... — may pass
//             gates on other qubits and vice versa): every amplitude sees the same rounded operations in the
//             same order as in the gate-by-gate path, hence IEEE-equal results;
//   tile = 2  any gate may be hoisted over skipped gates it shares no qubit with (they commute mathematically,
//             not in floating point); equal to the reference up to rounding (1e-12 bar).
// ---------------------------------------------------------------------------------------
struct TileItem {
  bool tileable = false;
  // every matrix entry is in {0, +-1, +-i}: the gate moves / negates / rotates amplitudes by 90 degrees without
  // any rounding, so it commutes with gates on other qubits EXACTLY (IEEE ==), not just mathematically
  bool exact = false;
  int kind = 0;                 // TileGate kind
  std::vector<uint32_t> pos;    // every involved bit position
  uint32_t t0 = 0, t1 = 0, t2 = 0;  // target position(s)
  std::vector<uint32_t> cpos;
  double m[8] = {0};
  uint32_t nz = 0;
  std::vector<double> mat;  // kind 3 / 4: 4x4 / 8x8 row-major as re,im pairs, sub-index MSB = t0
  // how the gate acts on each of its bits: `nd_mask` = it exchanges amplitudes across the bit (dense target, swap
  // bits), `d_mask` = it only tests the bit (controls, diagonal targets).  Two gates commute when on every bit
  // they share both only test it.  Ops that are not tileable count every bit as exchanged.
  uint64_t nd_mask = 0, d_mask = 0;
  // an uncontrolled Swap(h) (any h): its h transpositions as pairs of bit positions.  A run of such ops composes to ONE
  // permutation of the index bits (launch_permute)
  std::vector<std::pair<uint32_t, uint32_t>> swap_pairs;
};

static int classify_tile_item(int dtype, uint32_t n, const qip_op* op, TileItem* it) {
  FlatOp f;
  QCHK(flatten_op(n, op, false, &f));
  Plan p;
  QCHK(make_plan(dtype, n, f, false, &p));
  it->tileable = false;
  it->pos.clear();
  for (uint32_t c : p.cpos) it->pos.push_back(c);
  for (uint32_t t : p.opos) it->pos.push_back(t);
  it->cpos = p.cpos;
  it->swap_pairs.clear();
  if (!f.distinct) return QIP_OK;
  const uint32_t k = (uint32_t)p.opos.size();
  if (p.cls == KC_SWAP_BITS && p.cpos.empty())
    for (uint32_t j = 0; j < k / 2; ++j) it->swap_pairs.push_back({p.opos[j], p.opos[k / 2 + j]});
  auto unit_axis = [](double re, double im) {  // 0, +-1 or +-i
    return (re == 0.0 && (im == 0.0 || im == 1.0 || im == -1.0)) || (im == 0.0 && (re == 1.0 || re == -1.0));
  };
  if (p.cls == KC_GATE1Q_PAIR) {
    it->kind = 0;
    it->t0 = p.opos[0];
    memcpy(it->m, p.m, sizeof it->m);
    it->nz = p.nz;
    it->tileable = true;
    it->exact = true;
    for (int e = 0; e < 4; ++e) it->exact = it->exact && unit_axis(p.m[2 * e], p.m[2 * e + 1]);
    // at most one non-zero entry per row, else the row is a sum of two terms (rounded)
    it->exact = it->exact && !((p.nz & 1u) && (p.nz & 2u)) && !((p.nz & 4u) && (p.nz & 8u));
  } else if (p.cls == KC_PHASE && k == 1) {
    it->kind = 1;
    it->t0 = p.opos[0];
    const bool on_one = p.phase_ones & 1ull;
    it->m[0] = on_one ? 1.0 : p.phase[0];
    it->m[1] = on_one ? 0.0 : p.phase[1];
    it->m[2] = on_one ? p.phase[0] : 1.0;
    it->m[3] = on_one ? p.phase[1] : 0.0;
    it->tileable = true;
    it->exact = unit_axis(p.phase[0], p.phase[1]);
  } else if (p.cls == KC_DIAG && k == 1) {
    it->kind = 1;
    it->t0 = p.opos[0];
    for (int e = 0; e < 4; ++e) it->m[e] = p.table[e];
    it->tileable = true;
    it->exact = unit_axis(p.table[0], p.table[1]) && unit_axis(p.table[2], p.table[3]);
  } else if (p.cls == KC_SWAP_BITS && k == 2) {
    it->kind = 2;
    it->t0 = std::min(p.opos[0], p.opos[1]);
    it->t1 = std::max(p.opos[0], p.opos[1]);
    it->tileable = true;
    it->exact = true;
  } else if (p.cls == KC_GATE_KQ && k == 2 && p.table.size() == 32) {
    it->kind = 3;  // dense 2-qubit gate: both targets exchange amplitudes
    it->t0 = p.opos[0];
    it->t1 = p.opos[1];
    it->mat = p.table;
    it->tileable = true;
  } else if (p.cls == KC_GATE_KQ && k == 3 && p.table.size() == 128) {
    it->kind = 4;  // dense 3-qubit gate: a pass whose three bits are its targets holds one group per lane
    it->t0 = p.opos[0];
    it->t1 = p.opos[1];
    it->t2 = p.opos[2];
    it->mat = p.table;
    it->tileable = true;
  } else if (p.cls == KC_NOOP) {
    it->exact = true;  // identity: nothing happens (not tileable, launches nothing)
  }
  it->nd_mask = it->d_mask = 0;
  if (it->tileable) {
    for (uint32_t c : p.cpos) it->d_mask |= 1ull << c;
    if (it->kind == 1) it->d_mask |= 1ull << it->t0;
    else it->nd_mask |= 1ull << it->t0;
    if (it->kind >= 2) it->nd_mask |= 1ull << it->t1;
    if (it->kind == 4) it->nd_mask |= 1ull << it->t2;
  } else {
    for (uint32_t b : it->pos) it->nd_mask |= 1ull << b;
  }
  return QIP_OK;
}

// Lane-id bit -> tile bit for one pass of k_tile_passes (TilePass::lanepos): lane bit j < S goes on tile bit j or
// j + S (whichever is not a pass bit), so the S swizzled slot bits enumerate the lanes of an LDS bank group; the
// other lane bits fill what is left, bits that are not folded (>= 2S) first, then partners of the pairs that hold
// lane bits 0 and 1 (lane bit 4 varies inside a ds_read_b128 group in a pattern that is closed under flipping lane
// bits 0 / 1 only).  S = 4 for 16-byte amplitudes, 5 for 8-byte ones.  `pb` ascending and distinct.  Returns 0 if
// the result is not a bijection onto the non-pass bits (cannot happen; the caller refuses to launch).
// Checked against the LDS banking model of MI355X_MICROARCH.md in tests/test_host_ops.py.
static uint64_t tile_lane_assignment(const uint32_t pb[3], uint32_t S) {
  auto has = [&](uint32_t t) { return t == pb[0] || t == pb[1] || t == pb[2]; };
  int pos_of[kTileLaneBits];
  for (int k = 0; k < kTileLaneBits; ++k) pos_of[k] = -1;
  bool used[kTileBits] = {false};
  std::vector<int> rest_bits;
  for (uint32_t j = 0; j < S; ++j) {
    int where = -1;
    for (uint32_t c : {j, j + S})
      if (c < (uint32_t)kTileBits && !has(c) && !used[c]) {
        where = (int)c;
        break;
      }
    if (where >= 0) {
      pos_of[j] = where;
      used[where] = true;
    } else {
      rest_bits.push_back((int)j);
    }
  }
  for (int k = (int)S; k < kTileLaneBits; ++k) rest_bits.push_back(k);
  std::vector<int> rest_pos;
  for (int t = 0; t < kTileBits; ++t)
    if (!has((uint32_t)t) && !used[t]) rest_pos.push_back(t);
  auto rank = [&](int t) {
    if (t >= (int)(2 * S)) return 0;
    const int j = t >= (int)S ? t - (int)S : t;
    for (int k = 0; k < 2; ++k)
      if (pos_of[k] == j || pos_of[k] == j + (int)S) return 1;
    return 2;
  };
  std::stable_sort(rest_pos.begin(), rest_pos.end(), [&](int x, int y) { return rank(x) < rank(y); });
  if (rest_bits.size() != rest_pos.size()) return 0;
  for (size_t q = 0; q < rest_bits.size(); ++q) pos_of[rest_bits[q]] = rest_pos[q];
  uint64_t lanepos = 0;
  uint32_t covered = 0;
  for (int k = 0; k < kTileLaneBits; ++k) {
    lanepos |= (uint64_t)pos_of[k] << (4 * k);
    covered |= 1u << pos_of[k];
  }
  for (int j = 0; j < 3; ++j) covered |= 1u << pb[j];
  // (all-zero is not a valid assignment: thread-id bits 0 and 1 cannot both sit on tile bit 0)
  return covered == (1u << kTileBits) - 1u ? lanepos : 0ull;
}

extern "C" int qip_hip_tile_lane_assignment(int dtype, const uint32_t* pass_bits, uint64_t* lanepos) try {
  if (!pass_bits || !lanepos) return fail(QIP_ERR_INVALID, "null argument");
  if (dtype != QIP_C64 && dtype != QIP_C32) return fail(QIP_ERR_INVALID, "bad dtype %d", dtype);
  uint32_t pb[3] = {pass_bits[0], pass_bits[1], pass_bits[2]};
  if (!(pb[0] < pb[1] && pb[1] < pb[2] && pb[2] < (uint32_t)kTileBits))
    return fail(QIP_ERR_INVALID, "pass bits must be ascending, distinct and below %d", kTileBits);
  *lanepos = tile_lane_assignment(pb, dtype == QIP_C64 ? 4u : 5u);
  if (*lanepos == 0ull) return fail(QIP_ERR_UNSUPPORTED, "lane-bit assignment is not a bijection (internal error)");
  return QIP_OK;
} QIP_CATCH_ALL

// Everything the host decides about one segment before anything touches the device: which amplitude-index
// positions the tile's free bits 6..10 stand for, the gate descriptors, the passes and each gate's resolution
// against its pass.  Pure host code: qip_hip_debug_tile_plan serialises it so that tests can replay a plan on
// the CPU (tests/test_tile_plan_cpu.py) and check it against the oracle without a GPU.
template <typename T> struct TileSegmentPlan {
  std::vector<uint32_t> high;  // amplitude-index position of tile bit 6 + j
  std::vector<TileGate<T>> gates;
  std::vector<amp_t<T>> mats;  // 4x4 matrices of the dense 2-qubit gates (kind 3), 16 entries each
  TilePassDesc pd;             // passes (only when `passes`)
};

template <typename T>
static int build_tile_segment(uint32_t n, bool passes, const std::vector<const TileItem*>& seg,
                              std::vector<uint32_t> high, TileSegmentPlan<T>* out) {
  // pad the free bits with unused positions >= kTileLow so the tile always has kTileHigh of them
  for (uint32_t p = kTileLow; high.size() < (size_t)kTileHigh && p < n; ++p)
    if (std::find(high.begin(), high.end(), p) == high.end()) high.push_back(p);
  // the first kTileWaveBits free positions are wave bits at load / store time, the last three are the lane's own
  // elements: give the free positions that are exchange targets least often to the wave bits
  std::vector<uint32_t> uses(64, 0);
  for (const TileItem* it : seg) {
    if (it->kind == 0) uses[it->t0] += 1;
    if (it->kind >= 2) {
      uses[it->t0] += 1;
      uses[it->t1] += 1;
    }
    if (it->kind == 4) uses[it->t2] += 1;
  }
  std::stable_sort(high.begin(), high.end(), [&](uint32_t a, uint32_t b) { return uses[a] < uses[b]; });
  auto tile_bit = [&](uint32_t pos) -> uint32_t {  // kTileOutside when the position is not part of the tile
    if (pos < (uint32_t)kTileLow) return pos;
    const auto f = std::find(high.begin(), high.end(), pos);
    return f == high.end() ? kTileOutside : kTileLow + (uint32_t)(f - high.begin());
  };
  std::vector<TileGate<T>>& gates = out->gates;
  std::vector<amp_t<T>>& mats = out->mats;
  gates.assign(seg.size(), TileGate<T>());
  mats.clear();
  for (size_t i = 0; i < seg.size(); ++i) {
    const TileItem& it = *seg[i];
    TileGate<T>& g = gates[i];
    memset(&g, 0, sizeof g);
    g.kind = (uint32_t)it.kind;
    g.b0 = tile_bit(it.t0);
    g.b1 = it.kind >= 2 ? tile_bit(it.t1) : 0;
    if (it.kind == 2 && g.b0 > g.b1) std::swap(g.b0, g.b1);  // (kind 3 keeps b0 = the sub-index MSB)
    if (it.kind == 3 || it.kind == 4) {
      g.nz = (uint32_t)(mats.size() / 16);  // where its 4x4 / 8x8 starts in the matrix block behind the gate list (units of 16)
      const int cnt = it.kind == 3 ? 16 : 64;
      for (int e = 0; e < cnt; ++e) mats.push_back(mk<T>(it.mat[2 * e], it.mat[2 * e + 1]));
    }
    if (it.kind == 4) g.tpos_out = tile_bit(it.t2);  // (tile bit of the sub-index LSB; the field is otherwise unused for this kind)
    if (it.kind == 1 && g.b0 == kTileOutside) g.tpos_out = it.t0;
    for (uint32_t c : it.cpos) {
      const uint32_t tb = tile_bit(c);
      if (tb == kTileOutside) g.omask |= 1ull << c;
      else g.cmask |= 1u << tb;
    }
    if (it.kind != 3 && it.kind != 4) g.nz = it.nz;
    if (it.kind == 0) {
      for (int e = 0; e < 4; ++e) g.m[e] = mk<T>(it.m[2 * e], it.m[2 * e + 1]);
      if (passes) {  // flop-saving flags (k_tile_passes only; k_tile_gates reads b1 = 0)
        const bool real = it.m[1] == 0 && it.m[3] == 0 && it.m[5] == 0 && it.m[7] == 0;
        const bool is_x = it.nz == 6u && it.m[2] == 1 && it.m[3] == 0 && it.m[4] == 1 && it.m[5] == 0;
        g.b1 = (real ? 1u : 0u) | (is_x ? 2u : 0u);
      }
    } else if (it.kind == 1) {
      g.m[0] = mk<T>(it.m[0], it.m[1]);
      g.m[1] = mk<T>(it.m[2], it.m[3]);
    }
  }
  out->high = high;
  memset(&out->pd, 0, sizeof out->pd);
  if (passes) {
    // group consecutive gates into passes of at most three distinct exchange bits (see k_tile_passes)
    TilePassDesc& pd = out->pd;
    memset(&pd, 0, sizeof pd);
    for (int j = 0; j < kTileHigh; ++j) pd.hpos[j] = high[j];
    std::vector<uint32_t> bits;
    uint32_t first = 0;
    constexpr uint32_t S = sizeof(amp_t<T>) == 16 ? 4u : 5u;  // swizzle fold width, see TilePass
    bool pass_layout_ok = true;
    auto close_pass = [&](uint32_t end) {
      std::vector<uint32_t> b = bits;
      auto has = [&](uint32_t t) { return std::find(b.begin(), b.end(), t) != b.end(); };
      // Pad to three bits.  A free slot is best spent on a bit the pass's DIAGONAL gates test: a control on a
      // pass bit, or the target of a gate with one unit entry (phase, T, S, Z, controlled-phase), turns "multiply
      // all eight elements" into "multiply the four (two) that can change" by a scalar branch.  Otherwise from the
      // top; never completing a pair (t, t +- S) unless nothing else is left: such a pass is 2-way
      // bank-conflicted (and low pad bits are what made the linear layout 8-way).
      int score[kTileBits] = {0};
      for (uint32_t gi = first; gi < end; ++gi) {
        const TileGate<T>& g = gates[gi];
        if (g.kind != 1) continue;
        for (int t = 0; t < kTileBits; ++t)
          if ((g.cmask >> t) & 1u) score[t] += 1;
        const bool unit0 = g.m[0].x == (T)1 && g.m[0].y == (T)0, unit1 = g.m[1].x == (T)1 && g.m[1].y == (T)0;
        if (g.b0 != kTileOutside && (unit0 || unit1)) score[g.b0] += 1;
      }
      std::vector<int> order;
      for (int t = kTileBits - 1; t >= 0; --t) order.push_back(t);
      std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return score[x] > score[y]; });
      for (int relax = 0; relax < 2 && b.size() < 3; ++relax)
        for (size_t q = 0; q < order.size() && b.size() < 3; ++q) {
          const int t = order[q];
          if (has((uint32_t)t)) continue;
          const bool pairs = has((uint32_t)t + S) || (t >= (int)S && has((uint32_t)t - S));
          if (pairs && relax == 0) continue;
          b.push_back((uint32_t)t);
        }
      std::sort(b.begin(), b.end());
      TilePass& ps = pd.pass[pd.npasses++];
      ps.first = first;
      ps.count = end - first;
      for (int j = 0; j < 3; ++j) ps.pb[j] = b[j];
      ps.lanepos = tile_lane_assignment(ps.pb, S);
      if (ps.lanepos == 0ull) pass_layout_ok = false;  // not a bijection: refuse to launch
      first = end;
      bits.clear();
    };
    for (uint32_t i = 0; i < (uint32_t)gates.size(); ++i) {
      std::vector<uint32_t> add;
      if (gates[i].kind == 0) add = {gates[i].b0};
      if (gates[i].kind == 2 || gates[i].kind == 3) add = {gates[i].b0, gates[i].b1};
      if (gates[i].kind == 4) add = {gates[i].b0, gates[i].b1, gates[i].tpos_out};  // exactly the pass
      if (!add.empty()) {
        // a control of a dense gate / swap is a scalar branch on a pass bit but a per-lane select on a lane
        // bit (k_tile_passes): make the in-tile controls pass bits too whenever the three slots allow
        std::vector<uint32_t> with_ctl = add;
        for (uint32_t t = 0; t < (uint32_t)kTileBits; ++t)
          if ((gates[i].cmask >> t) & 1u) with_ctl.push_back(t);
        if (with_ctl.size() <= 3) add = with_ctl;
      }
      auto merge = [&](const std::vector<uint32_t>& extra) {
        std::vector<uint32_t> m = bits;
        for (uint32_t b : extra)
          if (std::find(m.begin(), m.end(), b) == m.end()) m.push_back(b);
        return m;
      };
      std::vector<uint32_t> merged = merge(add);
      if (merged.size() > 3) {
        close_pass(i);
        merged = add;
      }
      bits = merged;
    }
    close_pass((uint32_t)gates.size());
    if (!pass_layout_ok) return fail(QIP_ERR_UNSUPPORTED, "tile pass: lane-bit assignment is not a bijection (internal error)");
    // resolve every gate against its pass: code path, pass-bit index of its bit(s), controls split into pass-bit
    // and lane-bit parts (see TileGate::op)
    for (uint32_t pi = 0; pi < pd.npasses; ++pi) {
      const TilePass& ps = pd.pass[pi];
      const uint32_t passmask = (1u << ps.pb[0]) | (1u << ps.pb[1]) | (1u << ps.pb[2]);
      auto jof = [&](uint32_t bit) { return bit == ps.pb[0] ? 0u : bit == ps.pb[1] ? 1u : 2u; };
      for (uint32_t gi = ps.first; gi < ps.first + ps.count; ++gi) {
        TileGate<T>& g = gates[gi];
        g.cm_reg = g.cmask & passmask;
        g.cm_lane = g.cmask & ~passmask;
        const bool lane_ctl = g.cm_lane != 0u;
        if (g.kind == 1) {
          const bool outside = g.b0 == kTileOutside;
          if (outside && !lane_ctl) g.op = TOP_DIAG_UNIFORM;
          else if (outside || !((passmask >> g.b0) & 1u)) g.op = lane_ctl ? TOP_DIAG_LANE_CTL : TOP_DIAG_LANE;
          else g.op = TOP_DIAG_REG0 + jof(g.b0);
        } else if (g.kind == 0) {
          g.op = (lane_ctl ? TOP_DENSE_LANE0 : TOP_DENSE0) + jof(g.b0);
        } else if (g.kind == 4) {
          const uint32_t ja = jof(g.b0), jb = jof(g.b1), jc = jof(g.tpos_out);
          static const uint32_t t3[3][3] = {{0, TOP_DENSE3Q_012, TOP_DENSE3Q_021}, {TOP_DENSE3Q_102, 0, TOP_DENSE3Q_120},
                                            {TOP_DENSE3Q_201, TOP_DENSE3Q_210, 0}};
          g.op = t3[ja][jb];
          (void)jc;  // = 3 - ja - jb
        } else if (g.kind == 3) {
          const uint32_t ja = jof(g.b0), jb = jof(g.b1);
          static const uint32_t table[3][3] = {{0, TOP_DENSE2Q_01, TOP_DENSE2Q_02},
                                               {TOP_DENSE2Q_10, 0, TOP_DENSE2Q_12},
                                               {TOP_DENSE2Q_20, TOP_DENSE2Q_21, 0}};
          g.op = table[ja][jb];
        } else {
          const uint32_t ja = jof(g.b0), jb = jof(g.b1);  // b0 < b1 and pass bits ascend, so ja < jb
          g.op = ja == 0 ? (jb == 1 ? TOP_SWAP_01 : TOP_SWAP_02) : TOP_SWAP_12;
        }
      }
    }
  }
  return QIP_OK;
}

// ---------------------------------------------------------------------------------------
// Segment-specialised tile sweeps (option "tile_jit"): the interpreter k_tile_passes spends most of its issue slots
// on decoding — per gate and per wave ~54 scalar + ~58 vector instructions that only depend on the segment
// (profiles/r01_tile_pmc.md).  Here the host writes the segment out as straight-line HIP source — the same load /
// pass / store skeleton, one call of the very same pass_* helper per gate with the gate descriptor as a constexpr
// value — and compiles it with hiprtc against the embedded qip_kernels.h.  Every op code, bit position, control mask,
// zero / real / X flag folds away; what is left per gate is its arithmetic, operation for operation what the
// interpreter executes, so results are bit-identical.  Kernels are cached per process by their source text, so a
// circuit replayed many times (programs, variational loops with fixed angles) compiles once (~0.3-1 s per segment).
// ---------------------------------------------------------------------------------------
static const char kKernelsHeaderSrc[] =
#include "qip_kernels_embed.inc"
    ;

struct Hiprtc {
  void* handle = nullptr;
  int (*CreateProgram)(void**, const char*, const char*, int, const char**, const char**) = nullptr;
  int (*CompileProgram)(void*, int, const char**) = nullptr;
  int (*GetProgramLogSize)(void*, size_t*) = nullptr;
  int (*GetProgramLog)(void*, char*) = nullptr;
  int (*GetCodeSize)(void*, size_t*) = nullptr;
  int (*GetCode)(void*, char*) = nullptr;
  int (*DestroyProgram)(void**) = nullptr;
};
static Hiprtc g_rtc;
static int hiprtc_load() {
  if (g_rtc.handle) return QIP_OK;
  void* h = nullptr;
  for (const char* name : {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"}) {
    h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (h) break;
  }
  if (!h) return fail(QIP_ERR_UNSUPPORTED, "option tile_jit needs libhiprtc: %s", dlerror());
#define RSYM(field, name)                                                           \
  do {                                                                              \
    *(void**)(&g_rtc.field) = dlsym(h, name);                                       \
    if (!g_rtc.field) return fail(QIP_ERR_UNSUPPORTED, "libhiprtc lacks %s", name); \
  } while (0)
  RSYM(CreateProgram, "hiprtcCreateProgram");
  RSYM(CompileProgram, "hiprtcCompileProgram");
  RSYM(GetProgramLogSize, "hiprtcGetProgramLogSize");
  RSYM(GetProgramLog, "hiprtcGetProgramLog");
  RSYM(GetCodeSize, "hiprtcGetCodeSize");
  RSYM(GetCode, "hiprtcGetCode");
  RSYM(DestroyProgram, "hiprtcDestroyProgram");
#undef RSYM
  g_rtc.handle = h;
  return QIP_OK;
}

// source -> code object (host only: works without a device, which is how the CPU tests cover it)
static int hiprtc_compile(const std::string& src, std::vector<char>* code) {
  QCHK(hiprtc_load());
  void* prog = nullptr;
  const char* hdr_src[1] = {kKernelsHeaderSrc};
  const char* hdr_name[1] = {"qip_kernels.h"};
  if (g_rtc.CreateProgram(&prog, src.c_str(), "qip_segment.hip", 1, hdr_src, hdr_name) != 0)
    return fail(QIP_ERR_DEVICE, "hiprtcCreateProgram failed");
  const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize"};  // as rustqip_amd/build.py
  const int rc = g_rtc.CompileProgram(prog, (int)(sizeof opts / sizeof opts[0]), opts);
  if (rc != 0) {
    size_t n = 0;
    g_rtc.GetProgramLogSize(prog, &n);
    std::string log(n + 1, '\0');
    if (n) g_rtc.GetProgramLog(prog, &log[0]);
    g_rtc.DestroyProgram(&prog);
    return fail(QIP_ERR_DEVICE, "hiprtc could not compile a tile segment (%d): %.800s", rc, log.c_str());
  }
  size_t sz = 0;
  g_rtc.GetCodeSize(prog, &sz);
  code->resize(sz);
  g_rtc.GetCode(prog, code->data());
  g_rtc.DestroyProgram(&prog);
  return QIP_OK;
}

struct JitKernel {
  hipModule_t module = nullptr;
  hipFunction_t fn = nullptr;
};
static std::map<std::string, JitKernel> g_jit_cache;  // key: device ordinal + source text
static uint64_t g_jit_compiles = 0;
static double g_jit_compile_ms = 0;

extern "C" int qip_hip_jit_stats(uint64_t* kernels_compiled, double* compile_ms) try {
  if (kernels_compiled) *kernels_compiled = g_jit_compiles;
  if (compile_ms) *compile_ms = g_jit_compile_ms;
  return QIP_OK;
} QIP_CATCH_ALL

template <typename T> static std::string fnum(T v) {
  char buf[64];
  if (std::is_same<T, double>::value) snprintf(buf, sizeof buf, "%.17g", (double)v);
  else snprintf(buf, sizeof buf, "%.9gf", (double)v);
  std::string r = buf;
  // a bare integer literal would not be floating point ("1" / "1f")
  if (std::is_same<T, double>::value && r.find_first_of(".eEn") == std::string::npos) r += ".0";
  if (!std::is_same<T, double>::value && r.find_first_of(".eEn") == std::string::npos) r.insert(r.size() - 1, ".0");
  return r;
}

// The segment as HIP source (see the block comment above).  Mirrors k_tile_passes statement by statement.
template <typename T>
static std::string tile_jit_source(const TileSegmentPlan<T>& plan, const Ins& ins, bool nt) {
  const char* tname = std::is_same<T, double>::value ? "double" : "float";
  std::string o;
  auto L = [&](const std::string& line) { o += line; o += "\n"; };
  auto U = [](uint64_t v) { return std::to_string(v) + "ull"; };
  auto amp = [&](amp_t<T> a) { return "{" + fnum<T>(a.x) + ", " + fnum<T>(a.y) + "}"; };
  const TilePassDesc& d = plan.pd;
  L("#include \"qip_kernels.h\"");
  L("using namespace qipk;");
  L(std::string("typedef ") + tname + " T;");
  L("typedef amp_t<T> A;");
  L("extern \"C\" __global__ __launch_bounds__(kTileBlock, 5) void qip_segment(A* __restrict__ st) {");
  L("  extern __shared__ __attribute__((aligned(16))) unsigned char tile_raw[];");
  L("  A* tile = reinterpret_cast<A*>(tile_raw);");
  L(std::string("  constexpr bool NT = ") + (nt ? "true" : "false") + ";");
  L("  const uint32_t tid = threadIdx.x, lane = tid & 63u;");
  L("  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);");
  L("  uint64_t wbase = (blockIdx.x + (uint64_t)blockIdx.y * gridDim.x) << kTileLow;");
  for (uint32_t j = 0; j < ins.npos; ++j) {  // insert_bits with the positions as literals
    const std::string p = std::to_string(ins.pos[j]);
    L("  wbase = ((wbase >> " + p + ") << " + std::to_string(ins.pos[j] + 1) + ") | (wbase & ((1ull << " + p + ") - 1ull));");
  }
  if (ins.ormask) L("  wbase |= " + U(ins.ormask) + ";");
  L("  const uint64_t base = wbase;");
  for (int j = 0; j < kTileWaveBits; ++j)
    L("  wbase |= (uint64_t)((wave >> " + std::to_string(j) + ") & 1u) << " + std::to_string(d.hpos[j]) + ";");
  L("  const uint32_t slot_tid = tile_slot<A>(tid);");
  auto ub = [&](int u) {
    uint64_t off = 0;
    for (int b = 0; b < 3; ++b)
      if ((u >> b) & 1) off |= 1ull << d.hpos[kTileWaveBits + b];
    return U(off);
  };
  L("  {");
  L("    A x[8];");
  for (int u = 0; u < 8; ++u) L("    x[" + std::to_string(u) + "] = ldg<NT>(st + (wbase | " + ub(u) + ") + lane);");
  for (int u = 0; u < 8; ++u)
    L("    tile[slot_tid ^ tile_slot<A>(" + std::to_string(u) + "u << kTileLaneBits)] = x[" + std::to_string(u) + "];");
  L("  }");
  L("  __syncthreads();");
  for (uint32_t pi = 0; pi < d.npasses; ++pi) {
    const TilePass& ps = d.pass[pi];
    L("  {  // pass " + std::to_string(pi));
    L("    uint32_t tb = 0;");
    for (int k = 0; k < kTileLaneBits; ++k)
      L("    tb |= ((tid >> " + std::to_string(k) + ") & 1u) << " + std::to_string((unsigned)((ps.lanepos >> (4 * k)) & 15ull)) + ";");
    L("    const uint32_t slot_tb = tile_slot<A>(tb);");
    std::string cs = "    const uint32_t c[8] = {";
    for (int i = 0; i < 8; ++i) {
      const uint32_t c = ((uint32_t)(i & 1) << ps.pb[0]) | ((uint32_t)((i >> 1) & 1) << ps.pb[1]) | ((uint32_t)((i >> 2) & 1) << ps.pb[2]);
      cs += std::to_string(c) + "u" + (i < 7 ? ", " : "};");
    }
    L(cs);
    L("    A e[8];");
    for (int i = 0; i < 8; ++i) L("    e[" + std::to_string(i) + "] = tile[slot_tb ^ tile_slot<A>(c[" + std::to_string(i) + "])];");
    for (uint32_t gi = ps.first; gi < ps.first + ps.count; ++gi) {
      const TileGate<T>& g = plan.gates[gi];
      L("    {  // gate " + std::to_string(gi));
      L("      constexpr TileGate<T> g = {" + std::to_string(g.kind) + "u, " + std::to_string(g.b0) + "u, " + std::to_string(g.b1) + "u, " +
        std::to_string(g.cmask) + "u, " + std::to_string(g.nz) + "u, " + std::to_string(g.tpos_out) + "u, " + U(g.omask) + ", " +
        std::to_string(g.op) + "u, " + std::to_string(g.cm_reg) + "u, " + std::to_string(g.cm_lane) + "u, 0u, {" + amp(g.m[0]) + ", " +
        amp(g.m[1]) + ", " + amp(g.m[2]) + ", " + amp(g.m[3]) + "}};");
      const std::string lane_args = "g.cm_lane != 0u, (tb & g.cm_lane) == g.cm_lane";
      std::string call;
      switch (g.op) {
        case TOP_DIAG_UNIFORM:
          call = "const A f = ((base >> g.tpos_out) & 1ull) ? g.m[1] : g.m[0]; if (!(f.x == (T)1 && f.y == (T)0)) pass_scale<T, 0, -1>(f, e, c, g.cm_reg);";
          break;
        case TOP_DIAG_LANE:
        case TOP_DIAG_LANE_CTL:
          call = std::string("const bool one = ") + (g.b0 == kTileOutside ? "((base >> g.tpos_out) & 1ull) != 0" : "((tb >> g.b0) & 1u) != 0") +
                 "; A f = tile_sel(one, g.m[1], g.m[0]); " +
                 (g.op == TOP_DIAG_LANE_CTL ? "{ const bool lane_ok = (tb & g.cm_lane) == g.cm_lane; f.x = lane_ok ? f.x : (T)1; f.y = lane_ok ? f.y : (T)0; } " : "") +
                 "pass_scale<T, 0, -1>(f, e, c, g.cm_reg);";
          break;
        case TOP_DIAG_REG0: case TOP_DIAG_REG1: case TOP_DIAG_REG2:
          call = "pass_diag<T, " + std::to_string(g.op - TOP_DIAG_REG0) + ">(g, e, c, g.cm_reg, " + lane_args + ");";
          break;
        case TOP_DENSE0: case TOP_DENSE1: case TOP_DENSE2:
          call = "pass_dense<T, " + std::to_string(g.op - TOP_DENSE0) + ">(g, e, c, g.cm_reg);";
          break;
        case TOP_DENSE_LANE0: case TOP_DENSE_LANE1: case TOP_DENSE_LANE2:
          call = "pass_dense_lane<T, " + std::to_string(g.op - TOP_DENSE_LANE0) + ">(g, e, c, g.cm_reg, (tb & g.cm_lane) == g.cm_lane);";
          break;
        case TOP_DENSE2Q_01: case TOP_DENSE2Q_02: case TOP_DENSE2Q_10: case TOP_DENSE2Q_12: case TOP_DENSE2Q_20: case TOP_DENSE2Q_21: {
          static const int ja[6] = {0, 0, 1, 1, 2, 2}, jb[6] = {1, 2, 0, 2, 0, 1};
          std::string m = "const A M[16] = {";
          for (int e = 0; e < 16; ++e) m += amp(plan.mats[16 * g.nz + e]) + (e < 15 ? ", " : "}; ");
          call = m + "pass_dense2<T, " + std::to_string(ja[g.op - TOP_DENSE2Q_01]) + ", " + std::to_string(jb[g.op - TOP_DENSE2Q_01]) + ">(M, e, c, g.cm_reg, " + lane_args + ");";
          break;
        }
        case TOP_DENSE3Q_012: case TOP_DENSE3Q_021: case TOP_DENSE3Q_102: case TOP_DENSE3Q_120: case TOP_DENSE3Q_201: case TOP_DENSE3Q_210: {
          static const int ja[6] = {0, 0, 1, 1, 2, 2}, jb[6] = {1, 2, 0, 2, 0, 1};
          const int a = ja[g.op - TOP_DENSE3Q_012], b = jb[g.op - TOP_DENSE3Q_012];
          std::string m = "const A M[64] = {";
          for (int e = 0; e < 64; ++e) m += amp(plan.mats[16 * g.nz + e]) + (e < 63 ? ", " : "}; ");
          call = m + "pass_dense3<T, " + std::to_string(a) + ", " + std::to_string(b) + ", " + std::to_string(3 - a - b) + ">(M, e, " + lane_args + ");";
          break;
        }
        case TOP_SWAP_01: call = "pass_swap<T, 0, 1>(e, c, g.cm_reg, " + lane_args + ");"; break;
        case TOP_SWAP_02: call = "pass_swap<T, 0, 2>(e, c, g.cm_reg, " + lane_args + ");"; break;
        case TOP_SWAP_12: call = "pass_swap<T, 1, 2>(e, c, g.cm_reg, " + lane_args + ");"; break;
        default: break;
      }
      if (g.omask) L("      if ((base & g.omask) == g.omask) { " + call + " }");  // an outside control is 0 for this whole tile
      else L("      { " + call + " }");
      L("    }");
    }
    for (int i = 0; i < 8; ++i) L("    tile[slot_tb ^ tile_slot<A>(c[" + std::to_string(i) + "])] = e[" + std::to_string(i) + "];");
    L("    __syncthreads();");
    L("  }");
  }
  for (int u = 0; u < 8; ++u)
    L("  stg<NT>(st + (wbase | " + ub(u) + ") + lane, tile[slot_tid ^ tile_slot<A>(" + std::to_string(u) + "u << kTileLaneBits)]);");
  L("}");
  return o;
}

static int jit_get_kernel(qip_hip_state* s, const std::string& src, hipFunction_t* fn) {
  const std::string key = std::to_string(s->device) + "\n" + src;
  auto it = g_jit_cache.find(key);
  if (it != g_jit_cache.end()) {
    *fn = it->second.fn;
    return QIP_OK;
  }
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<char> code;
  QCHK(hiprtc_compile(src, &code));
  JitKernel k;
  HIPCHK(hipModuleLoadData(&k.module, code.data()));
  HIPCHK(hipModuleGetFunction(&k.fn, k.module, "qip_segment"));
  g_jit_compiles += 1;
  g_jit_compile_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  g_jit_cache[key] = k;
  *fn = k.fn;
  return QIP_OK;
}

template <typename T>
static int launch_tile_segment(qip_hip_state* s, const std::vector<const TileItem*>& seg,
                               std::vector<uint32_t> high_in) {
  TileSegmentPlan<T> plan;
  QCHK(build_tile_segment<T>(s->n, s->tile_passes != 0, seg, std::move(high_in), &plan));
  const std::vector<uint32_t>& high = plan.high;
  std::vector<TileGate<T>>& gates = plan.gates;
  std::vector<amp_t<T>>& mats = plan.mats;
  TilePassDesc& pd = plan.pd;
  const size_t gates_bytes = gates.size() * sizeof(TileGate<T>);
  static_assert(sizeof(TileGate<T>) % 16 == 0, "the matrix block behind the gate list stays 16-byte aligned");
  auto upload_gates = [&]() -> int {  // after the gates are final (k_tile_passes resolves them per pass first)
    QCHK(ensure_arena(s, gates_bytes + mats.size() * sizeof(amp_t<T>)));  // one allocation: growing frees the old arena
    QCHK(arena_upload(s, gates.data(), gates_bytes, 0));
    if (!mats.empty()) QCHK(arena_upload(s, mats.data(), mats.size() * sizeof(amp_t<T>), gates_bytes));
    return QIP_OK;
  };
  TileDesc d;
  memset(&d, 0, sizeof d);
  d.ngates = (uint32_t)gates.size();
  for (int j = 0; j < kTileHigh; ++j) d.hpos[j] = high[j];
  Ins ins = make_ins(high, 0);  // make_ins sorts its own copy; `high` keeps the tile-bit order
  const uint64_t ntiles = 1ull << (s->n - kTileBits);
  const size_t lds = sizeof(amp_t<T>) << kTileBits;
  const TileGate<T>* dg = nullptr;  // device addresses: valid only after the upload (the arena may grow / move)
  const amp_t<T>* dmats = nullptr;
  ProfRec rec;
  rec.cls = KC_TILE_GATES;
  auto begin = [&]() -> int {  // descriptors up, then the timed region starts
    QCHK(upload_gates());
    dg = (const TileGate<T>*)s->arena;
    dmats = (const amp_t<T>*)((const char*)s->arena + gates_bytes);
    if (s->profile) QCHK(prof_begin(s, KC_TILE_GATES, 2.0 * (double)s->amp_bytes * (double)s->namps, &rec));
    return QIP_OK;
  };
  if (s->tile_passes && s->tile_jit) {
    // the segment as its own kernel: nothing to upload, the descriptors are constants of the code
    hipFunction_t fn = nullptr;
    QCHK(jit_get_kernel(s, tile_jit_source<T>(plan, ins, use_nt(s)), &fn));
    if (s->jit_prepare) return QIP_OK;
    if (s->profile) QCHK(prof_begin(s, KC_TILE_GATES, 2.0 * (double)s->amp_bytes * (double)s->namps, &rec));
    void* st_ptr = s->cur;
    void* args[] = {&st_ptr};
    const dim3 grid = grid2d(ntiles, 1);
    HIPCHK(hipModuleLaunchKernel(fn, grid.x, grid.y, 1, kTileBlock, 1, 1, (unsigned)lds, s->stream, args, nullptr));
    if (s->profile) QCHK(prof_end(s, &rec));
    return QIP_OK;
  }
  if (s->jit_prepare) return QIP_OK;
  if (s->tile_passes) {
    QCHK(begin());
#define TP(NTV) hipLaunchKernelGGL((k_tile_passes<T, NTV>), grid2d(ntiles, 1), dim3(kTileBlock), lds, s->stream, \
                                   (amp_t<T>*)s->cur, ins, pd, dg, dmats)
    if (use_nt(s)) TP(true);
    else TP(false);
#undef TP
  } else {
    QCHK(begin());
    if (use_nt(s))
      hipLaunchKernelGGL((k_tile_gates<T, true>), grid2d(ntiles, 1), dim3(kBlock), lds, s->stream,
                         (amp_t<T>*)s->cur, ins, d, dg);
    else
      hipLaunchKernelGGL((k_tile_gates<T, false>), grid2d(ntiles, 1), dim3(kBlock), lds, s->stream,
                         (amp_t<T>*)s->cur, ins, d, dg);
  }
  HIPCHK(hipGetLastError());
  if (s->profile) QCHK(prof_end(s, &rec));
  return QIP_OK;
}

// One step of a tiled schedule: a segment of >= 2 gates applied in one sweep, or a single op applied by
// its own kernel (not tileable, or alone — a lone gate's own kernel touches only what can change).
struct TileStep {
  std::vector<uint64_t> ops;   // indices into the circuit, in application order
  std::vector<uint32_t> high;  // the free bit positions the segment claimed (<= kTileHigh)
  std::vector<uint32_t> perm;  // non-empty: the ops are a run of uncontrolled Swap ops applied as ONE bit-permutation
                               // sweep, new[j] = old[src(j)], bit perm[d] of src(j) = bit d of j (launch_permute)
};

// Pure host scheduling (no device, no launches).  Invariants, checked by tests/test_host_ops.py through
// qip_hip_plan_tiles: every op appears in exactly one step; an op only overtakes ops it commutes with (on every
// shared bit both only test it); without `reorder` only when it, or every op it overtakes, is rounding-free.
static int schedule_tiles(int dtype, uint32_t n, const qip_op* ops, uint64_t count, bool reorder,
                          std::vector<TileItem>* items_out, std::vector<TileStep>* steps, bool allow_2q = true,
                          bool allow_permute = true) {
  std::vector<TileItem>& items = *items_out;
  items.assign(count, TileItem());
  for (uint64_t i = 0; i < count; ++i) {
    int rc = classify_tile_item(dtype, n, &ops[i], &items[i]);
    if (rc != QIP_OK) {
      std::string msg = g_last_error;
      return fail(rc, "op %llu: %s", (unsigned long long)i, msg.c_str());
    }
    if (items[i].kind >= 3 && !allow_2q) items[i].tileable = false;  // k_tile_gates has no 2- / 3-qubit form
  }
  std::vector<char> done(count, 0);
  uint64_t head = 0;
  const uint64_t window = 256;
  while (head < count) {
    if (done[head]) {
      ++head;
      continue;
    }
    // A run of uncontrolled Swap ops that one segment cannot hold (more than kTileHigh of the moved positions lie above
    // the tile's fixed low bits — QFT's closing bit reversal is 15 transpositions over all 30 positions) composes to one
    // permutation of the index bits and goes as ONE out-of-place sweep.  Swaps only move amplitudes, so this is
    // bit-identical to applying them one by one.  Ops of the run that an earlier segment already hoisted are skipped:
    // the hoist was only legal because they commute with everything in between.
    if (allow_permute && !items[head].swap_pairs.empty()) {
      TileStep run;
      run.perm.resize(n);
      for (uint32_t b = 0; b < n; ++b) run.perm[b] = b;
      uint64_t moved = 0, j = head;
      for (; j < count; ++j) {
        if (done[j]) continue;
        if (items[j].swap_pairs.empty()) break;
        std::vector<uint32_t> tau(n);
        for (uint32_t b = 0; b < n; ++b) tau[b] = b;
        for (const auto& pr : items[j].swap_pairs) {
          tau[pr.first] = pr.second;
          tau[pr.second] = pr.first;
          moved |= (1ull << pr.first) | (1ull << pr.second);
        }
        std::vector<uint32_t> next(n);
        for (uint32_t b = 0; b < n; ++b) next[b] = run.perm[tau[b]];  // out2[j] = out1[tau(j)] = in[pi(tau(j))]
        run.perm = next;
        run.ops.push_back(j);
      }
      if (run.ops.size() >= 2 && __builtin_popcountll(moved >> kTileLow) > kTileHigh) {
        for (uint64_t i : run.ops) done[i] = 1;
        steps->push_back(run);
        continue;
      }
    }
    if (!items[head].tileable) {
      steps->push_back(TileStep{{head}, {}, {}});
      done[head++] = 1;
      continue;
    }
    // Grow a segment from `head`, scanning ahead.  A later gate may join over the gates skipped so far only if
    // it commutes with each of them — on every bit they share, both gates only TEST the bit (control or diagonal
    // target), neither exchanges amplitudes across it — and, unless `reorder` (which accepts rounding-level
    // differences), the commutation is EXACT: the gate itself, or every skipped gate, is rounding-free
    // (entries in {0, +-1, +-i}: X, Y, Z, S, CNOT, CZ, Toffoli, SWAP ...).  Exact commutations leave every
    // amplitude's sequence of rounded operations unchanged, so the result stays IEEE-equal to circuit order.
    TileStep st;
    uint64_t blocked_nd = 0, blocked_d = 0;  // bits the skipped gates exchange across / only test
    bool skipped_inexact = false;            // some skipped gate rounds
    bool any_skipped = false;
    size_t exch_gates = 0;  // gates that may open a pass (kTileMaxExchGates bounds the pass table)
    for (uint64_t i = head; i < count && i <= head + (any_skipped ? window : count) && st.ops.size() < (size_t)kTileMaxGates; ++i) {
      if (done[i]) continue;
      const TileItem& it = items[i];
      const bool commutes = !(it.nd_mask & (blocked_nd | blocked_d)) && !(it.d_mask & blocked_nd);
      bool fits = it.tileable && commutes && (reorder || it.exact || !skipped_inexact) &&
                  (it.kind == 1 || exch_gates < (size_t)kTileMaxExchGates);
      std::vector<uint32_t> need;
      if (fits) {
        // only bits the gate exchanges amplitudes across must be tile bits: a dense target, both swap bits;
        // controls and diagonal targets may stay outside (block-uniform predicates)
        std::vector<uint32_t> exch;
        if (it.kind == 0) exch = {it.t0};
        if (it.kind == 2 || it.kind == 3) exch = {it.t0, it.t1};
        if (it.kind == 4) exch = {it.t0, it.t1, it.t2};
        for (uint32_t p : exch)
          if (p >= (uint32_t)kTileLow && std::find(st.high.begin(), st.high.end(), p) == st.high.end() &&
              std::find(need.begin(), need.end(), p) == need.end())
            need.push_back(p);
        fits = st.high.size() + need.size() <= (size_t)kTileHigh;
      }
      if (fits) {
        for (uint32_t p : need) st.high.push_back(p);
        st.ops.push_back(i);
        done[i] = 1;
        exch_gates += it.kind != 1;
      } else {
        blocked_nd |= it.nd_mask;
        blocked_d |= it.d_mask;
        skipped_inexact = skipped_inexact || !it.exact;
        any_skipped = true;
      }
    }
    steps->push_back(st);
  }
  return QIP_OK;
}

// ---------------------------------------------------------------------------------------
// Qubit relabelling above the tile sweeps (option "tile_relabel"; `mode` bit 2 in the host-only hooks).
//
// A tile always holds index bits 0..5 (that is what makes its rows contiguous), so six of its eleven bits are spent on
// whatever qubits happen to live there.  With a logical -> physical map of the bit positions the scheduler decides who
// lives there: at the end of every segment the tile's eleven qubits are rearranged — in-tile bit swaps riding along in the
// same sweep — so that the six whose next amplitude-exchanging use comes soonest sit on positions 0..5 (Belady's rule),
// and the next segment spends its five free positions on five OTHER qubits.  An uncontrolled Swap op costs nothing at all:
// it only exchanges two labels.  One bit-permutation sweep at the end puts every qubit back where the caller expects it.
// Everything added is a pure move of amplitudes and every gate keeps its place in the order of the plain schedule, so the
// result is bit-identical to tile = 1 / 2 without relabelling (and, for tile = 1, to the gate-by-gate path).
// configs[1] at n = 30 (256 gates): 19 -> 13 + 1 sweeps; 1024 gates: 70 -> 47 + 1.
// ---------------------------------------------------------------------------------------
struct TileSchedule {
  const qip_op* circuit = nullptr;  // what the steps' op numbers index: the caller's array, or `owned`
  uint64_t count = 0;
  std::vector<qip_op> owned;               // relabelled: the caller's ops under the labels in force when they run + inserted swaps
  std::deque<std::vector<uint64_t>> idx;   // their index lists
  std::vector<int64_t> origin;             // relabelled: position in the caller's circuit, -1 = inserted swap
  std::vector<TileItem> items;             // one per entry of `circuit`
  std::vector<TileStep> steps;
  uint64_t absorbed = 0, inserted = 0;     // Swap ops turned into label exchanges / in-tile swaps added
};

static int schedule_tiles_relabel(int dtype, uint32_t n, const qip_op* ops, uint64_t count, bool reorder, bool allow_2q,
                                  TileSchedule* out) {
  std::vector<TileItem> L(count);  // the caller's ops, logical bit positions
  for (uint64_t i = 0; i < count; ++i) {
    int rc = classify_tile_item(dtype, n, &ops[i], &L[i]);
    if (rc != QIP_OK) {
      std::string msg = g_last_error;
      return fail(rc, "op %llu: %s", (unsigned long long)i, msg.c_str());
    }
    if (L[i].kind >= 3 && !allow_2q) L[i].tileable = false;
  }
  std::vector<uint32_t> phys(n);  // phys[p] = physical position of logical bit position p
  for (uint32_t p = 0; p < n; ++p) phys[p] = p;
  auto push_op = [&](const qip_op& o, int64_t origin) -> int {
    out->owned.push_back(o);
    out->origin.push_back(origin);
    out->items.emplace_back();
    int rc = classify_tile_item(dtype, n, &out->owned.back(), &out->items.back());
    if (rc == QIP_OK && out->items.back().kind >= 3 && !allow_2q) out->items.back().tileable = false;
    return rc;
  };
  // the caller's op i under the labels in force now: same descriptor, qubit indices mapped through `phys`
  auto emit = [&](uint64_t i, uint64_t* at) -> int {
    qip_op o = ops[i];
    out->idx.emplace_back(o.n_indices);
    std::vector<uint64_t>& v = out->idx.back();
    for (uint32_t j = 0; j < o.n_indices; ++j) v[j] = (uint64_t)(n - 1 - phys[n - 1 - (uint32_t)o.indices[j]]);
    o.indices = v.data();
    *at = out->owned.size();
    return push_op(o, (int64_t)i);
  };
  auto emit_swap = [&](uint32_t pa, uint32_t pb, uint64_t* at) -> int {  // physical positions
    qip_op o;
    memset(&o, 0, sizeof o);
    o.kind = QIP_OP_SWAP;
    o.n_indices = 2;
    out->idx.emplace_back(std::vector<uint64_t>{(uint64_t)(n - 1 - pa), (uint64_t)(n - 1 - pb)});
    o.indices = out->idx.back().data();
    *at = out->owned.size();
    return push_op(o, -1);
  };
  auto absorb = [&](const TileItem& it) {  // an uncontrolled Swap: the two qubits trade places by name
    for (const auto& pr : it.swap_pairs) std::swap(phys[pr.first], phys[pr.second]);
    out->absorbed += 1;
  };
  std::vector<char> done(count, 0);
  uint64_t head = 0;
  const uint64_t window = 256;
  const size_t max_circuit_ops = (size_t)kTileMaxGates - (size_t)kTileLow;  // room for the segment's closing swaps
  while (head < count) {
    if (done[head]) {
      ++head;
      continue;
    }
    if (!L[head].swap_pairs.empty()) {  // everything before it is done; ops hoisted over it share no qubit with it
      absorb(L[head]);
      done[head++] = 1;
      continue;
    }
    if (!L[head].tileable) {
      uint64_t at = 0;
      QCHK(emit(head, &at));
      out->steps.push_back(TileStep{{at}, {}, {}});
      done[head++] = 1;
      continue;
    }
    // the segment: schedule_tiles' rules on the logical masks (commutation does not depend on names), the tile test on
    // the physical positions
    TileStep st;
    uint64_t blocked_nd = 0, blocked_d = 0;
    bool skipped_inexact = false, any_skipped = false;
    size_t joined = 0, exch_gates = 0;
    for (uint64_t i = head; i < count && i <= head + (any_skipped ? window : count) && joined < max_circuit_ops; ++i) {
      if (done[i]) continue;
      const TileItem& it = L[i];
      if (!any_skipped && !it.swap_pairs.empty()) {  // in circuit order, nothing pending before it: a label exchange
        absorb(it);
        done[i] = 1;
        continue;
      }
      const bool commutes = !(it.nd_mask & (blocked_nd | blocked_d)) && !(it.d_mask & blocked_nd);
      bool fits = it.tileable && commutes && (reorder || it.exact || !skipped_inexact) &&
                  (it.kind == 1 || exch_gates < (size_t)kTileMaxExchGates - (size_t)kTileLow);  // (room for the closing swaps)
      std::vector<uint32_t> need;
      if (fits) {
        std::vector<uint32_t> exch;
        if (it.kind == 0) exch = {it.t0};
        if (it.kind == 2 || it.kind == 3) exch = {it.t0, it.t1};
        if (it.kind == 4) exch = {it.t0, it.t1, it.t2};
        for (uint32_t p : exch) {
          const uint32_t pp = phys[p];
          if (pp >= (uint32_t)kTileLow && std::find(st.high.begin(), st.high.end(), pp) == st.high.end() &&
              std::find(need.begin(), need.end(), pp) == need.end())
            need.push_back(pp);
        }
        fits = st.high.size() + need.size() <= (size_t)kTileHigh;
      }
      if (fits) {
        for (uint32_t pp : need) st.high.push_back(pp);
        uint64_t at = 0;
        QCHK(emit(i, &at));
        st.ops.push_back(at);
        done[i] = 1;
        joined += 1;
        exch_gates += it.kind != 1;
      } else {
        blocked_nd |= it.nd_mask;
        blocked_d |= it.d_mask;
        skipped_inexact = skipped_inexact || !it.exact;
        any_skipped = true;
      }
    }
    // Who should live on positions 0..5 next?  Next amplitude-exchanging use of every qubit (ops not done yet, circuit
    // order).  First spend the tile's unclaimed free positions on the soonest-needed qubits outside the tile (they can
    // then be brought down as well), then bring the soonest-needed of the tile's qubits down, evicting the ones needed
    // last.  A lone gate keeps its own kernel (it touches only what can change): no swaps for it.
    if (st.ops.size() >= 2) {
      std::vector<uint64_t> nxt(n, ~0ull);
      {
        uint32_t found = 0;
        for (uint64_t i = head; i < count && found < n; ++i) {
          if (done[i] || !L[i].tileable) continue;
          uint64_t m = L[i].nd_mask;
          while (m) {
            const uint32_t p = (uint32_t)__builtin_ctzll(m);
            m &= m - 1;
            if (nxt[p] == ~0ull) {
              nxt[p] = i;
              ++found;
            }
          }
        }
      }
      auto in_tile = [&](uint32_t pp) { return pp < (uint32_t)kTileLow || std::find(st.high.begin(), st.high.end(), pp) != st.high.end(); };
      std::vector<uint32_t> by_use;  // logical positions with a future use, soonest first
      for (uint32_t p = 0; p < n; ++p)
        if (nxt[p] != ~0ull) by_use.push_back(p);
      std::stable_sort(by_use.begin(), by_use.end(), [&](uint32_t a, uint32_t b) { return nxt[a] < nxt[b]; });
      for (uint32_t p : by_use) {
        if (st.high.size() >= (size_t)kTileHigh) break;
        if (!in_tile(phys[p])) st.high.push_back(phys[p]);
      }
      std::vector<uint32_t> tile_log;
      for (uint32_t p = 0; p < n; ++p)
        if (in_tile(phys[p])) tile_log.push_back(p);
      std::stable_sort(tile_log.begin(), tile_log.end(), [&](uint32_t a, uint32_t b) { return nxt[a] < nxt[b]; });
      std::vector<uint32_t> bring, evict;
      for (size_t r = 0; r < tile_log.size(); ++r) {
        const uint32_t p = tile_log[r];
        const bool wanted = r < (size_t)kTileLow && nxt[p] != ~0ull;
        if (wanted && phys[p] >= (uint32_t)kTileLow) bring.push_back(p);
        if (!wanted && phys[p] < (uint32_t)kTileLow) evict.push_back(p);
      }
      std::reverse(evict.begin(), evict.end());  // needed last (or never) goes first
      for (size_t r = 0; r < bring.size() && r < evict.size() && st.ops.size() < (size_t)kTileMaxGates; ++r) {
        uint64_t at = 0;
        QCHK(emit_swap(phys[bring[r]], phys[evict[r]], &at));
        st.ops.push_back(at);
        std::swap(phys[bring[r]], phys[evict[r]]);
        out->inserted += 1;
      }
    }
    out->steps.push_back(st);
  }
  // every qubit back to the position the caller expects: final index bit d takes the bit that lives on phys[d] now
  bool identity = true;
  for (uint32_t p = 0; p < n; ++p) identity = identity && phys[p] == p;
  if (!identity) {
    TileStep back;
    back.perm = phys;
    out->steps.push_back(back);
  }
  out->circuit = out->owned.data();
  out->count = out->owned.size();
  return QIP_OK;
}

// mode: bits 0-1 = the "tile" option (1 = circuit order, 2 = commuting reorder), bit 2 = relabel the qubits when that
// shortens the plan, bit 3 = relabel unconditionally
static int make_tile_schedule(int dtype, uint32_t n, const qip_op* ops, uint64_t count, int mode, bool allow_2q, TileSchedule* out,
                              bool allow_permute = true) {
  const bool reorder = (mode & 3) >= 2;
  if ((mode & 4) && allow_permute) {
    // relabelling pays for random circuits; layered ones (Grover's X / H walls, QFT) gain nothing and would only pay the
    // closing permutation: schedule both ways (host work, microseconds per gate) and keep the shorter plan
    QCHK(schedule_tiles_relabel(dtype, n, ops, count, reorder, allow_2q, out));
    TileSchedule plain;
    QCHK(schedule_tiles(dtype, n, ops, count, reorder, &plain.items, &plain.steps, allow_2q));
    if (out->steps.size() < plain.steps.size() || (mode & 8)) return QIP_OK;  // bit 3: keep it regardless (tests)
    *out = TileSchedule();
  }
  out->circuit = ops;
  out->count = count;
  return schedule_tiles(dtype, n, ops, count, reorder, &out->items, &out->steps, allow_2q, allow_permute);
}

extern "C" int qip_hip_plan_tiles(int dtype, uint32_t n, const qip_op* ops, uint64_t count, int mode,
                                  int64_t* step_of_op, uint64_t* n_steps) try {
  if ((count && (!ops || !step_of_op)) || !n_steps) return fail(QIP_ERR_INVALID, "null argument");
  if (dtype != QIP_C64 && dtype != QIP_C32) return fail(QIP_ERR_INVALID, "bad dtype %d", dtype);
  if (n < (uint32_t)kTileBits) return fail(QIP_ERR_UNSUPPORTED, "tile sweeps need n >= %d", kTileBits);
  TileSchedule sc;
  QCHK(make_tile_schedule(dtype, n, ops, count, mode, true, &sc));
  for (uint64_t i = 0; i < count; ++i) step_of_op[i] = -1;  // relabelled: an absorbed Swap op belongs to no step
  for (size_t si = 0; si < sc.steps.size(); ++si)
    for (uint64_t i : sc.steps[si].ops) {
      const int64_t o = sc.origin.empty() ? (int64_t)i : sc.origin[i];
      if (o >= 0) step_of_op[o] = (int64_t)si;
    }
  *n_steps = sc.steps.size();
  return QIP_OK;
} QIP_CATCH_ALL

// Host-only: the complete tile plan of a circuit as JSON (schedule, and for every multi-gate step the segment
// plan of build_tile_segment).  Test infrastructure for the host half of the tile path: tests replay the plan
// on the CPU with a numpy model of k_tile_passes and compare with the oracle, no GPU involved.
template <typename T>
static int tile_plan_json(int dtype, uint32_t n, const qip_op* ops, uint64_t count, int mode, std::string* out) {
  TileSchedule sc;
  QCHK(make_tile_schedule(dtype, n, ops, count, mode, true, &sc));
  const std::vector<TileItem>& items = sc.items;
  const std::vector<TileStep>& steps = sc.steps;
  char buf[256];
  auto num = [&](double v) {
    snprintf(buf, sizeof buf, "%.17g", v);
    return std::string(buf);
  };
  std::string& js = *out;
  js = "{\"n\":" + std::to_string(n);
  if (!sc.origin.empty()) {  // relabelled: the circuit the steps index (origin = the caller's op, -1 = inserted swap; qubit indices)
    js += ",\"absorbed\":" + std::to_string(sc.absorbed) + ",\"inserted\":" + std::to_string(sc.inserted) + ",\"circuit\":[";
    for (uint64_t i = 0; i < sc.count; ++i) {
      js += std::string(i ? "," : "") + "{\"o\":" + std::to_string(sc.origin[i]) + ",\"i\":[";
      for (uint32_t j = 0; j < sc.circuit[i].n_indices; ++j) js += (j ? "," : "") + std::to_string(sc.circuit[i].indices[j]);
      js += "]}";
    }
    js += "]";
  }
  js += ",\"steps\":[";
  for (size_t si = 0; si < steps.size(); ++si) {
    const TileStep& st = steps[si];
    if (si) js += ",";
    js += "{\"ops\":[";
    for (size_t k = 0; k < st.ops.size(); ++k) js += (k ? "," : "") + std::to_string(st.ops[k]);
    js += "]";
    if (!st.perm.empty()) {
      js += ",\"perm\":[";
      for (size_t k = 0; k < st.perm.size(); ++k) js += (k ? "," : "") + std::to_string(st.perm[k]);
      js += "]";
    } else if (st.ops.size() > 1) {
      std::vector<const TileItem*> seg;
      for (uint64_t i : st.ops) seg.push_back(&items[i]);
      TileSegmentPlan<T> plan;
      QCHK(build_tile_segment<T>(n, true, seg, st.high, &plan));
      js += ",\"high\":[";
      for (size_t k = 0; k < plan.high.size(); ++k) js += (k ? "," : "") + std::to_string(plan.high[k]);
      js += "],\"passes\":[";
      for (uint32_t pi = 0; pi < plan.pd.npasses; ++pi) {
        const TilePass& ps = plan.pd.pass[pi];
        if (pi) js += ",";
        js += "{\"first\":" + std::to_string(ps.first) + ",\"count\":" + std::to_string(ps.count) + ",\"pb\":[" +
              std::to_string(ps.pb[0]) + "," + std::to_string(ps.pb[1]) + "," + std::to_string(ps.pb[2]) +
              "],\"lanepos\":[";
        for (int k = 0; k < kTileLaneBits; ++k) js += (k ? "," : "") + std::to_string((unsigned)((ps.lanepos >> (4 * k)) & 15ull));
        js += "]}";
      }
      js += "],\"gates\":[";
      for (size_t gi = 0; gi < plan.gates.size(); ++gi) {
        const TileGate<T>& g = plan.gates[gi];
        if (gi) js += ",";
        js += "{\"kind\":" + std::to_string(g.kind) + ",\"op\":" + std::to_string(g.op) + ",\"b0\":" +
              std::to_string(g.b0) + ",\"b1\":" + std::to_string(g.b1) + ",\"cmask\":" + std::to_string(g.cmask) +
              ",\"cm_reg\":" + std::to_string(g.cm_reg) + ",\"cm_lane\":" + std::to_string(g.cm_lane) +
              ",\"omask\":" + std::to_string(g.omask) + ",\"tpos_out\":" + std::to_string(g.tpos_out) +
              ",\"nz\":" + std::to_string(g.nz) + ",\"m\":[";
        for (int e = 0; e < 4; ++e)
          js += std::string(e ? "," : "") + "[" + num((double)g.m[e].x) + "," + num((double)g.m[e].y) + "]";
        js += "]}";
      }
      js += "],\"mats\":[";
      for (size_t e = 0; e < plan.mats.size(); ++e)
        js += std::string(e ? "," : "") + "[" + num((double)plan.mats[e].x) + "," + num((double)plan.mats[e].y) + "]";
      js += "]";
    }
    js += "}";
  }
  js += "]}";
  return QIP_OK;
}

extern "C" const char* qip_hip_debug_tile_plan(int dtype, uint32_t n, const qip_op* ops, uint64_t count, int mode) {
  static thread_local std::string json;
  try {
    if (count && !ops) return fail(QIP_ERR_INVALID, "null op array"), nullptr;
    if (dtype != QIP_C64 && dtype != QIP_C32) return fail(QIP_ERR_INVALID, "bad dtype %d", dtype), nullptr;
    if (n < (uint32_t)kTileBits) return fail(QIP_ERR_UNSUPPORTED, "tile sweeps need n >= %d", kTileBits), nullptr;
    const int rc = dtype == QIP_C64 ? tile_plan_json<double>(dtype, n, ops, count, mode, &json)
                                    : tile_plan_json<float>(dtype, n, ops, count, mode, &json);
    return rc == QIP_OK ? json.c_str() : nullptr;
  } catch (const std::exception& e) {
    fail(QIP_ERR_INVALID, "internal error: %s", e.what());
    return nullptr;
  }
}

// Host-only test hook: the run-time-compiled source of every multi-gate step of a circuit's tile schedule, each
// compiled with hiprtc (no device needed: hiprtc cross-compiles for gfx950).  Returns the number of segments compiled
// and the total source / code size through the out parameters; `first_source` (may be NULL) receives a pointer to the
// first segment's source text (owned by the library, valid until the calling thread's next call).
template <typename T>
static int debug_jit_t(int dtype, uint32_t n, const qip_op* ops, uint64_t count, int mode, uint64_t* nseg, uint64_t* src_bytes,
                       uint64_t* code_bytes, std::string* first) {
  TileSchedule sc;
  QCHK(make_tile_schedule(dtype, n, ops, count, mode, true, &sc));
  const std::vector<TileItem>& items = sc.items;
  *nseg = *src_bytes = *code_bytes = 0;
  for (const TileStep& st : sc.steps) {
    if (st.ops.size() < 2 || !st.perm.empty()) continue;
    std::vector<const TileItem*> seg;
    for (uint64_t i : st.ops) seg.push_back(&items[i]);
    TileSegmentPlan<T> plan;
    QCHK(build_tile_segment<T>(n, true, seg, st.high, &plan));
    Ins ins = make_ins(plan.high, 0);
    const std::string src = tile_jit_source<T>(plan, ins, true);
    std::vector<char> code;
    QCHK(hiprtc_compile(src, &code));
    if (*nseg == 0 && first) *first = src;
    *nseg += 1;
    *src_bytes += src.size();
    *code_bytes += code.size();
  }
  return QIP_OK;
}

extern "C" int qip_hip_debug_tile_jit(int dtype, uint32_t n, const qip_op* ops, uint64_t count, int mode, uint64_t* segments,
                                      uint64_t* source_bytes, uint64_t* code_bytes, const char** first_source) try {
  static thread_local std::string first;
  if ((count && !ops) || !segments || !source_bytes || !code_bytes) return fail(QIP_ERR_INVALID, "null argument");
  if (dtype != QIP_C64 && dtype != QIP_C32) return fail(QIP_ERR_INVALID, "bad dtype %d", dtype);
  if (n < (uint32_t)kTileBits) return fail(QIP_ERR_UNSUPPORTED, "tile sweeps need n >= %d", kTileBits);
  first.clear();
  QCHK(dtype == QIP_C64 ? debug_jit_t<double>(dtype, n, ops, count, mode, segments, source_bytes, code_bytes, &first)
                        : debug_jit_t<float>(dtype, n, ops, count, mode, segments, source_bytes, code_bytes, &first));
  if (first_source) *first_source = first.c_str();
  return QIP_OK;
} QIP_CATCH_ALL

extern "C" int qip_hip_tile_bits(void) { return kTileBits; }

static int tile_mode_of(const qip_hip_state* s) {  // option tile_relabel: 1 = when it shortens the plan, 2 = always
  return (int)std::min<int64_t>(s->tile, 2) | (s->tile_relabel ? 4 : 0) | (s->tile_relabel >= 2 ? 8 : 0);
}

template <typename T>
static int apply_ops_tiled(qip_hip_state* s, const qip_op* ops_in, uint64_t count, bool /*reorder*/) {
  TileSchedule sc;
  QCHK(make_tile_schedule(s->dtype, s->n, ops_in, count, tile_mode_of(s), s->tile_passes != 0, &sc));
  {
    // a bit-permutation sweep is out of place: get the second buffer BEFORE the first gate runs; if HBM cannot hold it
    // (a state above half of the 288 GB), fall back to the plan without permutation sweeps instead of failing half way
    bool permutes = false;
    for (const TileStep& st : sc.steps) permutes = permutes || !st.perm.empty();
    if (permutes && !s->jit_prepare && !s->alt && ensure_alt(s) != QIP_OK) {
      (void)hipGetLastError();
      sc = TileSchedule();
      QCHK(make_tile_schedule(s->dtype, s->n, ops_in, count, tile_mode_of(s), s->tile_passes != 0, &sc, /*allow_permute=*/false));
    }
  }
  const qip_op* ops = sc.circuit;
  const std::vector<TileItem>& items = sc.items;
  for (const TileStep& st : sc.steps) {
    if (!st.perm.empty()) {  // a run of Swap ops as one bit-permutation sweep
      if (s->jit_prepare) continue;
      QCHK(launch_permute(s, st.perm.data()));
      continue;
    }
    if (st.ops.size() == 1) {
      QCHK(apply_op_t<T>(s, &ops[st.ops[0]]));
      continue;
    }
    std::vector<const TileItem*> seg;
    for (uint64_t i : st.ops) seg.push_back(&items[i]);
    QCHK(launch_tile_segment<T>(s, seg, st.high));
  }
  return QIP_OK;
}

extern "C" int qip_hip_state_apply_ops(qip_hip_state* s, const qip_op* ops, uint64_t count) try {
  STATE_ENTER(s);
  if (count && !ops) return fail(QIP_ERR_INVALID, "null op array");
  if (s->tile >= 1 && !s->force_generic && !g_force_generic && s->n >= (uint32_t)kTileBits)
    return s->dtype == QIP_C64 ? apply_ops_tiled<double>(s, ops, count, s->tile >= 2)
                               : apply_ops_tiled<float>(s, ops, count, s->tile >= 2);
  if (s->fuse >= 2 && !s->force_generic && !g_force_generic) {
    const uint32_t K = (uint32_t)std::min<int64_t>(s->fuse, kMaxMfmaK);  // both precisions have a matrix-core k = 5 kernel
    if (s->n >= K + 4)
      return s->dtype == QIP_C64 ? apply_ops_fused<double>(s, ops, count, K) : apply_ops_fused<float>(s, ops, count, K);
  }
  for (uint64_t i = 0; i < count; ++i) {
    int rc = s->dtype == QIP_C64 ? apply_op_t<double>(s, &ops[i]) : apply_op_t<float>(s, &ops[i]);
    if (rc != QIP_OK) {
      std::string msg = g_last_error;
      return fail(rc, "op %llu: %s", (unsigned long long)i, msg.c_str());
    }
  }
  return QIP_OK;
} QIP_CATCH_ALL

// ---------------------------------------------------------------------------------------
// programs: a circuit captured once into a hipGraph and replayed with one launch
// ---------------------------------------------------------------------------------------
struct qip_hip_program {
  qip_hip_state* s = nullptr;
  const qip_op* ops = nullptr;
  uint64_t count = 0;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  void* captured_cur = nullptr;
  uint64_t captured_arena_gen = 0;  // the graph's memcpy / kernel nodes hold arena addresses
  std::deque<std::vector<char>> staging;  // payloads the graph's memcpy nodes read at every replay
  int last_was_graph = 0;
};

static void program_drop_graph(qip_hip_program* p);
static void programs_orphan(qip_hip_state* s) {
  for (qip_hip_program* p : s->programs) {
    program_drop_graph(p);
    p->s = nullptr;
  }
  s->programs.clear();
}

static void program_drop_graph(qip_hip_program* p) {
  if (p->exec) (void)hipGraphExecDestroy(p->exec);
  if (p->graph) (void)hipGraphDestroy(p->graph);
  p->exec = nullptr;
  p->graph = nullptr;
  p->staging.clear();
}

// Try to capture; on any obstacle leave the program in eager mode (exec == nullptr) and report success.
static int program_capture(qip_hip_program* p) {
  qip_hip_state* s = p->s;
  program_drop_graph(p);
  if (s->force_generic || g_force_generic || s->profile) return QIP_OK;
  // an op on the out-of-place path would swap the buffers under the graph: stay eager
  for (uint64_t i = 0; i < p->count; ++i) {
    FlatOp f;
    QCHK(flatten_op(s->n, &p->ops[i], false, &f));
    Plan pl;
    QCHK(make_plan(s->dtype, s->n, f, false, &pl));
    const bool f64 = s->dtype == QIP_C64;
    const uint32_t k = f.n_op;
    const bool reg_or_mfma = pl.cls == KC_GATE_KQ && (k <= kMaxRegK || (s->mfma && k <= kMaxBigK && s->n >= f.k_all + 4));
    (void)f64;
    if (pl.cls == KC_GATHER_GENERIC || (pl.cls == KC_GATE_KQ && !reg_or_mfma)) return QIP_OK;
  }
  if (s->tile >= 1 && s->n >= (uint32_t)kTileBits) {  // a bit-permutation sweep is out of place too
    TileSchedule sc;
    QCHK(make_tile_schedule(s->dtype, s->n, p->ops, p->count, tile_mode_of(s), s->tile_passes != 0, &sc));
    for (const TileStep& st : sc.steps)
      if (!st.perm.empty()) return QIP_OK;
  }
  if (s->tile >= 1 && s->tile_jit) {  // run-time compilation cannot happen inside a stream capture: do it now
    s->jit_prepare = true;
    const int rc = qip_hip_state_apply_ops(s, p->ops, p->count);
    s->jit_prepare = false;
    QCHK(rc);
  }
  for (int attempt = 0; attempt < 3; ++attempt) {
    s->capture_arena_need = 0;
    if (hipStreamBeginCapture(s->stream, hipStreamCaptureModeRelaxed) != hipSuccess) {
      (void)hipGetLastError();
      return QIP_OK;
    }
    s->capture_staging = &p->staging;
    int rc = qip_hip_state_apply_ops(s, p->ops, p->count);
    s->capture_staging = nullptr;
    hipGraph_t g = nullptr;
    const hipError_t e = hipStreamEndCapture(s->stream, &g);
    if (rc == QIP_OK && e == hipSuccess && g) {
      if (hipGraphInstantiate(&p->exec, g, nullptr, nullptr, 0) == hipSuccess) {
        p->graph = g;
        p->captured_cur = s->cur;
        p->captured_arena_gen = s->arena_gen;
        return QIP_OK;
      }
      (void)hipGetLastError();
      (void)hipGraphDestroy(g);
      p->exec = nullptr;
      p->staging.clear();
      return QIP_OK;
    }
    if (g) (void)hipGraphDestroy(g);
    (void)hipGetLastError();
    p->staging.clear();
    if (s->capture_arena_need > s->arena_cap) {  // grow outside the capture, then retry
      const size_t need = s->capture_arena_need;
      s->capture_arena_need = 0;
      QCHK(ensure_arena(s, need));
      continue;
    }
    if (rc != QIP_OK && rc != QIP_ERR_UNSUPPORTED) return rc;  // a real descriptor error
    return QIP_OK;
  }
  return QIP_OK;
}

extern "C" int qip_hip_program_create(qip_hip_state* s, const qip_op* ops, uint64_t count,
                                      qip_hip_program** out) try {
  STATE_ENTER(s);
  if (!out || (count && !ops)) return fail(QIP_ERR_INVALID, "null argument");
  for (uint64_t i = 0; i < count; ++i) {
    FlatOp f;
    int rc = flatten_op(s->n, &ops[i], false, &f);
    if (rc != QIP_OK) {
      std::string msg = g_last_error;
      return fail(rc, "op %llu: %s", (unsigned long long)i, msg.c_str());
    }
  }
  qip_hip_program* p = new qip_hip_program();
  p->s = s;
  p->ops = ops;
  p->count = count;
  int rc = program_capture(p);
  if (rc != QIP_OK) {
    delete p;
    return rc;
  }
  s->programs.push_back(p);
  *out = p;
  return QIP_OK;
} QIP_CATCH_ALL

extern "C" int qip_hip_program_run(qip_hip_program* p) try {
  if (!p) return fail(QIP_ERR_INVALID, "null program");
  qip_hip_state* s = p->s;
  if (!s) return fail(QIP_ERR_INVALID, "the state this program was recorded against has been destroyed");
  STATE_ENTER(s);
  if (p->exec && (p->captured_cur != s->cur || p->captured_arena_gen != s->arena_gen || s->profile || s->force_generic ||
                  g_force_generic)) {
    if (s->profile || s->force_generic || g_force_generic) program_drop_graph(p);
    else QCHK(program_capture(p));  // the state moved to its other buffer, or the arena was re-allocated (an eager op
                                    // needed a larger payload): the recorded addresses are stale, re-record
  }
  if (p->exec) {
    HIPCHK(hipGraphLaunch(p->exec, s->stream));
    p->last_was_graph = 1;
    return QIP_OK;
  }
  p->last_was_graph = 0;
  return qip_hip_state_apply_ops(s, p->ops, p->count);
} QIP_CATCH_ALL

extern "C" int qip_hip_program_is_graph(const qip_hip_program* p) try { return p ? p->last_was_graph : 0; } QIP_CATCH_ALL

extern "C" int qip_hip_program_destroy(qip_hip_program* p) try {
  if (!p) return QIP_OK;
  if (p->s) {
    (void)hipSetDevice(p->s->device);
    (void)hipStreamSynchronize(p->s->stream);
    auto& v = p->s->programs;
    v.erase(std::remove(v.begin(), v.end(), p), v.end());
  }
  program_drop_graph(p);
  delete p;
  return QIP_OK;
} QIP_CATCH_ALL

// ---------------------------------------------------------------------------------------
// host-pointer twin of apply_op / apply_op_overwrite
// ---------------------------------------------------------------------------------------
template <typename T>
static int apply_op_host_t(int dtype, uint32_t n, const qip_op* op, const void* in, uint64_t in_len,
                           void* out, uint64_t out_len, uint64_t in_off, uint64_t out_off,
                           int accumulate) {
  const uint64_t N = 1ull << n;
  const size_t ab = sizeof(amp_t<T>);
  FlatOp f;
  QCHK(flatten_op(n, op, false, &f));
  if (out_len == 0) return QIP_OK;
  qip_hip_state* s = nullptr;
  QCHK(qip_hip_state_create(n, dtype, 0, &s));
  int rc = QIP_OK;
  auto body = [&]() -> int {
    const bool full = in_off == 0 && out_off == 0 && in_len == N && out_len == N;
    if (full) {
      QCHK(qip_hip_state_upload(s, in, 0, N));
      QCHK(apply_op_t<T>(s, op));
      if (accumulate) {
        QCHK(ensure_alt(s));
        HIPCHK(hipMemcpyAsync(s->alt, out, N * ab, hipMemcpyHostToDevice, s->stream));
        hipLaunchKernelGGL((k_add_into<T>), dim3(grid_stride(N)), dim3(kBlock), 0, s->stream,
                           (amp_t<T>*)s->alt, (const amp_t<T>*)s->cur, N);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(out, s->alt, N * ab, hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
      } else {
        QCHK(qip_hip_state_download(s, out, 0, N));
      }
      return QIP_OK;
    }
    // windowed: literal gather kernel on dedicated buffers
    void *d_in = nullptr, *d_out = nullptr;
    HIPCHK(hipMalloc(&d_in, std::max<size_t>(in_len * ab, 16)));
    hipError_t e = hipMalloc(&d_out, out_len * ab);
    if (e != hipSuccess) {
      (void)hipFree(d_in);
      return fail(QIP_ERR_DEVICE, "hipMalloc failed: %s", hipGetErrorString(e));
    }
    auto inner = [&]() -> int {
      if (in_len) HIPCHK(hipMemcpyAsync(d_in, in, in_len * ab, hipMemcpyHostToDevice, s->stream));
      if (accumulate) HIPCHK(hipMemcpyAsync(d_out, out, out_len * ab, hipMemcpyHostToDevice, s->stream));
      QCHK(launch_gather<T>(s, f, (const amp_t<T>*)d_in, in_len, (amp_t<T>*)d_out, out_len, in_off,
                            out_off, accumulate));
      HIPCHK(hipMemcpyAsync(out, d_out, out_len * ab, hipMemcpyDeviceToHost, s->stream));
      HIPCHK(hipStreamSynchronize(s->stream));
      return QIP_OK;
    };
    int r2 = inner();
    (void)hipStreamSynchronize(s->stream);
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    return r2;
  };
  rc = body();
  std::string keep = g_last_error;
  qip_hip_state_destroy(s);
  if (rc != QIP_OK) g_last_error = keep;
  return rc;
}

extern "C" int qip_hip_apply_op_host(int dtype, uint32_t n, const qip_op* op, const void* in,
                                     uint64_t in_len, void* out, uint64_t out_len, uint64_t in_off,
                                     uint64_t out_off, int accumulate) try {
  if (dtype != QIP_C64 && dtype != QIP_C32) return fail(QIP_ERR_INVALID, "bad dtype %d", dtype);
  if ((in_len && !in) || (out_len && !out)) return fail(QIP_ERR_INVALID, "null buffer");
  if (n == 0 || n > 40) return fail(QIP_ERR_INVALID, "n = %u out of range [1, 40]", n);
  return dtype == QIP_C64
             ? apply_op_host_t<double>(dtype, n, op, in, in_len, out, out_len, in_off, out_off, accumulate)
             : apply_op_host_t<float>(dtype, n, op, in, in_len, out, out_len, in_off, out_off, accumulate);
} QIP_CATCH_ALL

// ---------------------------------------------------------------------------------------
// host-pointer twins of apply_op_row and of the windowed measurement functions
// ---------------------------------------------------------------------------------------
extern "C" int qip_hip_apply_op_row_host(int dtype, uint32_t n, const qip_op* op, const void* in, uint64_t in_len,
                                         uint64_t outputrow, uint64_t in_off, uint64_t out_off, void* out_value) try {
  if (!out_value) return fail(QIP_ERR_INVALID, "null output");
  // apply_op_row (matrix_ops.rs:38-59): the value of row out_off + outputrow = a one-row output window there
  return qip_hip_apply_op_host(dtype, n, op, in, in_len, out_value, 1, in_off, out_off + outputrow, 0);
} QIP_CATCH_ALL

template <typename T>
static int measure_probs_host_t(uint32_t n, const uint64_t* indices, uint32_t k, const void* in, uint64_t in_len,
                                uint64_t in_off, double* out) {
  if (k == 0 || k > n || k > 26 || !indices) return fail(QIP_ERR_INVALID, "bad measurement index list");
  MeasDesc md;
  memset(&md, 0, sizeof md);
  md.k = k;
  uint64_t seen = 0;
  for (uint32_t i = 0; i < k; ++i) {
    if (indices[i] >= n) return fail(QIP_ERR_INVALID, "measured qubit index out of range");
    if (seen & (1ull << indices[i])) return fail(QIP_ERR_INVALID, "repeated measured qubit index");
    seen |= 1ull << indices[i];
    md.mpos[i] = (uint32_t)(n - 1 - indices[i]);
  }
  const uint64_t outcomes = 1ull << k;
  for (uint64_t m = 0; m < outcomes; ++m) out[m] = 0.0;
  if (in_len == 0) return QIP_OK;
  if (qip_hip_device_count() <= 0) return fail(QIP_ERR_NO_DEVICE, "no HIP device visible: qip_hip has no CPU fallback");
  HIPCHK(hipSetDevice(0));
  void *d_in = nullptr, *d_out = nullptr;
  HIPCHK(hipMalloc(&d_in, in_len * sizeof(amp_t<T>)));
  hipError_t e = hipMalloc(&d_out, outcomes * sizeof(double));
  if (e == hipSuccess) e = hipMemcpy(d_in, in, in_len * sizeof(amp_t<T>), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemset(d_out, 0, outcomes * sizeof(double));
  if (e == hipSuccess) {
    hipLaunchKernelGGL((k_measure_probs_scatter<T>), dim3(grid_stride(in_len)), dim3(kBlock), 0, nullptr, (const amp_t<T>*)d_in,
                       in_len, md, in_off, (double*)d_out);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(out, d_out, outcomes * sizeof(double), hipMemcpyDeviceToHost);
  (void)hipFree(d_in);
  if (d_out) (void)hipFree(d_out);
  if (e != hipSuccess) return fail(QIP_ERR_DEVICE, "windowed measure_probs failed: %s", hipGetErrorString(e));
  return QIP_OK;
}

extern "C" int qip_hip_measure_probs_host(int dtype, uint32_t n, const uint64_t* indices, uint32_t k, const void* in,
                                          uint64_t in_len, uint64_t in_off, double* out) try {
  if (dtype != QIP_C64 && dtype != QIP_C32) return fail(QIP_ERR_INVALID, "bad dtype %d", dtype);
  if ((in_len && !in) || !out) return fail(QIP_ERR_INVALID, "null buffer");
  if (n == 0 || n > 62) return fail(QIP_ERR_INVALID, "n = %u out of range [1, 62]", n);
  if (in_off > (1ull << n) || in_len > (1ull << n) - in_off) return fail(QIP_ERR_INVALID, "window outside the 2^n vector");
  return dtype == QIP_C64 ? measure_probs_host_t<double>(n, indices, k, in, in_len, in_off, out)
                          : measure_probs_host_t<float>(n, indices, k, in, in_len, in_off, out);
} QIP_CATCH_ALL

extern "C" int qip_hip_measure_prob_host(int dtype, uint32_t n, uint64_t measured, const uint64_t* indices, uint32_t k,
                                         const void* in, uint64_t in_len, uint64_t in_off, double* out) try {
  if (!out) return fail(QIP_ERR_INVALID, "null output");
  if (k > 26) return fail(QIP_ERR_UNSUPPORTED, "windowed measure_prob over %u qubits", k);
  if (k > 0 && (measured >> k) != 0) return fail(QIP_ERR_INVALID, "measured value has more than k bits");
  std::vector<double> probs(1ull << k);
  QCHK(qip_hip_measure_probs_host(dtype, n, indices, k, in, in_len, in_off, probs.data()));
  *out = probs[measured];
  return QIP_OK;
} QIP_CATCH_ALL

// ---------------------------------------------------------------------------------------
// measurement
// ---------------------------------------------------------------------------------------
static int check_measure_indices(qip_hip_state* s, const uint64_t* indices, uint32_t k, MeasDesc* md,
                                 std::vector<uint32_t>* pos) {
  if (k == 0 || k > s->n || !indices) return fail(QIP_ERR_INVALID, "bad measurement index list");
  memset(md, 0, sizeof *md);
  md->k = k;
  uint64_t seen = 0;
  for (uint32_t i = 0; i < k; ++i) {
    if (indices[i] >= s->n) return fail(QIP_ERR_INVALID, "measured qubit index out of range");
    if (seen & (1ull << indices[i])) return fail(QIP_ERR_INVALID, "repeated measured qubit index");
    seen |= 1ull << indices[i];
    md->mpos[i] = (uint32_t)(s->n - 1 - indices[i]);
    pos->push_back(md->mpos[i]);
  }
  return QIP_OK;
}

template <typename T>
static int chunk_norms(qip_hip_state* s, uint64_t* chunk_out, std::vector<double>* sums) {
  uint64_t chunk = std::max<uint64_t>(s->namps / 4096, 1024);
  chunk = std::min<uint64_t>(chunk, s->namps);
  const uint64_t nchunks = (s->namps + chunk - 1) / chunk;
  QCHK(ensure_partial(s, nchunks));
  hipLaunchKernelGGL((k_chunk_norms<T>), dim3((unsigned)nchunks), dim3(kBlock), 0, s->stream,
                     (const amp_t<T>*)s->cur, s->namps, chunk, s->d_partial);
  HIPCHK(hipGetLastError());
  sums->resize(nchunks);
  HIPCHK(hipMemcpyAsync(sums->data(), s->d_partial, nchunks * sizeof(double), hipMemcpyDeviceToHost,
                        s->stream));
  HIPCHK(hipStreamSynchronize(s->stream));
  *chunk_out = chunk;
  return QIP_OK;
}

extern "C" int qip_hip_state_norm_sqr(qip_hip_state* s, double* out) try {
  STATE_ENTER(s);
  if (!out) return fail(QIP_ERR_INVALID, "null output");
  std::vector<double> sums;
  uint64_t chunk = 0;
  QCHK(s->dtype == QIP_C64 ? chunk_norms<double>(s, &chunk, &sums) : chunk_norms<float>(s, &chunk, &sums));
  double t = 0;
  for (double v : sums) t += v;
  *out = t;
  return QIP_OK;
} QIP_CATCH_ALL

template <typename T>
static int measure_probs_t(qip_hip_state* s, const MeasDesc& md, const std::vector<uint32_t>& pos,
                           uint64_t m_first, uint64_t m_count, double* out) {
  const uint32_t k = md.k;
  if (m_count == (1ull << k) && k <= 4) {
    // all outcomes of a few qubits: one coalesced pass, 2^k running sums per lane
    const unsigned gx = (unsigned)std::min<uint64_t>(std::max<uint64_t>(s->namps / (kBlock * 8), 1), 2048);
    const size_t M = (size_t)1 << k;
    QCHK(ensure_partial(s, (size_t)gx * M));
    const amp_t<T>* st = (const amp_t<T>*)s->cur;
    switch (k) {
      case 1: hipLaunchKernelGGL((k_measure_probs_small<T, 1>), dim3(gx), dim3(kBlock), 0, s->stream, st, s->namps, md, s->d_partial); break;
      case 2: hipLaunchKernelGGL((k_measure_probs_small<T, 2>), dim3(gx), dim3(kBlock), 0, s->stream, st, s->namps, md, s->d_partial); break;
      case 3: hipLaunchKernelGGL((k_measure_probs_small<T, 3>), dim3(gx), dim3(kBlock), 0, s->stream, st, s->namps, md, s->d_partial); break;
      default: hipLaunchKernelGGL((k_measure_probs_small<T, 4>), dim3(gx), dim3(kBlock), 0, s->stream, st, s->namps, md, s->d_partial); break;
    }
    HIPCHK(hipGetLastError());
    std::vector<double> part((size_t)gx * M);
    HIPCHK(hipMemcpyAsync(part.data(), s->d_partial, part.size() * sizeof(double), hipMemcpyDeviceToHost,
                          s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    for (size_t m = 0; m < M; ++m) {
      double t = 0;
      for (unsigned b = 0; b < gx; ++b) t += part[(size_t)b * M + m];
      out[m] = t;
    }
    return QIP_OK;
  }
  if (m_count == (1ull << k)) {
    // more outcomes, no atomics: measured positions >= 8 on the grid (a few of them walked per lane when the grid
    // would be huge), the others resolved per lane (k_measure_probs_grid)
    MeasGridDesc gd;
    memset(&gd, 0, sizeof gd);
    std::vector<std::pair<uint32_t, uint32_t>> high;  // (position, outcome bit) of the measured positions >= 8
    uint32_t lbit[8];
    for (uint32_t i = 0; i < k; ++i) {
      if (md.mpos[i] >= 8) {
        high.push_back({md.mpos[i], i});
      } else {
        lbit[gd.kl] = i;
        gd.lpos[gd.kl++] = md.mpos[i];
      }
    }
    std::sort(high.begin(), high.end());
    // step bits: the lowest high positions, as many (<= 3) as it takes to bring the grid down to ~2^11 outcomes
    uint32_t ki = 0;
    while (ki < 3 && high.size() - ki > 11) ++ki;
    uint32_t sbit[3] = {0, 0, 0}, gbit[kMaxIns];
    std::vector<uint32_t> opened;
    for (uint32_t i = 0; i < high.size(); ++i) {
      opened.push_back(high[i].first);
      if (i < ki) {
        gd.spos[i] = high[i].first;
        sbit[i] = high[i].second;
      } else {
        gbit[gd.kg] = high[i].second;
        gd.gpos[gd.kg++] = high[i].first;
      }
    }
    if (gd.kg <= 20) {
      Ins ins = make_ins(opened, 0);
      const uint64_t count = 1ull << (s->n - (uint32_t)high.size());  // indices per (grid outcome, step value)
      const uint64_t ny = 1ull << gd.kg, nl = 1ull << gd.kl, nc = 1ull << ki;
      // about 8192 blocks in all, each with at least four 4-KiB rows when the outcome has that many
      uint64_t gx = std::max<uint64_t>(8192 / ny, 1);
      gx = std::min<uint64_t>(gx, std::max<uint64_t>(count * nc / (kBlock * 4), 1));
      const size_t nout = (size_t)(ny * nc * nl), np = nout * (size_t)gx;
      QCHK(ensure_partial(s, np + nout));
      const dim3 grid((unsigned)(ny * gx));
#define MG(KI) hipLaunchKernelGGL((k_measure_probs_grid<T, KI>), grid, dim3(kBlock), 0, s->stream, (const amp_t<T>*)s->cur, \
                                  count, ins, gd, (uint32_t)gx, (uint64_t)nout, s->d_partial)
      switch (ki) {
        case 0: MG(0); break;
        case 1: MG(1); break;
        case 2: MG(2); break;
        default: MG(3); break;
      }
#undef MG
      HIPCHK(hipGetLastError());
      const double* res = s->d_partial;
      if (gx > 1) {  // fold the gx partials per outcome on the device: only 2^k doubles cross PCIe
        hipLaunchKernelGGL(k_sum_partials, dim3(grid_for(nout, kBlock / 64)), dim3(kBlock), 0, s->stream, s->d_partial, (uint32_t)gx,
                           (uint64_t)nout, s->d_partial + np);
        HIPCHK(hipGetLastError());
        res = s->d_partial + np;
      }
      std::vector<double> part(nout);
      HIPCHK(hipMemcpyAsync(part.data(), res, nout * sizeof(double), hipMemcpyDeviceToHost, s->stream));
      HIPCHK(hipStreamSynchronize(s->stream));
      for (uint64_t o = 0; o < nout; ++o) {
        const uint64_t l = o & (nl - 1), c = (o >> gd.kl) & (nc - 1), mg = o >> (gd.kl + ki);
        uint64_t m = 0;
        for (uint32_t i = 0; i < gd.kg; ++i) m |= ((mg >> i) & 1ull) << gbit[i];
        for (uint32_t i = 0; i < ki; ++i) m |= ((c >> i) & 1ull) << sbit[i];
        for (uint32_t i = 0; i < gd.kl; ++i) m |= ((l >> i) & 1ull) << lbit[i];
        out[m] = part[o];
      }
      return QIP_OK;
    }
  }
  if (m_count == 1) {
    // one outcome: sum over the sub-space whose measured bits read m (measure_prob_fn :65-112)
    Ins ins = make_ins(pos, 0);
    const uint64_t count = 1ull << (s->n - k);
    const unsigned gx = (unsigned)std::min<uint64_t>(std::max<uint64_t>(count / (kBlock * 8), 1), 1024);
    QCHK(ensure_partial(s, (size_t)gx));
    hipLaunchKernelGGL((k_measure_probs<T>), dim3(gx, 1), dim3(kBlock), 0, s->stream,
                       (const amp_t<T>*)s->cur, count, ins, md, m_first, s->d_partial);
    HIPCHK(hipGetLastError());
    std::vector<double> part((size_t)gx);
    HIPCHK(hipMemcpyAsync(part.data(), s->d_partial, part.size() * sizeof(double), hipMemcpyDeviceToHost,
                          s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    double t = 0;
    for (unsigned b = 0; b < gx; ++b) t += part[b];
    out[0] = t;
    return QIP_OK;
  }
  // many outcomes: scatter-add |amp|^2 into a device table of 2^k doubles
  const uint64_t outcomes = 1ull << k;
  QCHK(ensure_partial(s, outcomes));
  HIPCHK(hipMemsetAsync(s->d_partial, 0, outcomes * sizeof(double), s->stream));
  hipLaunchKernelGGL((k_measure_probs_scatter<T>), dim3(grid_stride(s->namps)), dim3(kBlock), 0,
                     s->stream, (const amp_t<T>*)s->cur, s->namps, md, (uint64_t)0, s->d_partial);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(out, s->d_partial, outcomes * sizeof(double), hipMemcpyDeviceToHost, s->stream));
  HIPCHK(hipStreamSynchronize(s->stream));
  return QIP_OK;
}

extern "C" int qip_hip_state_measure_probs(qip_hip_state* s, const uint64_t* indices, uint32_t k,
                                           double* out) try {
  STATE_ENTER(s);
  if (!out) return fail(QIP_ERR_INVALID, "null output");
  MeasDesc md;
  std::vector<uint32_t> pos;
  QCHK(check_measure_indices(s, indices, k, &md, &pos));
  if (k > 30) return fail(QIP_ERR_UNSUPPORTED, "measure_probs over %u qubits", k);
  return s->dtype == QIP_C64 ? measure_probs_t<double>(s, md, pos, 0, 1ull << k, out)
                             : measure_probs_t<float>(s, md, pos, 0, 1ull << k, out);
} QIP_CATCH_ALL

extern "C" int qip_hip_state_measure_prob(qip_hip_state* s, uint64_t measured, const uint64_t* indices,
                                          uint32_t k, double* out) try {
  STATE_ENTER(s);
  if (!out) return fail(QIP_ERR_INVALID, "null output");
  MeasDesc md;
  std::vector<uint32_t> pos;
  QCHK(check_measure_indices(s, indices, k, &md, &pos));
  if (k < 64 && (measured >> k) != 0) return fail(QIP_ERR_INVALID, "measured value has more than k bits");
  return s->dtype == QIP_C64 ? measure_probs_t<double>(s, md, pos, measured, 1, out)
                             : measure_probs_t<float>(s, md, pos, measured, 1, out);
} QIP_CATCH_ALL

static int collapse(qip_hip_state* s, const MeasDesc& md, uint64_t m, double p);

// soft_measure (measurement_ops.rs:153-176): first index at which r - Σ|amp|² <= 0.  The
// device sums contiguous chunks; the host walks the chunk sums, then replays the
// reference's sequential loop inside the one chunk that crosses zero.
template <typename T>
static int soft_measure_t(qip_hip_state* s, const MeasDesc& md, double rand_u01, uint64_t* measured) {
  // Two device passes: per-chunk sums of |amp|^2, then k_find_crossing inside the chunk(s) that may bring the
  // running remainder to <= 0.  The host only walks the few thousand chunk sums; no amplitude leaves HBM.
  std::vector<double> sums;
  uint64_t chunk = 0;
  QCHK(chunk_norms<T>(s, &chunk, &sums));
  QCHK(ensure_partial(s, 2));
  double r = (double)(T)rand_u01;
  uint64_t measured_indx = 0;
  for (size_t c = 0; c < sums.size(); ++c) {
    if (r - sums[c] > 1e-9 * (1.0 + sums[c])) {  // clearly past this chunk
      r -= sums[c];
      continue;
    }
    const uint64_t lo = (uint64_t)c * chunk, len = std::min<uint64_t>(chunk, s->namps - lo);
    hipLaunchKernelGGL((k_find_crossing<T>), dim3(1), dim3(kBlock), 0, s->stream, (const amp_t<T>*)s->cur, lo,
                       len, r, (uint64_t*)s->d_partial);
    HIPCHK(hipGetLastError());
    uint64_t res[2];
    HIPCHK(hipMemcpyAsync(res, s->d_partial, sizeof res, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    if (res[0] != ~0ull) {
      measured_indx = res[0];
      break;
    }
    memcpy(&r, &res[1], sizeof r);  // the chunk was scanned without crossing: carry its exact remainder on
  }
  // never crossing: the reference leaves measured_indx = 0 (:166)
  uint64_t m = 0;
  for (uint32_t i = 0; i < md.k; ++i) m |= ((measured_indx >> md.mpos[i]) & 1ull) << i;
  *measured = m;
  return QIP_OK;
}

extern "C" int qip_hip_state_soft_measure(qip_hip_state* s, const uint64_t* indices, uint32_t k,
                                          double rand_u01, uint64_t* measured) try {
  STATE_ENTER(s);
  if (!measured) return fail(QIP_ERR_INVALID, "null output");
  MeasDesc md;
  std::vector<uint32_t> pos;
  QCHK(check_measure_indices(s, indices, k, &md, &pos));
  return s->dtype == QIP_C64 ? soft_measure_t<double>(s, md, rand_u01, measured)
                             : soft_measure_t<float>(s, md, rand_u01, measured);
} QIP_CATCH_ALL

extern "C" int qip_hip_state_measure(qip_hip_state* s, const uint64_t* indices, uint32_t k,
                                     int64_t forced, double rand_u01, uint64_t* measured, double* prob) try {
  STATE_ENTER(s);
  if (!measured || !prob) return fail(QIP_ERR_INVALID, "null output");
  MeasDesc md;
  std::vector<uint32_t> pos;
  QCHK(check_measure_indices(s, indices, k, &md, &pos));
  uint64_t m = 0;
  if (forced >= 0) {
    m = (uint64_t)forced;
    if (k < 64 && (m >> k) != 0) return fail(QIP_ERR_INVALID, "forced outcome has more than k bits");
  } else {
    QCHK(s->dtype == QIP_C64 ? soft_measure_t<double>(s, md, rand_u01, &m)
                             : soft_measure_t<float>(s, md, rand_u01, &m));
  }
  double p = 0;
  QCHK(s->dtype == QIP_C64 ? measure_probs_t<double>(s, md, pos, m, 1, &p)
                           : measure_probs_t<float>(s, md, pos, m, 1, &p));
  *measured = m;
  *prob = p;
  return collapse(s, md, m, p);
} QIP_CATCH_ALL

static int collapse(qip_hip_state* s, const MeasDesc& md, uint64_t m, double p) {
  if (p == 0.0) return QIP_OK;  // measure_state is a no-op (:230)
  uint64_t row_mask = 0, measured_mask = 0;
  for (uint32_t i = 0; i < md.k; ++i) {
    row_mask |= 1ull << md.mpos[i];
    measured_mask |= ((m >> i) & 1ull) << md.mpos[i];
  }
  if (s->dtype == QIP_C64) {
    const double p_mult = 1.0 / std::sqrt(p);
    hipLaunchKernelGGL((k_collapse<double>), dim3(grid_stride(s->namps)), dim3(kBlock), 0, s->stream,
                       (amp_t<double>*)s->cur, s->namps, row_mask, measured_mask, p_mult);
  } else {
    const float p_mult = 1.0f / std::sqrt((float)p);
    hipLaunchKernelGGL((k_collapse<float>), dim3(grid_stride(s->namps)), dim3(kBlock), 0, s->stream,
                       (amp_t<float>*)s->cur, s->namps, row_mask, measured_mask, p_mult);
  }
  HIPCHK(hipGetLastError());
  return QIP_OK;
}

extern "C" int qip_hip_state_measure_state(qip_hip_state* s, const uint64_t* indices, uint32_t k,
                                           uint64_t measured, double prob) try {
  STATE_ENTER(s);
  MeasDesc md;
  memset(&md, 0, sizeof md);
  std::vector<uint32_t> pos;
  if (k > 0) QCHK(check_measure_indices(s, indices, k, &md, &pos));
  if (k < 64 && (measured >> k) != 0) return fail(QIP_ERR_INVALID, "measured value has more than k bits");
  if (!(prob >= 0.0)) return fail(QIP_ERR_INVALID, "probability must be >= 0");
  return collapse(s, md, measured, prob);
} QIP_CATCH_ALL

// ---------------------------------------------------------------------------------------
// the state sharded over several GPUs (qip_hip_dist_*)
// ---------------------------------------------------------------------------------------
#include "qip_dist.inc"
