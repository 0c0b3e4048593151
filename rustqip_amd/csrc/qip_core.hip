// qip_core.hip — C ABI (include/qip_hip.h) over the gfx950 kernels in qip_kernels.h: errors, options, op validation,
// kernel choice, state handles, profiling.  (The other translation units: qip_internal.h.)
//
// Host side of the drop-in boundary: validates op descriptors the way the reference's
// constructors do (qip/src/state_ops/matrix_ops.rs:12-122), classifies each op into the
// cheapest kernel that is result-identical to the reference's gather formulation
// (qip-iterators/src/matrix_ops.rs:62-152), and launches it on the handle's HIP stream.
// There is NO CPU fallback: without a HIP device every compute entry point fails with
// QIP_ERR_NO_DEVICE.
#include "qip_tile.h"

thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}


extern "C" const char* qip_hip_last_error(void) { return g_last_error.c_str(); }
extern "C" int qip_hip_abi_version(void) { return 8; }  // 8: + qip_hip_apply_op_device and the real / integer element types QIP_F64 / F32 / I64 / I32 of the slice-level calls, jit_stats folded into jit_stats2; 7: programs own device-resident payloads and record out-of-place ops (one graph per buffer parity), jit counters + background_segments / disk_trimmed, option jit_disk_cap_mb, debug hooks in qip_hip_debug.h; 6: + jit_stats2 / jit_set_cache_dir / jit_cache_dir / jit_compile_file, options jit_disk_cache / jit_procs / tile_auto; 5: + state_download_indices, copy_from completes before it returns; 4: + state_copy_from, state_max_abs_diff, dist stats v2 (rccl_ranks), options tile_fma / jit cache bound
extern "C" int qip_hip_device_count(void) try {
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return c;
} QIP_CATCH_ALL

int64_t g_force_generic = 0;
// Selector bits below this position stay in the grid as a per-lane predicate (whole lines are swept); see kLineBits.
uint32_t g_line_bits = qipk::kLineBits;
// Tile sweeps, measured on MI355X at n = 30 (tools/tune_tile.hip, profiles/r03_tile_skeleton.md): the time of a light sweep is set
// by WHICH five high positions the tile holds — 5.2 ms for {11..15}, 6.3 ms for {6..10}, 6.7 ms for the top five — not by the
// block structure (persistent / prefetching variants are no faster).  Free positions a segment does not need are therefore
// taken from 11 upwards, and the two lowest of the five are the wave bits (both worth ~1 % on the benchmark circuits).
int64_t g_tile_pad_from = 11, g_tile_wave_rule = 1, g_tile_remap = 0, g_tile_sched = 1;
// r4: the tile's sixth low position (qip_tile.h tile_p5): 11 = every wave-level access of a tile sweep is two 512-byte halves
// 32 KiB apart, 5 = one contiguous 1-KiB row (rounds 1-3).  Measured (tools/tune_tile probe, profiles/r04_tile_rows.md): the
// split form brings EVERY choice of the five high positions to 5.3 - 5.9 ms per light sweep at n = 30 (contiguous: 5.3 - 8.6).
int64_t g_tile_row_split = 11, g_tile_row_split_f32 = 5;
int64_t g_single_via_tile = 3, g_single_via_tile_f32 = 3;
int64_t g_force_k4_direct = 0;
int64_t g_collective_timeout_s = 120;  // global option "collective_timeout_s" (qip_dist.hip: how long a rank waits for an exchange)  // row bits of k_permute_bits for 16-byte elements: 0 = by the permutation, 5 / 6 = forced (tuning aid)
// Options (include/qip_hip.h lists the product's).  The measured alternatives of earlier rounds — each one a code path that lost
// its A/B run (profiles/r0*_*.md) — are fixed at their defaults in the product build; a build with -DQIP_HIP_TUNING
// (QIP_HIP_TUNING=1 python -m rustqip_amd.build) makes them switchable again for the tools/ bench scripts.
extern "C" int qip_hip_set_global_option(const char* key, int64_t value) try {
  if (!key) return fail(QIP_ERR_INVALID, "null option key");
  if (!strcmp(key, "force_generic")) { g_force_generic = value; return QIP_OK; }
  if (!strcmp(key, "jit_cache_cap")) return jit_set_cache_cap(value);
  if (!strcmp(key, "jit_disk_cap_mb")) return jit_set_disk_cap_mb(value);
  if (!strcmp(key, "jit_disk_cache")) { g_jit_disk = value != 0; return QIP_OK; }
  if (!strcmp(key, "jit_procs")) {
    if (value < 0 || value > 64) return fail(QIP_ERR_INVALID, "jit_procs must be 0 (automatic) .. 64");
    g_jit_procs = value;
    return QIP_OK;
  }
  if (!strcmp(key, "tile_sched")) { g_tile_sched = value; return QIP_OK; }
  if (!strcmp(key, "single_via_tile")) { g_single_via_tile = value; return QIP_OK; }
  if (!strcmp(key, "dist_fold_pack")) { g_dist_fold_pack = value; return QIP_OK; }
  if (!strcmp(key, "dist_plan_cost")) { g_dist_plan_cost = value != 0; return QIP_OK; }
  if (!strcmp(key, "collective_timeout_s")) {
    if (value < 0) return fail(QIP_ERR_INVALID, "collective_timeout_s must be >= 0 (0 = wait for ever)");
    g_collective_timeout_s = value;
    return QIP_OK;
  }
#ifdef QIP_HIP_TUNING
  if (!strcmp(key, "line_bits")) {
    if (value < 0 || value > 3) return fail(QIP_ERR_INVALID, "line_bits must be 0..3");
    g_line_bits = (uint32_t)value;
    return QIP_OK;
  }
  if (!strcmp(key, "tile_pad_from")) { g_tile_pad_from = value; return QIP_OK; }
  if (!strcmp(key, "soft_measure_one_pass")) { g_soft_measure_one_pass = value != 0; return QIP_OK; }
  if (!strcmp(key, "tile_wide_dense3_inline")) { g_tile_wide_dense3_inline = value != 0; return QIP_OK; }
  if (!strcmp(key, "tile_wide_pin")) { g_tile_wide_pin = value != 0; return QIP_OK; }
  if (!strcmp(key, "sparse_tile")) { g_sparse_tile = value != 0; return QIP_OK; }
  if (!strcmp(key, "debug_slice_sweeps")) {
    if (value != 0 && value != 2 && value != 4 && value != 8) return fail(QIP_ERR_INVALID, "debug_slice_sweeps is 0, 2, 4 or 8");
    g_debug_slice_sweeps = value;
    return QIP_OK;
  }
  if (!strcmp(key, "tile_diag_runs")) { g_tile_diag_runs = value != 0; return QIP_OK; }
  if (!strcmp(key, "jit_threads")) {
    if (value < 1 || value > 64) return fail(QIP_ERR_INVALID, "jit_threads must be 1..64");
    g_jit_threads = value;
    return QIP_OK;
  }
  if (!strcmp(key, "tile_row_split_f32")) {
    if (value != 5 && value != 12) return fail(QIP_ERR_INVALID, "tile_row_split_f32 is 12 (split rows) or 5 (contiguous rows)");
    g_tile_row_split_f32 = value;
    return QIP_OK;
  }
  if (!strcmp(key, "tile_row_split")) {
    if (value != 5 && value != 11) return fail(QIP_ERR_INVALID, "tile_row_split is 11 (split rows) or 5 (contiguous rows)");
    g_tile_row_split = value;
    return QIP_OK;
  }
  if (!strcmp(key, "tile_wave_rule")) { g_tile_wave_rule = value; return QIP_OK; }
  if (!strcmp(key, "tile_remap")) { g_tile_remap = value; return QIP_OK; }
  if (!strcmp(key, "k4_direct")) { g_force_k4_direct = value; return QIP_OK; }
  if (!strcmp(key, "single_via_tile_f32")) { g_single_via_tile_f32 = value; return QIP_OK; }
#endif
  return fail(QIP_ERR_INVALID, "unknown global option '%s'", key);
} QIP_CATCH_ALL

// ---------------------------------------------------------------------------------------
// op flattening + validation
// ---------------------------------------------------------------------------------------

// strict = what make_*_op rejects; always = what would panic / read out of bounds in the
// reference kernel.
int flatten_op(uint32_t n, const qip_op* op, bool strict, FlatOp* f) {
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
static const char* kKernelClassNames[KC_COUNT] = {
    "k_gate1q_pair", "k_gate1q_xlane", "k_phase",          "k_diag",           "k_diag1q",
    "k_swap_bits",   "k_gate_kq",      "k_gate_kq_mfma",   "k_tile_passes",    "k_gather_generic",
    "noop_identity", "k_sparse_kq", "k_gate_big_mfma", "k_permute_bits", "k_sparse_ell", "k_sparse_tile", "tile_sweep_parts", "k_dense_small"};

extern "C" int qip_hip_kernel_class_count(void) { return KC_COUNT; }
extern "C" const char* qip_hip_kernel_class_name(int cls) {
  return (cls >= 0 && cls < KC_COUNT) ? kKernelClassNames[cls] : "";
}

// ---------------------------------------------------------------------------------------
// planning: which kernel applies an op
// ---------------------------------------------------------------------------------------

template <typename T>
static void read_dense(const void* dense, uint64_t count, std::vector<double>* out) {
  const T* p = static_cast<const T*>(dense);
  out->resize(count * 2);
  for (uint64_t i = 0; i < count * 2; ++i) (*out)[i] = (double)p[i];
}


int make_plan(int dtype, uint32_t n, const FlatOp& f, bool force_generic, Plan* p) {
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
  if (k <= kMaxHugeK) {
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
  if (k <= kMaxHugeK) {
    // the launcher picks the matrix-core forms (k in 3..5 / 6..8 / 9..10) when >= 16 groups exist
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

int ensure_arena(qip_hip_state* s, size_t bytes) {
  if (s->capture_pool) {  // a program records: a region of its device pool per launch group (arena_begin_group)
    ProgPool* pp = s->capture_pool;
    if (s->arena && bytes <= s->arena_cap) return QIP_OK;
    if (s->arena) return fail(QIP_ERR_UNSUPPORTED, "internal: a launch group's payload region grew while a program was recorded");
    const size_t off = (pp->used + 255) & ~(size_t)255;
    pp->used = off + bytes;
    if (pp->used > pp->cap) pp->overflow = true;  // sizing pass: no malloc inside a stream capture — the caller grows the pool and records again
    s->arena = pp->overflow ? pp->base : (void*)((char*)pp->base + off);
    s->arena_cap = bytes;
    return QIP_OK;
  }
  if (bytes <= s->arena_cap) return QIP_OK;
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

int ensure_partial(qip_hip_state* s, size_t count) {
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

int ensure_alt(qip_hip_state* s) {
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
  int cus = 0;
  HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
  qip_hip_state* s = new qip_hip_state();
  if (cus > 0) s->num_cus = cus;
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
  if (s->d_ticket) (void)hipFree(s->d_ticket);
  if (s->owns_stream && s->stream) (void)hipStreamDestroy(s->stream);
  delete s;
  return QIP_OK;
} QIP_CATCH_ALL


extern "C" int qip_hip_state_init_basis(qip_hip_state* s, uint64_t index) try {
  STATE_ENTER_NOCHECK(s);
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
  s->layout.clear();  // whatever was there has been overwritten: no order of its own to restore, and a poisoned handle is
  s->poisoned = false;  // healthy again — only now that the writes have landed
  return QIP_OK;
} QIP_CATCH_ALL

extern "C" int qip_hip_state_upload(qip_hip_state* s, const void* src, uint64_t offset, uint64_t len) try {
  // a full upload overwrites whatever a failed batch left: it is the one call (with init_basis) a poisoned handle accepts.
  // The handle is declared healthy only AFTER the copy has landed: a null source or a failed copy leaves it poisoned.
  const bool heals = s && s->poisoned && offset == 0 && len == s->namps && len != 0;
  if (heals) {
    STATE_ENTER_NOCHECK(s);
  } else {
    STATE_ENTER(s);
  }
  if (offset > s->namps || len > s->namps - offset) return fail(QIP_ERR_INVALID, "upload range out of bounds");
  if (len == 0) return QIP_OK;
  if (!src) return fail(QIP_ERR_INVALID, "null source");
  HIPCHK(hipMemcpyAsync((char*)s->cur + offset * s->amp_bytes, src, len * s->amp_bytes,
                        hipMemcpyHostToDevice, s->stream));
  HIPCHK(hipStreamSynchronize(s->stream));
  if (heals) {
    s->poisoned = false;
    s->layout.clear();
  }
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
  if (!amps) return fail(QIP_ERR_INVALID, "null argument");
  STATE_ENTER(s);  // (a relabelled state is put back in the caller's order before its address is handed out)
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
  STATE_ENTER_NOCHECK(s);  // (no amplitude is addressed: a relabelled state stays as it is)
  HIPCHK(hipStreamSynchronize(s->stream));
  return QIP_OK;
} QIP_CATCH_ALL

extern "C" int qip_hip_state_set_option(qip_hip_state* s, const char* key, int64_t value) try {
  if (!s || !key) return fail(QIP_ERR_INVALID, "null argument");
  if (!strcmp(key, "force_generic")) s->force_generic = value;
  else if (!strcmp(key, "profile")) s->profile = value;
  else if (!strcmp(key, "mfma")) s->mfma = value;
  else if (!strcmp(key, "fuse")) s->fuse = value;
  else if (!strcmp(key, "tile")) s->tile = value;
  else if (!strcmp(key, "tile_jit")) {
#ifndef QIP_HIP_TUNING
    if (value != 0 && value != 1) return fail(QIP_ERR_INVALID, "tile_jit is 0 or 1");
#endif
    s->tile_jit = value;
  }
  else if (!strcmp(key, "tile_relabel")) s->tile_relabel = value;
  else if (!strcmp(key, "tile_fma")) s->tile_fma = value;
  else if (!strcmp(key, "tile_merge")) s->tile_merge = value;
  else if (!strcmp(key, "tile_wide")) s->tile_wide = value;
  else if (!strcmp(key, "tile_auto")) s->tile_auto = value;
  else if (!strcmp(key, "pair_floor")) s->pair_floor = value;
#ifdef QIP_HIP_TUNING
  else if (!strcmp(key, "lowbit_shuffle")) s->lowbit_shuffle = value;
  else if (!strcmp(key, "packed_f32")) s->packed_f32 = value;
  else if (!strcmp(key, "tile_passes")) s->tile_passes = value;
  else if (!strcmp(key, "unroll")) s->unroll = value;
  else if (!strcmp(key, "swap_single")) s->swap_single = value;
#endif
  else return fail(QIP_ERR_INVALID, "unknown option '%s'", key);
  return QIP_OK;
} QIP_CATCH_ALL

// ---- profiling ---------------------------------------------------------------------------
int prof_begin(qip_hip_state* s, int cls, double bytes, ProfRec* r) {
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
int prof_end(qip_hip_state* s, ProfRec* r) {
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
  STATE_ENTER_RAW(s);  // (no amplitude is addressed: a relabelled state stays as it is)
  if (cls < 0 || cls >= KC_COUNT) return fail(QIP_ERR_INVALID, "bad kernel class %d", cls);
  QCHK(prof_drain(s));
  if (launches) *launches = s->prof_launches[cls];
  if (total_ms) *total_ms = s->prof_ms[cls];
  if (algorithmic_bytes) *algorithmic_bytes = s->prof_bytes[cls];
  return QIP_OK;
} QIP_CATCH_ALL
extern "C" int qip_hip_state_profile_reset(qip_hip_state* s) try {
  STATE_ENTER_RAW(s);  // (no amplitude is addressed: a relabelled state stays as it is)
  QCHK(prof_drain(s));
  for (int c = 0; c < KC_COUNT; ++c) {
    s->prof_launches[c] = 0;
    s->prof_ms[c] = 0;
    s->prof_bytes[c] = 0;
  }
  return QIP_OK;
} QIP_CATCH_ALL

