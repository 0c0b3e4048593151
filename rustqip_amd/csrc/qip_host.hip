// qip_host.hip — host-pointer twins of the reference functions (parity tests call these).
#include "qip_internal.h"

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

// ---------------------------------------------------------------------------------------
// apply_op / apply_op_overwrite on DEVICE slices for any P (qip_hip_apply_op_device)
// ---------------------------------------------------------------------------------------
// qip-iterators' kernel is generic over P (matrix_ops.rs:98-107): a real or integer vector takes the same row fold as a
// complex one — acc = P::zero(); acc += val * input[col] over the iterator's columns (matrix_ops.rs:62-94, std::iter::Sum),
// then `+=` or `=` into the output row (:110 / :139).  One lane per output row, the reference's loops verbatim (index maps
// g_full_to_sub / g_sub_to_full of the complex literal kernel, the Control threshold of qubit_iterators.rs:130-169, the
// zero-skip of MatrixOpIterator :49 only), products and sums unfused (the build has -ffp-contract=off): bit-equal to the
// reference for every P.  Integers are computed in the unsigned type of their width (wrapping; the bits are two's complement).
// Algorithmic bytes per output row: sizeof(P) x (1 read of the input + 1 write, + 1 read when accumulating).
template <typename R> struct RealTab {  // a dense op on k <= 4 qubits inside the kernel arguments: no upload, no table to own
  R v[256];
};

template <typename R, int V> struct RVec { using type = R __attribute__((ext_vector_type(V))); };
template <typename R> struct RVec<R, 1> { using type = R; };

// V > 1: the lane's V consecutive rows share every index bit the op looks at (no op / control position below log2 V, windows
// and lengths multiples of V): one sub-index, one 16-byte access per term, a vector wholly inside or outside the input window
template <typename R, int V>
__device__ __forceinline__ typename RVec<R, V>::type r_term(const GatherDesc& d, uint64_t row, uint64_t col, R val, const R* __restrict__ in) {
  using X = typename RVec<R, V>::type;
  const uint64_t colbits = g_sub_to_full(d, col, row);  // matrix_ops.rs:79
  if (colbits < d.in_off) return (X)(R)0;                // :80-81
  const uint64_t vecrow = colbits - d.in_off;            // :83
  if (vecrow >= d.in_len) return (X)(R)0;                // :84-85
  return (X)val * reinterpret_cast<const X*>(in)[vecrow / V];  // :87
}

template <typename R, bool TAB, int V>
__global__ __launch_bounds__(kBlock) void k_gather_real(const R* __restrict__ in, R* __restrict__ out_, GatherDesc d,
                                                        RealTab<R> tab, const R* __restrict__ dense,
                                                        const uint64_t* __restrict__ rowptr, const uint64_t* __restrict__ cols,
                                                        const R* __restrict__ vals) {
  using X = typename RVec<R, V>::type;
  X* __restrict__ out = reinterpret_cast<X*>(out_);
  const uint64_t stride = (uint64_t)gridDim.x * kBlock, nvec = d.out_len / V;
  for (uint64_t r = (uint64_t)blockIdx.x * kBlock + threadIdx.x; r < nvec; r += stride) {
    const uint64_t row = d.out_off + r * V;
    const uint64_t matrow = g_full_to_sub(d, row);
    X acc = (X)(R)0;
    uint64_t shift = 0, irow = matrow;
    bool identity_row = false;
    if (d.n_control > 0) {
      const uint64_t thr = (1ull << (d.n_control + d.n_op)) - (1ull << d.n_op);
      if (matrow >= thr) {
        shift = thr;
        irow = matrow - thr;
      } else {
        identity_row = true;
      }
    }
    if (identity_row) {
      acc = acc + r_term<R, V>(d, row, matrow, (R)1, in);
    } else if (d.inner_kind == 0) {  // MATRIX
      const uint64_t side = 1ull << d.n_op;
      for (uint64_t c = 0; c < side; ++c) {
        const R v = TAB ? tab.v[irow * side + c] : dense[irow * side + c];
        if (!(v == (R)0)) acc = acc + r_term<R, V>(d, row, c + shift, v, in);
      }
    } else if (d.inner_kind == 1) {  // SPARSE
      for (uint64_t p = rowptr[irow]; p < rowptr[irow + 1]; ++p) acc = acc + r_term<R, V>(d, row, cols[p] + shift, vals[p], in);
    } else {  // SWAP
      const uint32_t half_n = d.n_op >> 1;
      const uint64_t lower_mask = ~(~0ull << half_n);
      const uint64_t col = ((irow & lower_mask) << half_n) + (irow >> half_n);
      acc = acc + r_term<R, V>(d, row, col + shift, (R)1, in);
    }
    out[r] = d.accumulate ? (X)(out[r] + acc) : acc;
  }
}

// ---- the whole vector (both windows [0, 2^n)), a dense op or Swap on distinct qubits with k_all <= 4 indices ------------------
// One lane owns V = 16 / sizeof(P) consecutive rows of every one of the 2^K rows of a group (K = controls + op qubits; the V
// rows differ only in index bits below every op / control position): it reads each of the group's 2^K input vectors ONCE
// (16-byte accesses), folds every output row exactly as the literal kernel does — acc = 0; acc += m[row][c] * x[c] for c
// ascending, entries equal to zero skipped (the matrix sits in the kernel arguments: the skip is a scalar branch); a row outside
// the control subspace or of a Swap is 0 + 1 * x[col] — and writes 2^K output vectors.  Bit-equal to the literal kernel; HBM
// traffic = the algorithmic bytes (the literal kernel reads every input line 2^k_op times, from different lanes).
// An op with an index bit below log2(V) takes V = 1 (8- / 4-byte accesses; a 4-byte P whose lowest index bit is position 1: V = 2).
struct RealGroupDesc {
  uint64_t nitems;      // 2^n / (2^K * V)
  uint64_t off[16];     // off[m] = the index bits of sub-index m (bit K-1-j of m at position pos[j]), in units of V rows
  int32_t accumulate;
};

template <typename R, int V, int K, int NC, bool SWAP, bool NT>
__global__ __launch_bounds__(kBlock) void k_real_groups(const R* __restrict__ in, R* __restrict__ out, Ins ins, RealGroupDesc d,
                                                        RealTab<R> tab) {
  using X = typename RVec<R, V>::type;
  constexpr int M = 1 << K, KOP = K - NC, SIDE = 1 << KOP, THR = M - SIDE;
  const uint64_t w = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (w >= d.nitems) return;
  const uint64_t base = insert_bits<K>(w, ins);
  const X* inv = reinterpret_cast<const X*>(in);
  X* outv = reinterpret_cast<X*>(out);
  X x[M];
#pragma unroll
  for (int m = 0; m < M; ++m) x[m] = ldg<NT>(inv + (base | d.off[m]));
#pragma unroll
  for (int m = 0; m < M; ++m) {
    X acc = (X)(R)0;
    if (m < THR) {  // outside the control subspace: exactly one (row, 1)  (qubit_iterators.rs:160-169)
      acc = acc + (X)(R)1 * x[m];
    } else if constexpr (SWAP) {
      constexpr int HALF = KOP >> 1;
      const int irow = m - THR;
      const int col = ((irow & ((1 << HALF) - 1)) << HALF) + (irow >> HALF);
      acc = acc + (X)(R)1 * x[col + THR];
    } else {
#pragma unroll
      for (int c = 0; c < SIDE; ++c) {
        const R v = tab.v[(m - THR) * SIDE + c];
        if (!(v == (R)0)) acc = acc + (X)v * x[c + THR];
      }
    }
    const uint64_t at = base | d.off[m];
    stg<NT>(outv + at, d.accumulate ? (X)(ldg<NT>(outv + at) + acc) : acc);
  }
}

template <typename R, int V, int K, bool NT>
static int launch_real_groups_k(const FlatOp& f, const R* d_in, R* d_out, const Ins& ins, const RealGroupDesc& d, const RealTab<R>& tab,
                                hipStream_t stream) {
  const dim3 grid((unsigned)((d.nitems + kBlock - 1) / kBlock)), block(kBlock);
  const bool swap = f.inner->kind == QIP_OP_SWAP;
#define RG(NC, SW)                                                                                                       \
  hipLaunchKernelGGL((k_real_groups<R, V, K, NC, SW, NT>), grid, block, 0, stream, d_in, d_out, ins, d, tab)
  const int nc = (int)f.n_control;
  if (swap) {
    if constexpr (K == 2) { RG(0, true); }
    else if constexpr (K == 3) { RG(1, true); }
    else if constexpr (K == 4) { if (nc == 0) RG(0, true); else RG(2, true); }
  } else {
    if constexpr (K == 1) { RG(0, false); }
    else if constexpr (K == 2) { if (nc == 0) RG(0, false); else RG(1, false); }
    else if constexpr (K == 3) { if (nc == 0) RG(0, false); else if (nc == 1) RG(1, false); else RG(2, false); }
    else { if (nc == 0) RG(0, false); else if (nc == 1) RG(1, false); else if (nc == 2) RG(2, false); else RG(3, false); }
  }
#undef RG
  HIPCHK(hipGetLastError());
  return QIP_OK;
}

// ---- ... with ONE index bit inside the 16-byte vector (positions 0 / 1: an op on the last qubits) -----------------------------
// The vector then holds both values of that bit, so the group's partner rows along it sit in the SAME access: the lane reads
// 2^(K-1) vectors instead of 2^K scalars and takes them apart in registers.  MODE 0: a vector of two = the bit at position 0
// (8-byte P: 16 bytes; 4-byte P with positions 0 AND 1 in the op: 8 bytes).  4-byte P, 16-byte vectors of four: MODE 1 =
// component bit 0 is the index bit at position 0 and component bit 1 a free index bit (two independent groups per lane),
// MODE 2 = component bit 1 is the index bit at position 1, bit 0 free.  LB = which bit of the sub-index that position is.
// Same folds, same order: bit-equal to k_real_groups / the literal kernel.  K <= 3 (wider ops with a low bit: V = 1 above).
template <typename R, int K, int NC, bool SWAP, int LB, int MODE>
__global__ __launch_bounds__(kBlock) void k_real_groups_low(const R* __restrict__ in, R* __restrict__ out, Ins ins, RealGroupDesc d,
                                                            RealTab<R> tab) {
  constexpr int VW = MODE == 0 ? 2 : 4, W = VW / 2;
  using XV = typename RVec<R, VW>::type;
  using F = typename RVec<R, W>::type;
  constexpr int M = 1 << K, KOP = K - NC, SIDE = 1 << KOP, THR = M - SIDE, LOWBIT = 1 << LB;
  const uint64_t w = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (w >= d.nitems) return;
  const uint64_t base = insert_bits<K - 1>(w, ins);
  const XV* inv = reinterpret_cast<const XV*>(in);
  XV* outv = reinterpret_cast<XV*>(out);
  F x[M];
#pragma unroll
  for (int m = 0; m < M; ++m) {
    if (m & LOWBIT) continue;
    const XV v = inv[base | d.off[m]];
    if constexpr (MODE == 0) {
      x[m] = v.x;
      x[m | LOWBIT] = v.y;
    } else if constexpr (MODE == 1) {
      x[m] = F{v.x, v.z};
      x[m | LOWBIT] = F{v.y, v.w};
    } else {
      x[m] = F{v.x, v.y};
      x[m | LOWBIT] = F{v.z, v.w};
    }
  }
  F o[M];
#pragma unroll
  for (int m = 0; m < M; ++m) {
    F acc = (F)(R)0;
    if (m < THR) {
      acc = acc + (F)(R)1 * x[m];
    } else if constexpr (SWAP) {
      constexpr int HALF = KOP >> 1;
      const int irow = m - THR;
      const int col = ((irow & ((1 << HALF) - 1)) << HALF) + (irow >> HALF);
      acc = acc + (F)(R)1 * x[col + THR];
    } else {
#pragma unroll
      for (int c = 0; c < SIDE; ++c) {
        const R v = tab.v[(m - THR) * SIDE + c];
        if (!(v == (R)0)) acc = acc + (F)v * x[c + THR];
      }
    }
    o[m] = acc;
  }
#pragma unroll
  for (int m = 0; m < M; ++m) {
    if (m & LOWBIT) continue;
    XV v;
    if constexpr (MODE == 0) {
      v = XV{o[m], o[m | LOWBIT]};
    } else if constexpr (MODE == 1) {
      v = XV{o[m].x, o[m | LOWBIT].x, o[m].y, o[m | LOWBIT].y};
    } else {
      v = XV{o[m].x, o[m].y, o[m | LOWBIT].x, o[m | LOWBIT].y};
    }
    const uint64_t at = base | d.off[m];
    outv[at] = d.accumulate ? (XV)(outv[at] + v) : v;  // (non-temporal accesses measured here too: no gain, r06_real_p.md)
  }
}

template <typename R, int K, int NC, bool SW, int MODE>
static void launch_real_low_lb(int lb, dim3 grid, hipStream_t stream, const R* d_in, R* d_out, const Ins& ins, const RealGroupDesc& d,
                               const RealTab<R>& tab) {
#define RL(LBV) hipLaunchKernelGGL((k_real_groups_low<R, K, NC, SW, LBV, MODE>), grid, dim3(kBlock), 0, stream, d_in, d_out, ins, d, tab)
  if (lb == 0) RL(0);
  if constexpr (K >= 2) { if (lb == 1) RL(1); }
  if constexpr (K >= 3) { if (lb == 2) RL(2); }
#undef RL
}

template <typename R, int MODE>
static int launch_real_low(const FlatOp& f, int lb, const R* d_in, R* d_out, const Ins& ins, const RealGroupDesc& d, const RealTab<R>& tab,
                           hipStream_t stream) {
  const dim3 grid((unsigned)((d.nitems + kBlock - 1) / kBlock));
  const bool swap = f.inner->kind == QIP_OP_SWAP;
  const int nc = (int)f.n_control;
#define RLK(KK, NC, SW) launch_real_low_lb<R, KK, NC, SW, MODE>(lb, grid, stream, d_in, d_out, ins, d, tab)
  switch (f.k_all) {
    case 1: RLK(1, 0, false); break;
    case 2:
      if (swap) RLK(2, 0, true);
      else if (nc == 0) RLK(2, 0, false);
      else RLK(2, 1, false);
      break;
    default:
      if (swap) RLK(3, 1, true);
      else if (nc == 0) RLK(3, 0, false);
      else if (nc == 1) RLK(3, 1, false);
      else RLK(3, 2, false);
      break;
  }
#undef RLK
  HIPCHK(hipGetLastError());
  return QIP_OK;
}

// true (and launched) when the op qualifies; false: the literal kernel takes it
template <typename R>
static int launch_real_groups(uint32_t n, const FlatOp& f, const R* d_in, R* d_out, int accumulate, hipStream_t stream, bool* done) {
  *done = false;
  const uint32_t K = f.k_all;
  if (!f.distinct || K > 4 || K >= n || g_force_generic) return QIP_OK;
  if (f.inner->kind == QIP_OP_SPARSE) return QIP_OK;
  if (f.inner->kind == QIP_OP_SWAP && (f.n_op & 1u)) return QIP_OK;
  constexpr int VMAX = 16 / (int)sizeof(R);
  constexpr uint32_t LOGV = sizeof(R) == 8 ? 1u : 2u;
  std::vector<uint32_t> pos(K);
  uint32_t lowest = 64;
  for (uint32_t j = 0; j < K; ++j) {
    pos[j] = (uint32_t)(n - 1 - f.outer->indices[j]);
    lowest = std::min(lowest, pos[j]);
  }
  const bool aligned = ((uintptr_t)d_in % 16 == 0) && ((uintptr_t)d_out % 16 == 0);
  const bool vec = lowest >= LOGV && n >= K + LOGV && aligned;
  if (!vec && aligned && K <= 3 && n >= K + 2) {  // one index bit inside the vector: k_real_groups_low
    bool p0 = false, p1 = false;
    for (uint32_t j = 0; j < K; ++j) {
      p0 = p0 || pos[j] == 0;
      p1 = p1 || pos[j] == 1;
    }
    const int mode = sizeof(R) == 8 ? (p0 ? 0 : -1) : (p0 && p1) ? 0 : p0 ? 1 : p1 ? 2 : -1;
    if (mode >= 0) {
      const uint32_t inpos = mode == 2 ? 1u : 0u, logvw = mode == 0 ? 1u : 2u;
      uint32_t jl = 0;
      for (uint32_t j = 0; j < K; ++j)
        if (pos[j] == inpos) jl = j;
      RealGroupDesc dl;
      memset(&dl, 0, sizeof dl);
      dl.nitems = 1ull << (n - (K - 1) - logvw);
      dl.accumulate = accumulate;
      std::vector<uint32_t> high;
      for (uint32_t j = 0; j < K; ++j)
        if (j != jl) high.push_back(pos[j] - logvw);
      for (uint32_t m = 0; m < (1u << K); ++m)
        for (uint32_t j = 0; j < K; ++j)
          if (j != jl) dl.off[m] |= (uint64_t)((m >> (K - 1 - j)) & 1u) << (pos[j] - logvw);
      const Ins insl = make_ins(high, 0);
      RealTab<R> tabl;
      memset(&tabl, 0, sizeof tabl);
      if (f.inner->kind == QIP_OP_MATRIX) memcpy(tabl.v, f.inner->dense, sizeof(R) << (2 * f.n_op));
      *done = true;
      const int lb = (int)(K - 1 - jl);
      if (mode == 0) return launch_real_low<R, 0>(f, lb, d_in, d_out, insl, dl, tabl, stream);
      if constexpr (sizeof(R) == 4) {
        if (mode == 1) return launch_real_low<R, 1>(f, lb, d_in, d_out, insl, dl, tabl, stream);
        return launch_real_low<R, 2>(f, lb, d_in, d_out, insl, dl, tabl, stream);
      }
    }
  }
  // a 4-byte P with its lowest index bit at position 1: pairs of rows (8-byte accesses) instead of single ones
  const bool half = !vec && sizeof(R) == 4 && lowest == 1 && n >= K + 1 && aligned;
  const uint32_t lv = vec ? LOGV : half ? 1u : 0u;
  RealGroupDesc d;
  memset(&d, 0, sizeof d);
  d.nitems = 1ull << (n - K - lv);
  d.accumulate = accumulate;
  for (uint32_t m = 0; m < (1u << K); ++m)
    for (uint32_t j = 0; j < K; ++j) d.off[m] |= (uint64_t)((m >> (K - 1 - j)) & 1u) << (pos[j] - lv);
  std::vector<uint32_t> sorted(K);
  for (uint32_t j = 0; j < K; ++j) sorted[j] = pos[j] - lv;
  const Ins ins = make_ins(sorted, 0);
  RealTab<R> tab;
  memset(&tab, 0, sizeof tab);
  if (f.inner->kind == QIP_OP_MATRIX) memcpy(tab.v, f.inner->dense, sizeof(R) << (2 * f.n_op));
  *done = true;
  // a vector far beyond the caches streams (non-temporal accesses, as the state kernels' sweeps do); V = 1 is the rare shape
  // (only when a wave's accesses cover whole 128-byte lines, i.e. no index bit within the low three vector positions: measured at
  // n = 28, f64, a dense op on qubits 3 and n-2 — 16-byte runs — 785 us with cached accesses, 1448 us with non-temporal ones)
  const bool nt = vec && (sizeof(R) << n) >= (64ull << 20) && lowest - lv >= 3;
#define RK(KK)                                                                                                   \
  if constexpr (sizeof(R) == 4)                                                                                  \
    if (half) return launch_real_groups_k<R, 2, KK, false>(f, d_in, d_out, ins, d, tab, stream);                 \
  return !vec ? launch_real_groups_k<R, 1, KK, false>(f, d_in, d_out, ins, d, tab, stream)                       \
         : nt ? launch_real_groups_k<R, VMAX, KK, true>(f, d_in, d_out, ins, d, tab, stream)                     \
              : launch_real_groups_k<R, VMAX, KK, false>(f, d_in, d_out, ins, d, tab, stream)
  switch (K) {
    case 1: RK(1);
    case 2: RK(2);
    case 3: RK(3);
    default: RK(4);
  }
#undef RK
}

template <typename R>
static int apply_op_real_device(uint32_t n, const qip_op* op, const R* d_in, uint64_t in_len, R* d_out, uint64_t out_len,
                                uint64_t in_off, uint64_t out_off, int accumulate, hipStream_t stream) {
  FlatOp f;
  QCHK(flatten_op(n, op, false, &f));
  if (out_len == 0) return QIP_OK;
  GatherDesc d;
  memset(&d, 0, sizeof d);
  d.n = n;
  d.k_all = f.k_all;
  d.n_control = f.n_control;
  d.n_op = f.n_op;
  d.inner_kind = f.inner->kind;
  d.accumulate = accumulate;
  d.in_len = in_len;
  d.out_len = out_len;
  d.in_off = in_off;
  d.out_off = out_off;
  for (uint32_t j = 0; j < f.k_all; ++j) d.pos[j] = (uint32_t)(n - 1 - f.outer->indices[j]);
  if (in_off == 0 && out_off == 0 && in_len == (1ull << n) && out_len == in_len) {  // the whole vector: each input read once
    bool done = false;
    QCHK(launch_real_groups<R>(n, f, d_in, d_out, accumulate, stream, &done));
    if (done) return QIP_OK;
  }
  // the literal kernel, V rows per lane when no index bit sits inside a 16-byte vector and both windows are made of whole vectors
  constexpr int VMAX = 16 / (int)sizeof(R);
  bool wide = ((uintptr_t)d_in % 16 == 0) && ((uintptr_t)d_out % 16 == 0) && in_off % VMAX == 0 && out_off % VMAX == 0 &&
              in_len % VMAX == 0 && out_len % VMAX == 0;
  for (uint32_t j = 0; j < f.k_all; ++j) wide = wide && d.pos[j] >= (sizeof(R) == 8 ? 1u : 2u);
  const dim3 grid(grid_stride(wide ? out_len / VMAX : out_len)), block(kBlock);
  if (f.inner->kind == QIP_OP_SWAP || (f.inner->kind == QIP_OP_MATRIX && f.n_op <= 4)) {
    RealTab<R> tab;
    memset(&tab, 0, sizeof tab);
    if (f.inner->kind == QIP_OP_MATRIX) memcpy(tab.v, f.inner->dense, sizeof(R) << (2 * f.n_op));
    if (wide)
      hipLaunchKernelGGL((k_gather_real<R, true, VMAX>), grid, block, 0, stream, d_in, d_out, d, tab, (const R*)nullptr,
                         (const uint64_t*)nullptr, (const uint64_t*)nullptr, (const R*)nullptr);
    else
      hipLaunchKernelGGL((k_gather_real<R, true, 1>), grid, block, 0, stream, d_in, d_out, d, tab, (const R*)nullptr,
                         (const uint64_t*)nullptr, (const uint64_t*)nullptr, (const R*)nullptr);
    HIPCHK(hipGetLastError());
    return QIP_OK;
  }
  // a payload too large for the kernel arguments: one device buffer for this call
  size_t b_dense = 0, b_rp = 0, b_cols = 0, b_vals = 0, o_cols = 0, o_vals = 0;
  if (f.inner->kind == QIP_OP_MATRIX) {
    b_dense = sizeof(R) << (2 * f.n_op);
  } else {
    const uint64_t rows = 1ull << f.n_op;
    const uint64_t nnz = f.inner->sparse_rowptr[rows];
    b_rp = (rows + 1) * 8;
    b_cols = nnz * 8;
    b_vals = nnz * sizeof(R);
    o_cols = (b_rp + 15) & ~(size_t)15;
    o_vals = (o_cols + b_cols + 15) & ~(size_t)15;
  }
  char* buf = nullptr;
  HIPCHK(hipMalloc((void**)&buf, std::max<size_t>(b_dense + o_vals + b_vals, 16)));
  auto body = [&]() -> int {
    if (b_dense) HIPCHK(hipMemcpyAsync(buf, f.inner->dense, b_dense, hipMemcpyHostToDevice, stream));
    if (b_rp) HIPCHK(hipMemcpyAsync(buf, f.inner->sparse_rowptr, b_rp, hipMemcpyHostToDevice, stream));
    if (b_cols) HIPCHK(hipMemcpyAsync(buf + o_cols, f.inner->sparse_cols, b_cols, hipMemcpyHostToDevice, stream));
    if (b_vals) HIPCHK(hipMemcpyAsync(buf + o_vals, f.inner->sparse_vals, b_vals, hipMemcpyHostToDevice, stream));
    RealTab<R> tab;
    memset(&tab, 0, sizeof tab);
    if (wide)
      hipLaunchKernelGGL((k_gather_real<R, false, VMAX>), grid, block, 0, stream, d_in, d_out, d, tab, (const R*)buf, (const uint64_t*)buf,
                         (const uint64_t*)(buf + o_cols), (const R*)(buf + o_vals));
    else
      hipLaunchKernelGGL((k_gather_real<R, false, 1>), grid, block, 0, stream, d_in, d_out, d, tab, (const R*)buf, (const uint64_t*)buf,
                         (const uint64_t*)(buf + o_cols), (const R*)(buf + o_vals));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(stream));
    return QIP_OK;
  };
  const int rc = body();
  if (rc != QIP_OK) (void)hipStreamSynchronize(stream);
  (void)hipFree(buf);
  return rc;
}

// complex P on device slices: the literal kernel of the state path (k_gather_generic) through a handle that adopts the caller's
// stream and owns only the payload arena
template <typename T>
static int apply_op_complex_device(int dtype, int device, hipStream_t stream, uint32_t n, const qip_op* op, const void* d_in,
                                   uint64_t in_len, void* d_out, uint64_t out_len, uint64_t in_off, uint64_t out_off,
                                   int accumulate) {
  FlatOp f;
  QCHK(flatten_op(n, op, false, &f));
  if (out_len == 0) return QIP_OK;
  qip_hip_state* s = nullptr;
  QCHK(qip_hip_state_wrap(n, dtype, device, d_out, nullptr, (void*)stream, &s));
  int rc = launch_gather<T>(s, f, (const amp_t<T>*)d_in, in_len, (amp_t<T>*)d_out, out_len, in_off, out_off, accumulate);
  std::string keep = g_last_error;
  qip_hip_state_destroy(s);  // (synchronises the stream: the payload arena dies with the handle)
  if (rc != QIP_OK) g_last_error = keep;
  return rc;
}

static size_t elem_bytes(int dtype) {
  switch (dtype) {
    case QIP_C64: return 16;
    case QIP_C32: case QIP_F64: case QIP_I64: return 8;
    case QIP_F32: case QIP_I32: return 4;
    default: return 0;
  }
}

extern "C" int qip_hip_apply_op_device(int dtype, int device, void* stream, uint32_t n, const qip_op* op, const void* d_in,
                                       uint64_t in_len, void* d_out, uint64_t out_len, uint64_t in_off, uint64_t out_off,
                                       int accumulate) try {
  if (!elem_bytes(dtype)) return fail(QIP_ERR_INVALID, "bad dtype %d", dtype);
  if ((in_len && !d_in) || (out_len && !d_out)) return fail(QIP_ERR_INVALID, "null buffer");
  if (n == 0 || n > 40) return fail(QIP_ERR_INVALID, "n = %u out of range [1, 40]", n);
  if (qip_hip_device_count() <= device || device < 0)
    return fail(QIP_ERR_NO_DEVICE, "no HIP device %d visible: qip_hip has no CPU fallback", device);
  HIPCHK(hipSetDevice(device));
  hipStream_t st = (hipStream_t)stream;
  switch (dtype) {
    case QIP_C64: return apply_op_complex_device<double>(dtype, device, st, n, op, d_in, in_len, d_out, out_len, in_off, out_off, accumulate);
    case QIP_C32: return apply_op_complex_device<float>(dtype, device, st, n, op, d_in, in_len, d_out, out_len, in_off, out_off, accumulate);
    case QIP_F64: return apply_op_real_device<double>(n, op, (const double*)d_in, in_len, (double*)d_out, out_len, in_off, out_off, accumulate, st);
    case QIP_F32: return apply_op_real_device<float>(n, op, (const float*)d_in, in_len, (float*)d_out, out_len, in_off, out_off, accumulate, st);
    case QIP_I64: return apply_op_real_device<uint64_t>(n, op, (const uint64_t*)d_in, in_len, (uint64_t*)d_out, out_len, in_off, out_off, accumulate, st);
    default: return apply_op_real_device<uint32_t>(n, op, (const uint32_t*)d_in, in_len, (uint32_t*)d_out, out_len, in_off, out_off, accumulate, st);
  }
} QIP_CATCH_ALL

// host slices of a real / integer P: upload, the device call above, download
static int apply_op_real_host(int dtype, uint32_t n, const qip_op* op, const void* in, uint64_t in_len, void* out,
                              uint64_t out_len, uint64_t in_off, uint64_t out_off, int accumulate) {
  FlatOp f;
  QCHK(flatten_op(n, op, false, &f));  // (argument errors before any device work, as the complex twin reports them)
  if (out_len == 0) return QIP_OK;
  if (qip_hip_device_count() <= 0) return fail(QIP_ERR_NO_DEVICE, "no HIP device visible: qip_hip has no CPU fallback");
  HIPCHK(hipSetDevice(0));
  const size_t eb = elem_bytes(dtype);
  void *d_in = nullptr, *d_out = nullptr;
  HIPCHK(hipMalloc(&d_in, std::max<size_t>(in_len * eb, 16)));
  hipError_t e = hipMalloc(&d_out, out_len * eb);
  if (e != hipSuccess) {
    (void)hipFree(d_in);
    return fail(QIP_ERR_DEVICE, "hipMalloc failed: %s", hipGetErrorString(e));
  }
  auto body = [&]() -> int {
    if (in_len) HIPCHK(hipMemcpy(d_in, in, in_len * eb, hipMemcpyHostToDevice));
    if (accumulate) HIPCHK(hipMemcpy(d_out, out, out_len * eb, hipMemcpyHostToDevice));
    QCHK(qip_hip_apply_op_device(dtype, 0, nullptr, n, op, d_in, in_len, d_out, out_len, in_off, out_off, accumulate));
    HIPCHK(hipMemcpy(out, d_out, out_len * eb, hipMemcpyDeviceToHost));  // (null stream: ordered behind the kernel)
    return QIP_OK;
  };
  const int rc = body();
  (void)hipDeviceSynchronize();
  (void)hipFree(d_in);
  (void)hipFree(d_out);
  return rc;
}

extern "C" int qip_hip_apply_op_host(int dtype, uint32_t n, const qip_op* op, const void* in,
                                     uint64_t in_len, void* out, uint64_t out_len, uint64_t in_off,
                                     uint64_t out_off, int accumulate) try {
  if (!elem_bytes(dtype)) return fail(QIP_ERR_INVALID, "bad dtype %d", dtype);
  if ((in_len && !in) || (out_len && !out)) return fail(QIP_ERR_INVALID, "null buffer");
  if (n == 0 || n > 40) return fail(QIP_ERR_INVALID, "n = %u out of range [1, 40]", n);
  if (dtype != QIP_C64 && dtype != QIP_C32) return apply_op_real_host(dtype, n, op, in, in_len, out, out_len, in_off, out_off, accumulate);
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

