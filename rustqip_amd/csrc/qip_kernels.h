// qip_kernels.h — gfx950 (CDNA4, wave64) kernels for the gate-application hot path.
//
// Data layout: the 2^n amplitudes are one interleaved {re,im} array in HBM
// (16 B per Complex<f64>, 8 B per Complex<f32>); qubit q is index bit n-1-q
// (qip-iterators/src/matrix_ops.rs:18,28).  All fast kernels update the vector
// IN PLACE: every amplitude that can change is read once and written once, which is
// the algorithmic-byte count of SURVEY.md §8(d).  The reference instead gathers into
// a second 2^n buffer per gate (matrix_ops.rs:127-152); only the literal fallback
// kernel (k_gather_generic) keeps that out-of-place shape.
//
// All kernels are HBM-bandwidth bound (a dense 1-qubit gate is 14 flop per 32 B), so
// the design rules are: 16-B per-lane accesses, contiguous 1-KiB wave rows whenever the
// touched bit positions allow it, several independent loads in flight per lane, and no
// fused multiply-add (the file is compiled with -ffp-contract=off so products and sums
// round exactly as the reference's unfused num-complex arithmetic does).
#pragma once

#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#include <stdint.h>
#else
// run-time compilation (hiprtc: segment-specialised tile sweeps, see tile_passes_body): no system headers there
typedef __hip_internal::uint32_t uint32_t;
typedef __hip_internal::uint64_t uint64_t;
typedef __hip_internal::int32_t int32_t;
typedef __hip_internal::int64_t int64_t;
#endif

namespace qipk {

constexpr int kBlock = 256;   // 4 waves of 64
constexpr int kMaxIns = 64;   // max bit positions removed from the work index

template <typename T> struct V2;
// native clang vectors (not HIP's double2 wrapper struct): they stay in registers when held in
// small arrays and load/store as one global_load_dwordx4 / dwordx2
template <> struct V2<double> { using type = double __attribute__((ext_vector_type(2))); };
template <> struct V2<float> { using type = float __attribute__((ext_vector_type(2))); };
template <typename T> using amp_t = typename V2<T>::type;
template <typename A, typename B> struct SameT { static constexpr bool v = false; };  // (std::is_same: no system headers under hiprtc)
template <typename A> struct SameT<A, A> { static constexpr bool v = true; };

// ---- index arithmetic ------------------------------------------------------

// Sorted-ascending bit positions to open up in a dense work index, plus bits to set.
// A work index w in [0, 2^(n-npos)) becomes the amplitude index whose bits at `pos`
// are zero (then OR-ed with `ormask`): the kernel-side replacement of the reference's
// per-row full_to_sub / sub_to_full bit loops (matrix_ops.rs:12-30).
struct Ins {
  uint64_t ormask;
  uint32_t npos;
  uint32_t pos[kMaxIns];
};

// NP >= 0: number of positions known at compile time (fully unrolled, positions live in
// SGPRs straight from the kernel-argument segment); NP < 0: run-time count.
template <int NP>
__device__ __forceinline__ uint64_t insert_bits(uint64_t w, const Ins& ins) {
  if constexpr (NP >= 0) {
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const uint32_t p = ins.pos[j];
      const uint64_t low = w & ((1ull << p) - 1ull);
      w = ((w >> p) << (p + 1)) | low;
    }
  } else {
    for (uint32_t j = 0; j < ins.npos; ++j) {
      const uint32_t p = ins.pos[j];
      const uint64_t low = w & ((1ull << p) - 1ull);
      w = ((w >> p) << (p + 1)) | low;
    }
  }
  return w | ins.ormask;
}

// ---- complex arithmetic, num-complex formulas, never contracted -------------

template <typename A> __device__ __forceinline__ A cmul(A a, A b) {
  A r;
  r.x = a.x * b.x - a.y * b.y;
  r.y = a.x * b.y + a.y * b.x;
  return r;
}
// e <- f * e IN PLACE: the four products and two sums of cmul(f, e), rounded one by one in the same way (v_add_f64 with a negated
// operand IS the subtraction), written so that no result needs a copy: inside the diagonal-run loop of k_tile_passes the compiler's
// own allocation of cmul spends two v_mov_b64 per element on moving the results back into the loop-carried registers (8 instead of
// 6 vector instructions per product; the loop is bound by vector issue).
__device__ __forceinline__ void cscale_inplace(amp_t<double> f, amp_t<double>& e) {
  double t0, t1;
  asm("v_mul_f64 %2, %4, %0\n\t"
      "v_mul_f64 %3, %5, %1\n\t"
      "v_mul_f64 %1, %4, %1\n\t"
      "v_mul_f64 %0, %5, %0\n\t"
      "v_add_f64 %1, %1, %0\n\t"
      "v_add_f64 %0, %2, -%3"
      : "+v"(e.x), "+v"(e.y), "=&v"(t0), "=&v"(t1)
      : "v"(f.x), "v"(f.y));
}
__device__ __forceinline__ void cscale_inplace(amp_t<float> f, amp_t<float>& e) { e = cmul(f, e); }
template <typename A> __device__ __forceinline__ A cadd(A a, A b) {
  A r;
  r.x = a.x + b.x;
  r.y = a.y + b.y;
  return r;
}
template <typename A> __device__ __forceinline__ A czero() {
  A r = {};
  return r;
}

// Packed single precision: one 16-B element = TWO adjacent Complex<f32> amplitudes (index bit 0 lives
// inside the element).  Any op that does not involve bit 0 acts on both halves alike, so the f32 state
// is swept with the same 16-B per-lane accesses as f64: an (n-1)-"qubit" vector of float4 elements.
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 cmul(amp_t<float> m, f32x4 x) {
  f32x4 r;
  r.x = m.x * x.x - m.y * x.y;
  r.y = m.x * x.y + m.y * x.x;
  r.z = m.x * x.z - m.y * x.w;
  r.w = m.x * x.w + m.y * x.z;
  return r;
}
__device__ __forceinline__ f32x4 cadd(f32x4 a, f32x4 b) { return a + b; }

template <typename E> __device__ __forceinline__ E shfl_xor_e(E v, int lane_mask);
template <> __device__ __forceinline__ amp_t<double> shfl_xor_e(amp_t<double> v, int m) {
  amp_t<double> r;
  r.x = __shfl_xor(v.x, m, 64);
  r.y = __shfl_xor(v.y, m, 64);
  return r;
}
template <> __device__ __forceinline__ amp_t<float> shfl_xor_e(amp_t<float> v, int m) {
  amp_t<float> r;
  r.x = __shfl_xor(v.x, m, 64);
  r.y = __shfl_xor(v.y, m, 64);
  return r;
}
template <> __device__ __forceinline__ f32x4 shfl_xor_e(f32x4 v, int m) {
  f32x4 r;
  r.x = __shfl_xor(v.x, m, 64);
  r.y = __shfl_xor(v.y, m, 64);
  r.z = __shfl_xor(v.z, m, 64);
  r.w = __shfl_xor(v.w, m, 64);
  return r;
}
template <typename E> __device__ __forceinline__ E shfl_e(E v, int src);
template <> __device__ __forceinline__ amp_t<double> shfl_e(amp_t<double> v, int src) {
  amp_t<double> r;
  r.x = __shfl(v.x, src, 64);
  r.y = __shfl(v.y, src, 64);
  return r;
}
template <> __device__ __forceinline__ amp_t<float> shfl_e(amp_t<float> v, int src) {
  amp_t<float> r;
  r.x = __shfl(v.x, src, 64);
  r.y = __shfl(v.y, src, 64);
  return r;
}
template <> __device__ __forceinline__ f32x4 shfl_e(f32x4 v, int src) {
  f32x4 r;
  r.x = __shfl(v.x, src, 64);
  r.y = __shfl(v.y, src, 64);
  r.z = __shfl(v.z, src, 64);
  r.w = __shfl(v.w, src, 64);
  return r;
}

// 2x2 gate, row-major, plus a 4-bit mask of entries that are not exactly zero:
// MatrixOpIterator skips zero entries (qubit_iterators.rs:49), so they add no term.
template <typename T> struct Mat2 {
  amp_t<T> m[4];
  uint32_t nz;
};

// The streaming kernels below come in two shapes selected by the launcher:
//   <U, GUARD = false, NT>  the power-of-two work-item count is a multiple of 2^(kStrideShift+log2 U):
//                           no bounds checks; a block's 256 lanes cover 256 consecutive work items
//                           (4 contiguous 1-KiB wave rows) and a lane's U items sit 2^kStrideShift
//                           items = 32 KiB apart.  Measured on MI355X at n = 30 (tools/tune_gate1q.hip,
//                           profiles/r01_tuning.md): adjacent 4-KiB runs per lane reach ~5.3 TB/s, the
//                           32-KiB spacing ~6.2 TB/s; 16 KiB or >= 64 KiB spacing is slower again.
//                           NT = non-temporal loads/stores for states that cannot live in the 256-MiB
//                           Infinity Cache anyway (+8 % at n = 30; every amplitude is touched once per gate).
//   <U = 1, GUARD = true>   tiny states.
constexpr int kStrideShift = 11;  // 2^11 work items x 16 B = 32 KiB between a lane's accesses

template <int LOGU> __device__ __forceinline__ uint64_t work_index(uint32_t u) {
  constexpr int S = kStrideShift - 8;  // kBlock = 2^8
  // 2-D grids only exist because HIP caps a launch at 2^32 threads (n = 33 with U = 2 would need exactly that)
  const uint64_t blk = blockIdx.x + (uint64_t)blockIdx.y * gridDim.x;
  return ((blk >> S) << (kStrideShift + LOGU)) | ((uint64_t)u << kStrideShift) |
         ((blk & ((1u << S) - 1u)) << 8) | threadIdx.x;
}
template <int U> struct Log2;
template <> struct Log2<1> { static constexpr int v = 0; };
template <> struct Log2<2> { static constexpr int v = 1; };
template <> struct Log2<4> { static constexpr int v = 2; };
template <> struct Log2<8> { static constexpr int v = 3; };

template <bool NT, typename A> __device__ __forceinline__ A ldg(const A* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}
template <bool NT, typename A> __device__ __forceinline__ void stg(A* p, A v) {
  if constexpr (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

// Selector bits below bit kLineBits (inside one 128-B line = 8 amplitudes) are NOT removed from the
// grid: a control or phase bit there would leave every line half-touched, and a partially written
// line costs a read-for-ownership on top of the write (measured on MI355X at n = 30: T on bit 0 took
// 8.8 ms as a strided half-line sweep vs 5.3 ms for a full sweep; CNOT with control bit 0: 9.4 ms vs
// 2.7 ms with a high control).  Such bits become a per-lane predicate instead — `(idx & mask) == val`
// picks between the updated and the unchanged amplitude — and the sweep reads and writes whole lines.
constexpr uint32_t kLineBits = 3;
struct Sel {
  uint64_t mask, val;
};
__device__ __forceinline__ bool sel_hit(uint64_t idx, const Sel& s) { return (idx & s.mask) == s.val; }

// ---- 1-qubit gate, pair per lane ---------------------------------------------
// Work item = one (|0>,|1>) pair of the target bit inside the all-controls-one subspace.
// `ins` opens the target bit and every control bit >= kLineBits; ormask sets those controls;
// `low` carries the controls below kLineBits.
// out0 = m00*a0 + m01*a1 ; out1 = m10*a0 + m11*a1, folded from 0 in column order
// (matrix_ops.rs:78-93, ops.rs:104-110).
template <typename T, int U, bool GUARD, bool NT, int NP, typename E = amp_t<T>>
__global__ __launch_bounds__(kBlock) void k_gate1q_pair(E* __restrict__ st, uint64_t npairs,
                                                        Ins ins, uint64_t tmask, Sel low, Mat2<T> g) {
  using A = E;
  if (GUARD && work_index<0>(0) >= npairs) return;
  uint64_t i0[U];
  A a0[U], a1[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    i0[u] = insert_bits<NP>(work_index<Log2<U>::v>(u), ins);
    a0[u] = ldg<NT>(st + i0[u]);
    a1[u] = ldg<NT>(st + (i0[u] | tmask));
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    A r0 = czero<A>(), r1 = czero<A>();
    if (g.nz & 1u) r0 = cadd(r0, cmul(g.m[0], a0[u]));
    if (g.nz & 2u) r0 = cadd(r0, cmul(g.m[1], a1[u]));
    if (g.nz & 4u) r1 = cadd(r1, cmul(g.m[2], a0[u]));
    if (g.nz & 8u) r1 = cadd(r1, cmul(g.m[3], a1[u]));
    const bool hit = sel_hit(i0[u], low);
    stg<NT>(st + i0[u], hit ? r0 : a0[u]);
    stg<NT>(st + (i0[u] | tmask), hit ? r1 : a1[u]);
  }
}

// ---- 1-qubit gate, amplitude per lane, partner by cross-lane exchange -----------
// For a target bit that lands inside the lane index (work-index bit tb < 6) the wave
// keeps fully contiguous 1-KiB rows: each lane loads ONE amplitude, fetches its partner
// from lane ^ (1<<tb) and computes only its own output row.  `ins` opens only the
// control bits >= kLineBits.  Requires the work-item count to be a multiple of 64.
template <typename T, int U, bool GUARD, bool NT, int NP, typename E = amp_t<T>>
__global__ __launch_bounds__(kBlock) void k_gate1q_xlane(E* __restrict__ st, uint64_t namps,
                                                         Ins ins, uint32_t tb, Sel low, Mat2<T> g) {
  using A = E;
  using M = amp_t<T>;
  if (GUARD && work_index<0>(0) >= namps) return;  // whole waves leave together (namps % 64 == 0)
  const bool hi = (threadIdx.x >> tb) & 1u;  // this lane holds the |1> member
  // row of the gate this lane evaluates: (m_lo, m_hi) multiply (|0> member, |1> member)
  const M m_lo = hi ? g.m[2] : g.m[0];
  const M m_hi = hi ? g.m[3] : g.m[1];
  const bool nz_lo = hi ? (g.nz & 4u) : (g.nz & 1u);
  const bool nz_hi = hi ? (g.nz & 8u) : (g.nz & 2u);
  uint64_t idx[U];
  A own[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    idx[u] = insert_bits<NP>(work_index<Log2<U>::v>(u), ins);
    own[u] = ldg<NT>(st + idx[u]);
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const A other = shfl_xor_e<A>(own[u], 1 << tb);
    const A lo = hi ? other : own[u];
    const A hv = hi ? own[u] : other;
    A r = czero<A>();
    if (nz_lo) r = cadd(r, cmul(m_lo, lo));
    if (nz_hi) r = cadd(r, cmul(m_hi, hv));
    stg<NT>(st + idx[u], sel_hit(idx[u], low) ? r : own[u]);
  }
}

// ---- packed f32, target = index bit 0: the pair is the two halves of one element ------------------
// `ins` / `low` are in element-index space (controls are at bit >= 1 of the amplitude index).
template <int U, bool GUARD, bool NT, int NP>
__global__ __launch_bounds__(kBlock) void k_gate1q_inelem(f32x4* __restrict__ st, uint64_t count, Ins ins,
                                                          Sel low, Mat2<float> g) {
  using M = amp_t<float>;
  if (GUARD && work_index<0>(0) >= count) return;
  uint64_t idx[U];
  f32x4 x[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    idx[u] = insert_bits<NP>(work_index<Log2<U>::v>(u), ins);
    x[u] = ldg<NT>(st + idx[u]);
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    M a0, a1;
    a0.x = x[u].x;
    a0.y = x[u].y;
    a1.x = x[u].z;
    a1.y = x[u].w;
    M r0 = czero<M>(), r1 = czero<M>();
    if (g.nz & 1u) r0 = cadd(r0, cmul(g.m[0], a0));
    if (g.nz & 2u) r0 = cadd(r0, cmul(g.m[1], a1));
    if (g.nz & 4u) r1 = cadd(r1, cmul(g.m[2], a0));
    if (g.nz & 8u) r1 = cadd(r1, cmul(g.m[3], a1));
    f32x4 r;
    r.x = r0.x;
    r.y = r0.y;
    r.z = r1.x;
    r.w = r1.y;
    stg<NT>(st + idx[u], sel_hit(idx[u], low) ? r : x[u]);
  }
}

// ---- scalar phase on a subspace ----------------------------------------------------
// Every amplitude whose involved bits match is multiplied by `value`: Z/S/T, controlled-phase,
// multi-controlled Z...  A diagonal gate whose other diagonal entries are exactly 1 leaves those
// amplitudes untouched (1*x == x), so whole lines outside the subspace are neither read nor
// written.  `ins` opens the involved bits >= kLineBits; `low` carries the ones inside a line.
template <typename T, int U, bool GUARD, bool NT, int NP, typename E = amp_t<T>>
__global__ __launch_bounds__(kBlock) void k_phase(E* __restrict__ st, uint64_t count, Ins ins,
                                                  Sel low, amp_t<T> value) {
  using A = E;
  if (GUARD && work_index<0>(0) >= count) return;
  uint64_t idx[U];
  A x[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    idx[u] = insert_bits<NP>(work_index<Log2<U>::v>(u), ins);
    x[u] = ldg<NT>(st + idx[u]);
  }
#pragma unroll
  for (int u = 0; u < U; ++u) stg<NT>(st + idx[u], sel_hit(idx[u], low) ? cmul(value, x[u]) : x[u]);
}

// ---- diagonal 1-qubit gate with both entries != 1 (Rz), optionally controlled ----------
// Amplitude per lane inside the control subspace; the factor is picked by the target bit.
template <typename T, int U, bool GUARD, bool NT, int NP, typename E = amp_t<T>>
__global__ __launch_bounds__(kBlock) void k_diag1q(E* __restrict__ st, uint64_t count, Ins ins,
                                                   uint64_t tmask, Sel low, amp_t<T> d0, amp_t<T> d1) {
  using A = E;
  if (GUARD && work_index<0>(0) >= count) return;
  uint64_t idx[U];
  A x[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    idx[u] = insert_bits<NP>(work_index<Log2<U>::v>(u), ins);
    x[u] = ldg<NT>(st + idx[u]);
  }
#pragma unroll
  for (int u = 0; u < U; ++u)
    stg<NT>(st + idx[u], sel_hit(idx[u], low) ? cmul((idx[u] & tmask) ? d1 : d0, x[u]) : x[u]);
}

// ---- exchange of two index bits (Swap with h = 1, optionally controlled) ---------------
// SwapOpIterator (qubit_iterators.rs:208-218) yields the single column whose A and B
// halves are exchanged, with value 1: a pure move.  Only amplitudes whose two bits differ
// change, so a work item is one (01,10) pair and the other half of the vector is untouched.
// Used when both bits are >= 6 (whole 1-KiB wave rows move).
template <typename T, int U, bool GUARD, bool NT, int NP, typename E = amp_t<T>>
__global__ __launch_bounds__(kBlock) void k_swap_bits(E* __restrict__ st, uint64_t npairs,
                                                      Ins ins, uint64_t amask, uint64_t bmask, Sel low) {
  using A = E;
  if (GUARD && work_index<0>(0) >= npairs) return;
  uint64_t i0[U];
  A xa[U], xb[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    i0[u] = insert_bits<NP>(work_index<Log2<U>::v>(u), ins);
    xa[u] = ldg<NT>(st + (i0[u] | amask));
    xb[u] = ldg<NT>(st + (i0[u] | bmask));
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const bool hit = sel_hit(i0[u], low);
    stg<NT>(st + (i0[u] | amask), hit ? xb[u] : xa[u]);
    stg<NT>(st + (i0[u] | bmask), hit ? xa[u] : xb[u]);
  }
}

// ---- exchange of two index bits when one (or both) lies inside the lane index -----------
// Whole rows stay contiguous: a lane loads the row element(s) it owns and receives the value that
// moves into its slot from another lane (ds_bpermute), then full lines are written back.
//   two-row form (lb < 6 <= hb): `ins` opens bit hb (and high controls); a lane holds r0 (hb = 0)
//     and r1 (hb = 1) at the same low index; slot (lb=1, hb=0) <-> slot (lb=0, hb=1).
//   one-row form (both < 6): work-index bits la, lb (after control removal) are both lane bits:
//     the lane reads from the lane whose two bits are exchanged.
template <typename T, int U, bool GUARD, bool NT, int NP, typename E = amp_t<T>>
__global__ __launch_bounds__(kBlock) void k_swap_xlane2(E* __restrict__ st, uint64_t nitems,
                                                        Ins ins, uint32_t lb_w, uint64_t hmask, Sel low) {
  using A = E;
  if (GUARD && work_index<0>(0) >= nitems) return;
  const bool lbit = (threadIdx.x >> lb_w) & 1u;
  uint64_t i0[U];
  A r0[U], r1[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    i0[u] = insert_bits<NP>(work_index<Log2<U>::v>(u), ins);
    r0[u] = ldg<NT>(st + i0[u]);
    r1[u] = ldg<NT>(st + (i0[u] | hmask));
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    // lanes with lb = 1 give away their r0 (slot lb=1,hb=0); lanes with lb = 0 give away their r1
    const A give = lbit ? r0[u] : r1[u];
    const A got = shfl_xor_e<A>(give, 1 << lb_w);
    // the partner lane has the same control bits (they are not lb), so one predicate serves both
    const bool hit = sel_hit(i0[u], low);
    const A n0 = (hit && lbit) ? got : r0[u];
    const A n1 = (hit && !lbit) ? got : r1[u];
    stg<NT>(st + i0[u], n0);
    stg<NT>(st + (i0[u] | hmask), n1);
  }
}

template <typename T, int U, bool GUARD, bool NT, int NP, typename E = amp_t<T>>
__global__ __launch_bounds__(kBlock) void k_swap_xlane1(E* __restrict__ st, uint64_t nitems,
                                                        Ins ins, uint32_t la_w, uint32_t lb_w, Sel low) {
  using A = E;
  if (GUARD && work_index<0>(0) >= nitems) return;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t ba = (lane >> la_w) & 1u, bb = (lane >> lb_w) & 1u;
  const uint32_t src = (lane & ~((1u << la_w) | (1u << lb_w))) | (bb << la_w) | (ba << lb_w);
  uint64_t idx[U];
  A x[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    idx[u] = insert_bits<NP>(work_index<Log2<U>::v>(u), ins);
    x[u] = ldg<NT>(st + idx[u]);
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const A got = shfl_e<A>(x[u], (int)src);
    stg<NT>(st + idx[u], sel_hit(idx[u], low) ? got : x[u]);
  }
}

// ---- several disjoint bit transpositions in ONE sweep (Swap with h >= 2) ---------------------------------------------
// Swap(h, A ++ B) is the product of the h transpositions (A[j] B[j]); moves are exact, so any grouping of them is
// bit-identical to the single permutation.  A lane holds the 2^NH amplitudes that differ on the NH "register" bits
// (swapped positions outside the lane index) and the transpositions act on that register file:
//   HH  both bits are register bits    -> nothing moves between lanes: the element loaded from combination c is
//       simply STORED at the combination with the two bits exchanged (off_st[c] = off_ld[pi_HH(c)], host tables);
//   HL  one register bit, one lane bit -> lanes whose lane bit differs from the register bit exchange with lane ^ bit
//       (the k_swap_xlane2 rule, for every combination of the other register bits);
//   LL  both bits are lane bits        -> every register comes from the lane with those bits exchanged; all LL
//       transpositions of the group fold into one source-lane map.
// The stages act on disjoint bits, so they commute.  Global accesses stay whole contiguous rows.
struct SwapNDesc {
  uint64_t off_ld[16], off_st[16];  // amplitude-index offset of register combination c (load / store side)
  uint32_t hl_lane[4];              // register bit r < NHL is exchanged with this lane bit
  uint32_t n_ll;
  uint32_t ll_a[4], ll_b[4];        // LL transpositions (lane bits)
};

// NH register bits, of which the first NHL are HL bits (the host orders them so; the others come in HH pairs);
// LL: the group has lane-lane transpositions.  NHL == 0 && !LL is the all-HH group: no data crosses lanes and the
// combinations no transposition moves are neither loaded nor stored.
template <typename T, int NH, int NHL, bool LL, int U, bool GUARD, bool NT, typename E = amp_t<T>>
__global__ __launch_bounds__(kBlock) void k_swapn(E* __restrict__ st, uint64_t nitems, Ins ins, SwapNDesc d, Sel low) {
  using A = E;
  constexpr int NR = 1 << NH;
  constexpr bool kPureHH = NHL == 0 && !LL;
  static_assert((NH - NHL) % 2 == 0, "register bits that are not HL bits come in HH pairs");
  if (GUARD && work_index<0>(0) >= nitems) return;  // whole waves leave together (nitems % 64 == 0)
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t src = lane;
  if constexpr (LL) {
    for (uint32_t q = 0; q < d.n_ll; ++q) {
      const uint32_t ba = (src >> d.ll_a[q]) & 1u, bb = (src >> d.ll_b[q]) & 1u;
      src = (src & ~((1u << d.ll_a[q]) | (1u << d.ll_b[q]))) | (bb << d.ll_a[q]) | (ba << d.ll_b[q]);
    }
  }
  // all-HH: combination c is a fixed point when each HH pair (bits 2j, 2j+1) holds equal bits
  auto fixed = [](int c) {
    if (!kPureHH) return false;
    for (int j = 0; j < NH / 2; ++j)
      if (((c >> (2 * j)) & 1) != ((c >> (2 * j + 1)) & 1)) return false;
    return true;
  };
  uint64_t i0[U];
  A x[U][NR];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    i0[u] = insert_bits<-1>(work_index<Log2<U>::v>(u), ins);
#pragma unroll
    for (int c = 0; c < NR; ++c)
      if (!fixed(c)) x[u][c] = ldg<NT>(st + (i0[u] | d.off_ld[c]));
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    // controls below kLineBits: a lane whose low control bits are not all 1 keeps everything it loaded.  Its HL / LL
    // partner lanes have the same control bits (controls are never swapped bits), so switching the exchange off per
    // lane is consistent: the registers are permuted in place, no second copy of the group is kept.
    const bool hit = sel_hit(i0[u], low);
    A(&y)[NR] = x[u];
#pragma unroll
    for (int r = 0; r < NHL; ++r) {
      const uint32_t la = d.hl_lane[r];
      const bool lbit = (lane >> la) & 1u;
      const bool take0 = hit && lbit, take1 = hit && !lbit;
      // slot (reg bit r, lane bit l) <-> slot (l, r): lanes with l = 1 give away their r = 0 element, lanes with
      // l = 0 their r = 1 element, and receive into the same slot
#pragma unroll
      for (int c = 0; c < NR; ++c)
        if (((c >> r) & 1) == 0) {
          const int k = c | (1 << r);
          const A yc = y[c], yk = y[k];
          const A got = shfl_xor_e<A>(lbit ? yc : yk, 1 << la);
          y[c] = take0 ? got : yc;  // selects on values, never a conditional store: that would address the register file
          y[k] = take1 ? got : yk;
        }
    }
    if constexpr (LL) {
      const int from = hit ? (int)src : (int)lane;
#pragma unroll
      for (int c = 0; c < NR; ++c) y[c] = shfl_e<A>(y[c], from);
    }
#pragma unroll
    for (int c = 0; c < NR; ++c)
      if (!fixed(c)) stg<NT>(st + (i0[u] | (hit ? d.off_st[c] : d.off_ld[c])), y[c]);  // HH: stored exchanged
  }
}

// ---- any permutation of the index bits in ONE out-of-place sweep ---------------------------------------------------
// out[j] = in[src(j)] where bit pi[d] of src(j) is bit d of j: the composition of any run of Swap ops (each one a
// product of bit transpositions, qubit_iterators.rs:208-218; pure moves, so the composition is bit-identical to the
// ops applied one after the other), the gather of a multi-GPU remap, a qubit relabelling.
// A block moves a tile of 2^TB elements chosen so that BOTH sides stream whole rows: the tile's destination bits are the R
// row bits of the destination, the destination bits that are fed by the source's row bits, and (r6, "split rows") index
// position 11 on both sides — thread bit 5, the upper half of a wave, is position 11, so a wave-level access is two 512-byte
// halves 32 KiB apart on the read side AND on the write side (byte-address bit 15: the r4 finding of the tile sweeps,
// profiles/r04_tile_rows.md, measured for this sweep in profiles/r06_permute.md: every permutation 5.2 - 5.9 ms at n = 30
// against 5.4 - 6.7 with contiguous rows).  Rows are read in source order, parked in LDS at their destination coordinate and
// written in destination order.  The LDS slot is the tile coordinate with up to FB of its higher bits XOR-folded into the low
// FB bits (FB = 3 for 16-byte elements, 4 for 8-byte ones: the lanes of one ds_write / ds_read bank group then hit distinct
// banks on both sides, MI355X_MICROARCH.md §LDS).
constexpr int kPermMaxTile = 13;  // (13: the pair form of 8-byte elements, k_permute_pairs)
struct PermDesc {
  uint32_t tbits[kPermMaxTile];    // destination position of tile-coordinate bit i (tbits[i] = i for i < R; thread bits 0..7, element bits 8..)
  uint32_t sbits[kPermMaxTile];    // source position of source-side coordinate bit i (sbits[i] = i for i < R)
  uint32_t u2c[kPermMaxTile];      // bit i of the source-side coordinate is bit u2c[i] of the tile (destination) coordinate
  uint32_t tsorted[kPermMaxTile];  // the tile's destination positions ascending (the block index fills the others)
  uint32_t nfold, fold_from[6], fold_to[6];
  uint32_t n_outer;                // destination positions outside the tile and the source position each one feeds
  unsigned char outer_dst[64], outer_src[64];
};
__device__ __forceinline__ uint32_t perm_fold(uint32_t c, const PermDesc& d) {
  uint32_t f = 0;
  for (uint32_t i = 0; i < d.nfold; ++i) f ^= ((c >> d.fold_from[i]) & 1u) << d.fold_to[i];
  return c ^ f;
}
template <typename A, int R, int TB, bool NT, int PIPE>
__global__ __launch_bounds__(kBlock) void k_permute_bits(const A* __restrict__ in, A* __restrict__ out, PermDesc d) {
  constexpr int E = (1 << TB) / kBlock, EB = TB - 8;  // elements per thread; coordinate bits 8.. come from e
  __shared__ __attribute__((aligned(16))) A tile[1 << TB];
  const uint32_t t = threadIdx.x;
  // thread part of: source offset, tile coordinate reached from the source side, destination offset
  uint64_t s_t = t & ((1u << R) - 1u), d_t = t & ((1u << R) - 1u);
  uint32_t c_t = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint32_t bit = (t >> i) & 1u;
    c_t |= bit << d.u2c[i];
    if (i >= R) {
      s_t |= (uint64_t)bit << d.sbits[i];
      d_t |= (uint64_t)bit << d.tbits[i];
    }
  }
  const uint32_t slot_ld = perm_fold(c_t, d), slot_st = perm_fold(t, d);  // the fold is linear: e's part is XOR-ed in below
  // a block moves PIPE consecutive tiles; the loads of tile p + 1 are in flight while tile p is read back from LDS and stored
  const uint64_t first = (blockIdx.x + (uint64_t)blockIdx.y * gridDim.x) * PIPE;
  uint64_t dbase, sbase;
  auto bases = [&](uint64_t b) {
    dbase = b;
#pragma unroll
    for (int i = 0; i < TB; ++i) {
      const uint32_t p = d.tsorted[i];
      dbase = ((dbase >> p) << (p + 1)) | (dbase & ((1ull << p) - 1ull));
    }
    sbase = 0;
    for (uint32_t i = 0; i < d.n_outer; ++i) sbase |= ((dbase >> d.outer_dst[i]) & 1ull) << d.outer_src[i];
  };
  A x[E];
  auto load = [&]() {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      uint64_t s_e = 0;
#pragma unroll
      for (int i = 0; i < EB; ++i) s_e |= (uint64_t)((e >> i) & 1) << d.sbits[8 + i];
      x[e] = ldg<NT>(in + (sbase | s_t | s_e));
    }
  };
  bases(first);
  load();
#pragma unroll
  for (int p = 0; p < PIPE; ++p) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      uint32_t c_e = 0;
#pragma unroll
      for (int i = 0; i < EB; ++i) c_e |= (uint32_t)((e >> i) & 1) << d.u2c[8 + i];
      tile[slot_ld ^ perm_fold(c_e, d)] = x[e];
    }
    __syncthreads();
    const uint64_t dcur = dbase;
    if (p + 1 < PIPE) {
      bases(first + p + 1);
      load();
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
      uint64_t d_e = 0;
#pragma unroll
      for (int i = 0; i < EB; ++i) d_e |= (uint64_t)((e >> i) & 1) << d.tbits[8 + i];
      stg<NT>(out + (dcur | d_t | d_e), tile[slot_st ^ perm_fold((uint32_t)e << 8, d)]);
    }
    if (p + 1 < PIPE) __syncthreads();
  }
}
// ---- the same sweep for 8-byte elements whose index bit 0 MOVES (Complex<f32>, unpacked view; r6) -------------------------
// k_permute_bits would read such a state in 512-byte wave rows of 64 elements, and the split that pays for 16-byte elements is two
// 256-byte pieces there (measured slower, profiles/r06_permute.md).  Here both global sides move 16-byte PAIRS — two elements
// that differ in index position 0 on that side — with thread bits 0..4 = positions 1..5 and thread bit 5 = position 12 (byte
// address bit 15) on the read side AND on the write side: the access pattern of the 16-byte kernel.  The 8-byte transposition
// happens inside LDS: a loaded pair is parked as two 8-byte elements at their destination coordinates (they differ in tile bit
// u2c[0]), a stored pair is one 16-byte LDS read (tile bit 0 = destination position 0; the slot's bit 0 is never folded).
// Tile = positions {0..5, 12} on both sides: 12 or 13 bits (32 / 64 KiB); a permutation that needs 14 stays with k_permute_bits.
// Coordinate bit 0 = the pair bit, bits 1..8 = the thread id, bits 9.. = the E units a thread moves.  PIPE as above.
template <int TB, bool NT, int PIPE>
__global__ __launch_bounds__(kBlock) void k_permute_pairs(const f32x4* __restrict__ in, f32x4* __restrict__ out, PermDesc d) {
  using A8 = amp_t<float>;
  constexpr int EB = TB - 9, E = 1 << EB;  // 16-byte units per thread
  __shared__ __attribute__((aligned(16))) A8 tile[1 << TB];
  const uint32_t t = threadIdx.x;
  uint64_t s_t = 0, d_t = 0;
  uint32_t c_t = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint32_t bit = (t >> i) & 1u;
    c_t |= bit << d.u2c[i + 1];
    s_t |= (uint64_t)bit << d.sbits[i + 1];
    d_t |= (uint64_t)bit << d.tbits[i + 1];
  }
  const uint32_t slot_ld = perm_fold(c_t, d), slot_hi = perm_fold(1u << d.u2c[0], d), slot_st = perm_fold(t << 1, d);
  const uint64_t first = (blockIdx.x + (uint64_t)blockIdx.y * gridDim.x) * PIPE;
  uint64_t dbase, sbase;
  auto bases = [&](uint64_t b) {
    dbase = b;
#pragma unroll
    for (int i = 0; i < TB; ++i) {
      const uint32_t p = d.tsorted[i];
      dbase = ((dbase >> p) << (p + 1)) | (dbase & ((1ull << p) - 1ull));
    }
    sbase = 0;
    for (uint32_t i = 0; i < d.n_outer; ++i) sbase |= ((dbase >> d.outer_dst[i]) & 1ull) << d.outer_src[i];
  };
  f32x4 x[E];
  auto load = [&]() {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      uint64_t s_e = 0;
#pragma unroll
      for (int i = 0; i < EB; ++i) s_e |= (uint64_t)((e >> i) & 1) << d.sbits[9 + i];
      x[e] = ldg<NT>(in + ((sbase | s_t | s_e) >> 1));
    }
  };
  bases(first);
  load();
#pragma unroll
  for (int p = 0; p < PIPE; ++p) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      uint32_t c_e = 0;
#pragma unroll
      for (int i = 0; i < EB; ++i) c_e |= (uint32_t)((e >> i) & 1) << d.u2c[9 + i];
      const uint32_t sl = slot_ld ^ perm_fold(c_e, d);
      A8 lo, hi;
      lo.x = x[e].x;
      lo.y = x[e].y;
      hi.x = x[e].z;
      hi.y = x[e].w;
      tile[sl] = lo;
      tile[sl ^ slot_hi] = hi;
    }
    __syncthreads();
    const uint64_t dcur = dbase;
    if (p + 1 < PIPE) {
      bases(first + p + 1);
      load();
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
      uint64_t d_e = 0;
#pragma unroll
      for (int i = 0; i < EB; ++i) d_e |= (uint64_t)((e >> i) & 1) << d.tbits[9 + i];
      const f32x4 v = *reinterpret_cast<const f32x4*>(&tile[slot_st ^ perm_fold((uint32_t)e << 9, d)]);
      stg<NT>(out + ((dcur | d_t | d_e) >> 1), v);
    }
    if (p + 1 < PIPE) __syncthreads();
  }
}
// states smaller than one tile: one element per thread, the source index bit by bit
struct PermSmall {
  uint32_t n;
  unsigned char pi[64];
};
template <typename A>
__global__ __launch_bounds__(kBlock) void k_permute_bits_small(const A* __restrict__ in, A* __restrict__ out, uint64_t count, PermSmall d) {
  const uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (j >= count) return;
  uint64_t src = 0;
  for (uint32_t b = 0; b < d.n; ++b) src |= ((j >> b) & 1ull) << d.pi[b];
  out[j] = in[src];
}

// ---- general diagonal gate on k qubits ------------------------------------------------
// Amplitude per lane inside the control subspace; factor = diag[sub-index] (table in the arena, served
// by the caches).  `tpos[j]` is the bit position of op index j (j = 0 is the MSB of the sub-index,
// matrix_ops.rs:12-21).  Every amplitude of the subspace is rewritten (1*x == x exactly for finite x),
// which keeps the sweep line-granular whatever the target bits are.
struct DiagDesc {
  uint32_t k;
  uint32_t tpos[32];
};

template <typename T, int U, bool GUARD, bool NT, int NP, typename E = amp_t<T>>
__global__ __launch_bounds__(kBlock) void k_diag(E* __restrict__ st, uint64_t count, Ins ins,
                                                 Sel low, DiagDesc d, const amp_t<T>* __restrict__ diag) {
  using A = E;
  using M = amp_t<T>;
  if (GUARD && work_index<0>(0) >= count) return;
  uint64_t idx[U];
  A x[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    idx[u] = insert_bits<NP>(work_index<Log2<U>::v>(u), ins);
    x[u] = ldg<NT>(st + idx[u]);
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    uint32_t sub = 0;
    for (uint32_t j = 0; j < d.k; ++j) sub = (sub << 1) | (uint32_t)((idx[u] >> d.tpos[j]) & 1ull);
    const M f = diag[sub];
    const bool unit = f.x == (T)1 && f.y == (T)0;
    stg<NT>(st + idx[u], (sel_hit(idx[u], low) && !unit) ? cmul(f, x[u]) : x[u]);
  }
}

// ---- dense k-qubit gate held in registers (k = 2..4) ------------------------------------
// One lane owns the 2^K amplitudes of one sub-space; `ins` opens the K target bits and the
// controls.  off[c] is the amplitude-index offset of sub-index c.  The matrix is read
// through wave-uniform (scalar) loads.  Zero entries are multiplied, not skipped: for
// finite amplitudes 0*x adds +-0, which leaves every sum IEEE-equal to the reference's.
template <typename T, int K, int U, bool GUARD, bool NT>
__global__ __launch_bounds__(kBlock) void k_gate_kq(amp_t<T>* __restrict__ st, uint64_t ngroups,
                                                    Ins ins, DiagDesc d,
                                                    const amp_t<T>* __restrict__ mat) {
  using A = amp_t<T>;
  constexpr int S = 1 << K;
  if (GUARD && work_index<0>(0) >= ngroups) return;
  uint64_t off[S];
#pragma unroll
  for (int c = 0; c < S; ++c) {
    uint64_t o = 0;
#pragma unroll
    for (int j = 0; j < K; ++j)
      if ((c >> (K - 1 - j)) & 1) o |= 1ull << d.tpos[j];
    off[c] = o;
  }
  // U groups per lane, 32 KiB apart in the work index, all U * 2^K loads issued before the first store
  uint64_t i0[U];
  A x[U][S];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    i0[u] = insert_bits<-1>(work_index<Log2<U>::v>(u), ins);
#pragma unroll
    for (int c = 0; c < S; ++c) x[u][c] = ldg<NT>(st + (i0[u] | off[c]));
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
#pragma unroll
    for (int r = 0; r < S; ++r) {
      A acc = czero<A>();
#pragma unroll
      for (int c = 0; c < S; ++c) acc = cadd(acc, cmul(mat[r * S + c], x[u][c]));
      stg<NT>(st + (i0[u] | off[r]), acc);
    }
  }
}

// ---- dense k-qubit gate on the f64 matrix cores (k = 3, 4, 5) ----------------------------------
// For k >= 3 the per-group update really is a small complex GEMM: with S = 2^k, the real form
// [y_re; y_im] = [[G_re, -G_im], [G_im, G_re]] [x_re; x_im] is a (2S x 2S) x (2S x 16) product for 16
// independent groups, i.e. (S/8)^2 tiles x (S/2) K-steps of v_mfma_f64_16x16x4_f64.  FLOPs per
// amplitude = 8S against 32 B of traffic (k=3: 2 flop/B, k=5: 8 flop/B; f64 MFMA peak / HBM peak
// ~ 10 flop/B), so the sweep stays HBM-bound through k = 5 while the VALU form does not.
//
// Mapping (one wave = 16 groups x S amplitudes per step; lane l: j = l & 15 group, q = l >> 4):
//   * sub-index bits are re-ordered by bit position (c~ bit b <-> b-th lowest target position) so
//     q = c~ & 3 walks the two lowest target bits and each 16-B load instruction covers the longest
//     contiguous runs the target positions allow; the 16 groups are the 4 lowest non-target bits;
//   * B operand of K-step s: lane supplies X[row (c~ = 4*(s>>1) + q, part = s&1)][group j], i.e. the
//     .x / .y of the ONE amplitude it loaded with a 16-B access (no re/im shuffles);
//   * f64 C/D layout (differs from every other dtype on gfx950): col = lane & 15,
//     row = (lane >> 4) + 4*reg.  Row (q + 4*reg) of tile rb is assigned to (c~' = 8*rb + 4*(reg>>1) + q,
//     part = reg & 1), so a lane ends up with re AND im of exactly the amplitudes it loaded and
//     stores them back in place with 16-B accesses;
//   * the A operand (gate matrix in that row/column order) is precomputed on the host per lane
//     (afrag[(rb*KS + s)*64 + lane]) and held in registers across the wave's grid-stride loop.
// Rounding: an MFMA is a k-ordered fma chain, so results differ from the reference's unfused fold
// by a few ulp (covered by the 1e-12 bar); 0/1 permutation matrices stay exact for finite data.
struct MfmaDesc {
  uint32_t tau[8];  // target bit positions, ascending
};

typedef double v4f64 __attribute__((ext_vector_type(4)));
typedef float v4f32 __attribute__((ext_vector_type(4)));

// one 16x16x4 step in the state's precision.  The f32 form (v_mfma_f32_16x16x4_f32) runs at the f32 VECTOR rate on gfx950:
// it is used for the same reason as the f64 one below k = 6 — operand bandwidth and whole-row addressing, not flops —
// and is an exact f32 fma chain.  Its C/D layout is row = 4 * (lane >> 4) + reg (f64: (lane >> 4) + 4 * reg); the host
// arranges the A rows accordingly (build_afrag), so both give a lane re and im of the amplitudes it loaded.
__device__ __forceinline__ v4f64 mfma16(double a, double b, v4f64 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ v4f32 mfma16(float a, float b, v4f32 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
template <typename T> struct Acc4;
template <> struct Acc4<double> { using type = v4f64; };
template <> struct Acc4<float> { using type = v4f32; };

// ---- r6: three real products instead of four for k >= 5 (and two for a real matrix) ---------------------------------------
// From k = 5 up the sweeps are bound by (k = 5: close to) the matrix pipe, so the real form above — four S x S real products
// per complex one — is replaced by three (Gauss): with Xs = X_re + X_im
//     K1 = G_re Xs,   Y_re = K1 - (G_re + G_im) X_im,   Y_im = K1 + (G_im - G_re) X_re
// as accumulator chains: acc = P Xs (S/4 steps of 16 complex rows x 4 columns), then re = acc + R X_im and im = acc + Q X_re
// with P = G_re, R = -(G_re + G_im), Q = G_im - G_re built on the host in double: 3 S^2 / 64 matrix instructions per item of
// 16 groups instead of S^2 / 16.  A matrix with no imaginary part needs P only: re = P X_re, im = P X_im, S^2 / 32 — and
// keeps a 0/1 matrix exact (every sum is one amplitude plus zeros), which the three-product form would not ((a + b) - b).
// Fragments: frag[((rb * NP + part) * (S/4) + s) * 64 + lane], NP = 3 (parts P, R, Q) or 1; 16 COMPLEX rows per block rb.
// C/D layout as above: a lane ends up with re (chain `re`) and im (chain `im`) of the amplitudes c~' = 16 rb + 4 reg + q, i.e.
// its own amplitudes m' = 4 rb + reg.  Rounding: a few ulp from the unfused fold (1e-12 bar), like every matrix-core form.
template <typename T, int K, int NP>
__device__ __forceinline__ void mfma3_item(const T (&a)[(1 << K) / 16][NP][(1 << K) / 4], const amp_t<T> (&x)[(1 << K) / 4],
                                           amp_t<T> (&y)[(1 << K) / 4]) {
  using V4 = typename Acc4<T>::type;
  constexpr int S = 1 << K, RB = S / 16, KS3 = S / 4;
  static_assert(RB % 2 == 0 && (NP == 1 || NP == 3), "shape");
#pragma unroll
  for (int rb = 0; rb < RB; rb += 2) {  // two row blocks at a time: independent chains over the same B operands
    V4 re0, im0, re1, im1;
    if constexpr (NP == 3) {
      V4 k0 = {(T)0, (T)0, (T)0, (T)0}, k1 = {(T)0, (T)0, (T)0, (T)0};
#pragma unroll
      for (int s = 0; s < KS3; ++s) {
        const T b = x[s].x + x[s].y;
        k0 = mfma16(a[rb][0][s], b, k0);
        k1 = mfma16(a[rb + 1][0][s], b, k1);
      }
      re0 = k0;
      im0 = k0;
      re1 = k1;
      im1 = k1;
#pragma unroll
      for (int s = 0; s < KS3; ++s) {
        re0 = mfma16(a[rb][1][s], x[s].y, re0);
        im0 = mfma16(a[rb][2][s], x[s].x, im0);
        re1 = mfma16(a[rb + 1][1][s], x[s].y, re1);
        im1 = mfma16(a[rb + 1][2][s], x[s].x, im1);
      }
    } else {
      re0 = im0 = re1 = im1 = V4{(T)0, (T)0, (T)0, (T)0};
#pragma unroll
      for (int s = 0; s < KS3; ++s) {
        re0 = mfma16(a[rb][0][s], x[s].x, re0);
        im0 = mfma16(a[rb][0][s], x[s].y, im0);
        re1 = mfma16(a[rb + 1][0][s], x[s].x, re1);
        im1 = mfma16(a[rb + 1][0][s], x[s].y, im1);
      }
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      y[4 * rb + reg].x = re0[reg];
      y[4 * rb + reg].y = im0[reg];
      y[4 * rb + 4 + reg].x = re1[reg];
      y[4 * rb + 4 + reg].y = im1[reg];
    }
  }
}

template <typename T, int K, int WU, bool NT>
__global__ __launch_bounds__(kBlock) void k_gate_kq_mfma(amp_t<T>* __restrict__ st,
                                                         uint64_t nitems, Ins ins, MfmaDesc d,
                                                         const T* __restrict__ afrag) {
  using A = amp_t<T>;
  using V4 = typename Acc4<T>::type;
  constexpr int S = 1 << K;
  constexpr int TT = S / 8;   // 16x16 tiles per dimension of the (2S x 2S) real matrix
  constexpr int KS = S / 2;   // K-steps of 4
  constexpr int NA = S / 4;   // amplitudes per lane per item
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t j = lane & 15u, q = lane >> 4;
  // A operand in registers for the whole wave lifetime.  (Measured alternative, k = 5 at n = 30: A in
  // LDS with the row-block loop rolled up frees 128 VGPRs but leaves only WU accumulator chains in
  // flight and ran 4.9 vs 5.3 TB/s: the f64 MFMA chain latency, not occupancy, is what has to be hidden.)
  T a[TT][KS];
#pragma unroll
  for (int rb = 0; rb < TT; ++rb)
#pragma unroll
    for (int s = 0; s < KS; ++s) a[rb][s] = afrag[(rb * KS + s) * 64 + lane];
  const uint64_t offq = ((uint64_t)(q & 1u) << d.tau[0]) | ((uint64_t)(q >> 1) << d.tau[1]);
  uint64_t offm[NA];
#pragma unroll
  for (int m = 0; m < NA; ++m) {
    uint64_t o = 0;
#pragma unroll
    for (int b = 0; b < K - 2; ++b)
      if ((m >> b) & 1) o |= 1ull << d.tau[b + 2];
    offm[m] = o;
  }
  auto load = [&](uint64_t w, A (&x)[WU][NA], uint64_t (&base)[WU]) {
#pragma unroll
    for (int i = 0; i < WU; ++i) {
      base[i] = insert_bits<-1>(((w + i) << 4) | j, ins) | offq;
#pragma unroll
      for (int m = 0; m < NA; ++m) x[i][m] = ldg<NT>(st + (base[i] | offm[m]));
    }
  };
  auto compute_store = [&](const A (&x)[WU][NA], const uint64_t (&base)[WU]) {
#pragma unroll
    for (int i = 0; i < WU; ++i) {
#pragma unroll
      for (int rb = 0; rb < TT; ++rb) {
        V4 acc = {(T)0, (T)0, (T)0, (T)0};
#pragma unroll
        for (int s = 0; s < KS; ++s) acc = mfma16(a[rb][s], (s & 1) ? x[i][s >> 1].y : x[i][s >> 1].x, acc);
        A y0, y1;
        y0.x = acc[0];
        y0.y = acc[1];
        y1.x = acc[2];
        y1.y = acc[3];
        stg<NT>(st + (base[i] | offm[2 * rb]), y0);
        stg<NT>(st + (base[i] | offm[2 * rb + 1]), y1);
      }
    }
  };
  // Grid-stride over the wave's items.  Measured alternatives at n = 30 (tools/bench_ops.py): prefetching
  // the next item's loads ahead of the MFMA chains (software pipeline) ran 5.08 vs 5.27 TB/s at k = 5 and
  // 4.07 vs 4.65 at k = 3 on bits 0-2, so the plain loop stays.
  const uint64_t step = (uint64_t)gridDim.x * (kBlock / 64) * WU;
  for (uint64_t w = ((uint64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * WU; w < nitems; w += step) {
    A x[WU][NA];
    uint64_t base[WU];
    load(w, x, base);
    compute_store(x, base);
  }
}

// ---- dense k-qubit gate on the matrix cores, k = 6, 7, 8: the gate matrix streams through LDS --------------------------
// Same lane mapping as k_gate_kq_mfma (a lane loads whole amplitudes with 16-B accesses, ends up with re AND im of exactly
// those amplitudes and stores them back in place), but the A operand no longer fits a wave's registers.  r6: the products are
// the three of mfma3_item's header (or the two of a matrix without an imaginary part): per block `rb` of 16 complex rows the
// parts P, R, Q (NP = 3; P alone for NP = 1), each S/4 K-steps = one chunk of the host-arranged fragment array
// (frag[((rb * NP + part) * (S/4) + s) * 64 + lane], L2-resident; 8 / 16 / 32 KiB per chunk in f64 for k = 6 / 7 / 8) —
// brought into LDS one chunk at a time, shared by the block's four waves (each on its own 16 groups) and double-buffered through
// registers: the loads of the next chunk are issued before the matrix instructions of this one and land in the other LDS half
// after them, one barrier per chunk.  X stays in registers for the whole item (S/4 amplitudes per lane); the four amplitudes a
// lane owns of a row block are written over the inputs when the block's last part is done.
// Roofline: 6 * 2^k flop per amplitude executed (8 * 2^k nominal; 4 * 2^k for a real matrix) against 32 B: at or past the f64
// matrix-core ridge (v_mfma_f64_16x16x4_f64 issues every 64 cycles per SIMD: 2048 flop / 64 clk * 1024 SIMDs * 2.4 GHz =
// 78.6 TFLOP/s, ridge 9.8 flop/B at 8 TB/s), and the A stream out of L2 (3 * 2^(2k) values per 64 groups) next in line at k = 8.
// Two accumulator chains per part (even / odd K-steps).  Measured at n = 30 (profiles/r06_dense_k5.md): k = 6 9.98 -> 8.68 ms,
// k = 7 18.4 -> 15.2, k = 8 38.2 -> 34.6; a real matrix: k = 6 7.3 ms, k = 8 18.4 ms.
template <typename T, int K, bool NT, int NP>
__global__ __launch_bounds__(kBlock, (K <= 7 || sizeof(T) == 4) ? 2 : 1) void k_gate_big_mfma(amp_t<T>* __restrict__ st, uint64_t nitems, Ins ins,
                                                          MfmaDesc d, const T* __restrict__ afrag) {
  using A = amp_t<T>;
  using V4 = typename Acc4<T>::type;
  constexpr int S = 1 << K;
  constexpr int RB = S / 16;       // blocks of 16 complex rows
  constexpr int KS3 = S / 4;       // K-steps of one part
  constexpr int NA = S / 4;        // amplitudes per lane per item
  constexpr int CH = KS3 * 64;     // values per chunk (one part of one row block)
  constexpr int NCH = RB * NP;     // chunks per item
  constexpr int PV = 16 / sizeof(T);     // values per 16-byte piece
  constexpr int PF = CH / (kBlock * PV); // 16-byte pieces per thread per chunk
  static_assert(PF >= 1 && (NP == 1 || NP == 3), "shape");
  typedef T v2f64 __attribute__((ext_vector_type(16 / sizeof(T))));  // one 16-byte piece of A
  __shared__ __attribute__((aligned(16))) T lds[2 * CH];  // two chunks
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t j = lane & 15u, q = lane >> 4;
  const uint64_t offq = ((uint64_t)(q & 1u) << d.tau[0]) | ((uint64_t)(q >> 1) << d.tau[1]);
  auto offm = [&](uint32_t m) {
    uint64_t o = 0;
#pragma unroll
    for (int b = 0; b < K - 2; ++b) o |= (uint64_t)((m >> b) & 1u) << d.tau[b + 2];
    return o;
  };
  const v2f64* __restrict__ af2 = reinterpret_cast<const v2f64*>(afrag);
  v2f64* lds2 = reinterpret_cast<v2f64*>(lds);
  // chunk 0 -> LDS half 0
#pragma unroll
  for (int p = 0; p < PF; ++p) lds2[p * kBlock + tid] = af2[p * kBlock + tid];
  __syncthreads();
  uint32_t buf = 0;
  const V4 zero = {(T)0, (T)0, (T)0, (T)0};
  const uint64_t nwg = (nitems + 3) / 4;
  for (uint64_t it = blockIdx.x; it < nwg; it += gridDim.x) {
    const uint64_t w = it * 4 + wave;
    const bool active = w < nitems;  // wave-uniform
    A x[NA];
    uint64_t base = 0;
    if (active) {
      base = insert_bits<-1>((w << 4) | j, ins) | offq;
#pragma unroll
      for (int m = 0; m < NA; ++m) x[m] = ldg<NT>(st + (base | offm((uint32_t)m)));
    } else {
#pragma unroll
      for (int m = 0; m < NA; ++m) x[m] = czero<A>();
    }
    V4 re = zero, im = zero;
    for (int c = 0; c < NCH; ++c) {
      // the next chunk of A (wrapping to chunk 0 for the next item) -> registers, in flight across the matrix instructions
      const int nc = c + 1 == NCH ? 0 : c + 1;
      v2f64 pre[PF];
#pragma unroll
      for (int p = 0; p < PF; ++p) pre[p] = af2[(size_t)nc * (CH / PV) + p * kBlock + tid];
      const T* a = lds + buf * CH + lane;
      const int part = NP == 1 ? 0 : c % NP;  // (wave-uniform)
      if (NP == 1) {
        V4 r0 = zero, i0 = zero;
#pragma unroll
        for (int s = 0; s < KS3; ++s) {
          r0 = mfma16(a[s * 64], x[s].x, r0);
          i0 = mfma16(a[s * 64], x[s].y, i0);
          if ((s & 7) == 7) __builtin_amdgcn_sched_barrier(0);  // (keeps the LDS operand loads from being hoisted en bloc: registers)
        }
        re = r0;
        im = i0;
      } else if (part == 0) {  // K1 = P (X_re + X_im): two chains (even / odd K-steps)
        V4 k0 = zero, k1 = zero;
#pragma unroll
        for (int s = 0; s < KS3; s += 2) {
          k0 = mfma16(a[s * 64], x[s].x + x[s].y, k0);
          k1 = mfma16(a[(s + 1) * 64], x[s + 1].x + x[s + 1].y, k1);
          if ((s & 7) == 6) __builtin_amdgcn_sched_barrier(0);
        }
        re = k0 + k1;
        im = re;
      } else if (part == 1) {  // re = K1 + R X_im
        V4 k1 = zero;
#pragma unroll
        for (int s = 0; s < KS3; s += 2) {
          re = mfma16(a[s * 64], x[s].y, re);
          k1 = mfma16(a[(s + 1) * 64], x[s + 1].y, k1);
          if ((s & 7) == 6) __builtin_amdgcn_sched_barrier(0);
        }
        re = re + k1;
      } else {                 // im = K1 + Q X_re
        V4 k1 = zero;
#pragma unroll
        for (int s = 0; s < KS3; s += 2) {
          im = mfma16(a[s * 64], x[s].x, im);
          k1 = mfma16(a[(s + 1) * 64], x[s + 1].x, k1);
          if ((s & 7) == 6) __builtin_amdgcn_sched_barrier(0);
        }
        im = im + k1;
      }
      if (active && (NP == 1 || part == 2)) {  // the row block is complete: its four amplitudes of this lane, over the inputs
        const uint32_t rb = (uint32_t)(c / NP);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          A y;
          y.x = re[reg];
          y.y = im[reg];
          stg<NT>(st + (base | offm(4u * rb + (uint32_t)reg)), y);
        }
      }
#pragma unroll
      for (int p = 0; p < PF; ++p) lds2[(buf ^ 1u) * (CH / PV) + p * kBlock + tid] = pre[p];
      __syncthreads();
      buf ^= 1u;
    }
  }
}

// ---- dense k-qubit gate on the matrix cores, k = 9, 10: X in LDS, the gate matrix streamed from L2 (r4) ------------------
// Nothing in the reference bounds k (qip-iterators/src/iterators/ops.rs:13); beyond k = 8 the 2^k amplitudes of 16 groups no
// longer fit a wave's registers (k_gate_big_mfma keeps X there).  Here a block of eight waves owns ONE item of 16 groups: its
// 16 x 2^k amplitudes (up to 128 KiB) live in LDS for the whole item (one phase of it, see below), every wave reads its B operands from there (one ds_read_b128
// feeds two K-steps, conflict-free: the 64 lanes of a K-step pair read 64 consecutive amplitudes), and the waves split the
// OUTPUT rows: wave v computes the 16-row blocks rb = v, v + 8, ... over ALL K-steps, so every A fragment is needed by
// exactly one wave and goes from the L2-resident fragment array straight into registers (16-byte loads holding two
// consecutive K-steps, eight pairs prefetched ahead of the matrix instructions), never through LDS.  The results stay in
// registers (2 amplitudes per lane per row block) until the item's last row block is done — X in LDS is never overwritten —
// and are then stored over the inputs.  Same lane mapping as k_gate_kq_mfma; the products and the host-built fragments are those
// of mfma3_item (three real products per complex one, two for a real matrix; fragments re-ordered in pairs of K-steps).  Where
// 16 groups x 2^k amplitudes exceed the LDS (Complex<f64>, k = 10: 256 KiB) the K dimension is walked in two phases, each with
// half of X in LDS, the running sums of a wave's row blocks carried in registers.
// Roofline: 6 * 2^k flop per amplitude executed (8 * 2^k nominal) against 32 B: bound by the f64 matrix pipe (78.6 TFLOP/s), with
// the A stream (1.5 * 2^(k+1) bytes of L2 traffic per amplitude: every block walks the whole 6- / 24-MiB fragment array per item)
// next in line.  n = 30 (profiles/r06_dense_k5.md): k = 9 79 -> 59 ms, k = 10 162 - 188 -> 129 ms.
struct HugeDesc {
  uint32_t tau[12];  // target bit positions, ascending
};
// r6: the products are the three of mfma3_item's header (NP = 3: parts P, R, Q; NP = 1: P alone for a matrix without an imaginary
// part), per block of 16 COMPLEX rows: three independent accumulator chains (K1, re, im) over the item's K-steps, re += K1 and
// im += K1 at the end of a phase.  Fragments in pairs of consecutive K-steps: frag[(((rb * (S/8) + sp) * NP + part) * 64 + lane) * 2 + e].
template <typename T, int K, int NPH, bool NT, int NP>
__global__ __launch_bounds__(512) void k_gate_huge_mfma(amp_t<T>* __restrict__ st, uint64_t nitems, Ins ins, HugeDesc d,
                                                       const T* __restrict__ afrag2) {
  using A = amp_t<T>;
  using V4 = typename Acc4<T>::type;
  constexpr int NG = 16;        // groups per item (the 16 columns of the matrix instruction)
  constexpr int S = 1 << K;
  constexpr int RB = S / 16;    // blocks of 16 complex rows
  constexpr int KP = S / 8;     // K-step PAIRS per part (two amplitudes of X each)
  constexpr int KPH = KP / NPH; // ... per phase: NPH > 1 when 16 x 2^K amplitudes exceed the LDS (Complex<f64>, K = 10) — the
                                // K dimension is walked in NPH phases, each with its share of X in LDS, the row blocks'
                                // running sums carried in registers from phase to phase
  constexpr int NW = 8;         // waves per block
  constexpr int RBW = RB / NW;  // row blocks per wave
  constexpr int PF = (NPH > 1 || NP == 1) ? 2 : 4;  // K-step pairs of A in flight ahead of the matrix instructions (registers: the carried sums of NPH > 1)
  static_assert(KPH % PF == 0 && RB % NW == 0 && KP % NPH == 0 && (NP == 1 || NP == 3), "shape");
  __shared__ __attribute__((aligned(16))) A xl[(S / NPH) * NG];  // xl[(c~ - first c~ of the phase) * NG + group]
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t j = lane & 15u, q = lane >> 4;
  const uint64_t offq = ((uint64_t)(q & 1u) << d.tau[0]) | ((uint64_t)(q >> 1) << d.tau[1]);
  auto offm = [&](uint32_t m) {
    uint64_t o = 0;
#pragma unroll
    for (int b = 0; b < K - 2; ++b) o |= (uint64_t)((m >> b) & 1u) << d.tau[b + 2];
    return o;
  };
  typedef T pair_t __attribute__((ext_vector_type(2)));  // A values of K-steps 2p and 2p + 1
  const pair_t* __restrict__ a2 = reinterpret_cast<const pair_t*>(afrag2);
  const V4 zero = {(T)0, (T)0, (T)0, (T)0};
  constexpr int KSPH = (S / 4) / NPH;  // amplitudes of X per lane and phase
  for (uint64_t w = blockIdx.x; w < nitems; w += gridDim.x) {
    const uint64_t base = insert_bits<-1>(w * NG + j, ins) | offq;
    V4 yre[RBW], yim[RBW];  // the row blocks' sums so far; statically indexed (a rolled loop picks its slot by comparison): registers
#pragma unroll
    for (int r = 0; r < RBW; ++r) yre[r] = yim[r] = zero;
#pragma unroll 1
    for (int ph = 0; ph < NPH; ++ph) {
      if (ph) __syncthreads();  // every wave is done with the previous phase's share of X
      // X -> LDS: wave v brings the sub-indices c~ = 4 m + q with m = first + v, first + v + NW, ...
#pragma unroll 4
      for (uint32_t m = wave; m < (uint32_t)KSPH; m += NW)
        xl[(4u * m + q) * NG + j] = ldg<NT>(st + (base | offm((uint32_t)ph * KSPH + m)));
      __syncthreads();
#pragma unroll 1
      for (int rbi = 0; rbi < RBW; ++rbi) {
        const uint32_t rb = wave + (uint32_t)rbi * NW;
        const pair_t* ap = a2 + (((size_t)rb * KP + (size_t)ph * KPH) * NP) * 64 + lane;  // + (sp * NP + part) * 64
        V4 k1 = zero, re = zero, im = zero;
        pair_t cur[PF][NP], nxt[PF][NP];
#pragma unroll
        for (int u = 0; u < PF; ++u)
#pragma unroll
          for (int pt = 0; pt < NP; ++pt) cur[u][pt] = ap[(size_t)(u * NP + pt) * 64];
#pragma unroll 1
        for (int p0 = 0; p0 < KPH; p0 += PF) {
          if (p0 + PF < KPH) {
#pragma unroll
            for (int u = 0; u < PF; ++u)
#pragma unroll
              for (int pt = 0; pt < NP; ++pt) nxt[u][pt] = ap[(size_t)((p0 + PF + u) * NP + pt) * 64];
          }
#pragma unroll
          for (int u = 0; u < PF; ++u) {
            const A xa = xl[(4u * (uint32_t)(2 * (p0 + u)) + q) * NG + j], xb = xl[(4u * (uint32_t)(2 * (p0 + u) + 1) + q) * NG + j];
            if constexpr (NP == 3) {
              k1 = mfma16(cur[u][0].x, xa.x + xa.y, k1);
              re = mfma16(cur[u][1].x, xa.y, re);
              im = mfma16(cur[u][2].x, xa.x, im);
              k1 = mfma16(cur[u][0].y, xb.x + xb.y, k1);
              re = mfma16(cur[u][1].y, xb.y, re);
              im = mfma16(cur[u][2].y, xb.x, im);
            } else {
              re = mfma16(cur[u][0].x, xa.x, re);
              im = mfma16(cur[u][0].x, xa.y, im);
              re = mfma16(cur[u][0].y, xb.x, re);
              im = mfma16(cur[u][0].y, xb.y, im);
            }
          }
#pragma unroll
          for (int u = 0; u < PF; ++u)
#pragma unroll
            for (int pt = 0; pt < NP; ++pt) cur[u][pt] = nxt[u][pt];
        }
#pragma unroll
        for (int r = 0; r < RBW; ++r)
          if (r == rbi) {
            yre[r] = yre[r] + (NP == 3 ? re + k1 : re);
            yim[r] = yim[r] + (NP == 3 ? im + k1 : im);
          }
      }
    }
#pragma unroll
    for (int r = 0; r < RBW; ++r) {
      const uint32_t rb = wave + (uint32_t)r * NW;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        A y;
        y.x = yre[r][reg];
        y.y = yim[r][reg];
        stg<NT>(st + (base | offm(4u * rb + (uint32_t)reg)), y);
      }
    }
    __syncthreads();  // every wave is done reading this item's X before the next item overwrites it
  }
}

// ---- SparseMatrix on k <= 5 qubits, in place ---------------------------------------------------------------------
// SparseMatrixOpIterator (qubit_iterators.rs:60-102): row r of the op is its stored list of (column, value), applied
// in stored order with nothing filtered; out[r] = 0 + v_0 * x[c_0] + v_1 * x[c_1] + ...  One lane owns one group of
// 2^K amplitudes (as in k_gate_kq: same addresses, same coalescing) and parks them in its own LDS column
// (slot [c][tid]: conflict-free, nobody else touches the column, so no barrier), because the column a stored entry
// names is a run-time value and registers cannot be indexed by one.  Rows are then folded exactly like the literal
// kernel does and written straight over the input: every input of the group is already in LDS.
// LDS: 2^K * block * 16 B = 32 KiB for every K (block = 256 lanes up to K = 3, 128 at K = 4, 64 at K = 5).
template <typename T, int K, bool NT>
__global__ void k_sparse_kq(amp_t<T>* __restrict__ st, uint64_t ngroups, Ins ins, DiagDesc d,
                            const uint64_t* __restrict__ rowptr, const uint64_t* __restrict__ cols,
                            const amp_t<T>* __restrict__ vals) {
  using A = amp_t<T>;
  constexpr int S = 1 << K;
  extern __shared__ __attribute__((aligned(16))) unsigned char sparse_raw[];
  A* col = reinterpret_cast<A*>(sparse_raw);
  const uint32_t nthr = blockDim.x;
  const uint64_t w = (blockIdx.x + (uint64_t)blockIdx.y * gridDim.x) * nthr + threadIdx.x;
  if (w >= ngroups) return;
  const uint64_t i0 = insert_bits<-1>(w, ins);
  uint64_t off[S];
#pragma unroll
  for (int c = 0; c < S; ++c) {
    uint64_t o = 0;
#pragma unroll
    for (int j = 0; j < K; ++j)
      if ((c >> (K - 1 - j)) & 1) o |= 1ull << d.tpos[j];
    off[c] = o;
  }
  A x[S];
#pragma unroll
  for (int c = 0; c < S; ++c) x[c] = ldg<NT>(st + (i0 | off[c]));
#pragma unroll
  for (int c = 0; c < S; ++c) col[c * nthr + threadIdx.x] = x[c];
  // same lane reads what it wrote: program order is enough (the compiler inserts the lgkmcnt wait)
#pragma unroll
  for (int r = 0; r < S; ++r) {
    A acc = czero<A>();
    const uint64_t pe = rowptr[r + 1];
    for (uint64_t p = rowptr[r]; p < pe; ++p) acc = cadd(acc, cmul(vals[p], col[(uint32_t)cols[p] * nthr + threadIdx.x]));
    stg<NT>(st + (i0 | off[r]), acc);
  }
}

// ---- SparseMatrix on k >= 6 qubits with few entries per row, out of place ---------------------------------------------
// The reference's own sparse bench shape (qip/benches/state_bench.rs:380-393: a 16-qubit SparseMatrix with one entry per
// row).  A group of 2^k amplitudes no longer fits a lane, so the sweep is the reference's gather formulation — one output
// row per lane item, rows in stored order folded from 0 (qubit_iterators.rs:60-102, ops.rs:104-110) — with the descriptor
// flattened for the device: rows in ELL form (at most E entries each: `nnz[m]`, `off[m * E + e]` = the column's bits already
// spread to their index positions, `val[m * E + e]`), the sub-index read off the row's index bits in runs of consecutive
// positions.  Rows whose controls are not all 1 are the iterator's single (row, 1) entry: a copy.  Every access of a lane's
// U items is a whole 1-KiB wave row on the output side; the input side is as contiguous as the op's columns allow (the
// identity of the bench: a plain copy).
struct EllDesc {
  uint64_t cmask;     // control bit positions (all must be 1)
  uint64_t opmask;    // the op's bit positions
  uint32_t nruns;     // the positions in ascending order as runs of consecutive bits
  uint32_t lo[32], len[32], shift[32];
};

// FULL: every row has exactly E entries (no count to fetch); CTL: the op has controls (rows outside their subspace copy).
// A lane's U rows go through the three dependent fetches — table entry, gathered amplitude, store — side by side: all table
// reads are issued before the first gather, all gathers before the first store.  Table slots beyond a row's count hold
// (offset 0, value 0): their loads are harmless and their terms are NOT added (the fold sees exactly the stored entries).
template <typename T, int E, int U, bool NT, bool FULL, bool CTL>
__global__ __launch_bounds__(kBlock) void k_sparse_ell(const amp_t<T>* __restrict__ in, amp_t<T>* __restrict__ out, EllDesc d,
                                                       const uint32_t* __restrict__ nnz, const uint64_t* __restrict__ off,
                                                       const amp_t<T>* __restrict__ val) {
  using A = amp_t<T>;
  uint64_t r[U], rbase[U];
  uint32_t m[U], cnt[U];
  A acc[U], keep[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    r[u] = work_index<Log2<U>::v>(u);
    uint32_t mm = 0;
    for (uint32_t j = 0; j < d.nruns; ++j) mm |= (uint32_t)((r[u] >> d.lo[j]) & ((1ull << d.len[j]) - 1ull)) << d.shift[j];
    m[u] = mm;
    rbase[u] = r[u] & ~d.opmask;
    acc[u] = czero<A>();
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    cnt[u] = FULL ? (uint32_t)E : nnz[m[u]];
    if (CTL) keep[u] = ldg<NT>(in + r[u]);
  }
#pragma unroll
  for (int e = 0; e < E; ++e) {
    uint64_t o[U];
    A v[U], x[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      o[u] = off[(uint64_t)m[u] * E + e];
      v[u] = val[(uint64_t)m[u] * E + e];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) x[u] = in[rbase[u] | o[u]];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (FULL || (uint32_t)e < cnt[u]) acc[u] = cadd(acc[u], cmul(v[u], x[u]));
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const bool active = !CTL || (r[u] & d.cmask) == d.cmask;
    stg<NT>(out + r[u], active ? acc[u] : keep[u]);  // outside the controls: 0 + 1 * x
  }
}

// ---- LDS-resident multi-gate sweep ------------------------------------------
// One sweep applies a whole LIST of gates: a 256-lane block stages a tile of 2^11 amplitudes in LDS
// (32 KiB for f64) — index bits 0..5 (one contiguous 1-KiB wave row) plus kTileHigh = 5 arbitrary
// higher bit positions chosen by the host scheduler — applies every gate of the segment to the tile
// (1-qubit dense / diagonal gates with any controls, bit swaps; all qubits inside the tile's 11 bits),
// and writes the tile back.  HBM traffic is ONE read + ONE write of the vector for the whole segment.
// Per gate the arithmetic is exactly that of k_gate1q_pair / k_phase / k_diag1q / k_swap_bits (same
// formulas, same zero-skipping, no FMA), so a segment that keeps the circuit's gate order is IEEE-equal
// to applying its gates one sweep at a time.
// k_tile_gates (below) round-trips the tile through LDS once per gate: ~550 LDS cycles per gate per tile
// against ~6500 cycles of HBM time per tile per CU.  k_tile_passes (further down, the default) keeps the
// amplitudes in registers across a pass of gates; there a gate riding along costs ~0.2 ms at n = 30 against
// 6.4 ms for a sweep of its own, so the scheduler lets a segment grow to kTileMaxGates.
constexpr int kTileLow = 6;                        // contiguous low bits (one wave row)
constexpr int kTileHigh = 5;                       // free bit positions per segment
constexpr int kTileBits = kTileLow + kTileHigh;    // 2048 amplitudes per tile (32 KiB for f64)
constexpr int kTileLaneBits = kTileBits - 3;       // k_tile_passes: thread-id bits (a lane holds 2^3 elements)
constexpr int kTileBlock = 1 << kTileLaneBits;     // ... 256 lanes = 4 waves per tile
constexpr int kTileWaveBits = kTileLaneBits - kTileLow;  // tile bits 6.. that the wave id fills at load / store time
// Which amplitude-index positions the tile's six LOW bits (the lane id at load / store time) stand for: 0..4 and `p5`.
//   p5 = 5   one contiguous 1-KiB row per wave-level access (rounds 1-3);
//   p5 = 11  (r4, the default for 16-byte amplitudes) two 512-byte halves 32 KiB apart.  Measured on MI355X at n = 30
//            (tools/tune_tile probe, profiles/r04_tile_rows.md): with contiguous rows a light sweep takes 5.3 ms when the
//            tile's five high positions are 11..15 but 6.1 - 8.6 ms when they are scattered over the DRAM row bits
//            (positions >= 17), whatever the block order; with the split rows EVERY choice of high positions runs in
//            5.3 - 5.9 ms.  Only byte-address bit 15 does this (every other second-half distance: no gain), and four
//            256-byte pieces are worse than either.  Position 5 is then an ordinary position a segment may claim.
// The block number fills every position outside the tile, ascending: computed as before in the space where p5 and 5 have
// traded places (`ins` opens the high positions with 5 standing for p5) and then put right by exchanging those two bits.
__device__ __forceinline__ uint64_t tile_block_base(uint64_t blk, const Ins& ins, uint32_t p5) {
  uint64_t w = insert_bits<-1>(blk << kTileLow, ins);
  if (p5 != 5u) {  // (bit 5 of w is zero here: the tile's low bits are not the block's)
    const uint64_t b = (w >> p5) & 1ull;
    w = (w & ~(1ull << p5)) | (b << 5);
  }
  return w;
}
__device__ __forceinline__ uint32_t tile_lane_off(uint32_t lane, uint32_t p5) { return (lane & 31u) | ((lane >> 5) << p5); }

// ---- SparseMatrix on k >= 6 qubits, in place through an LDS-staged tile (r4) ---------------------------------------------
// k_sparse_ell gathers straight from HBM: a row with two stored entries reads two input rows (48 instead of 32 bytes per
// amplitude when the partner rows are too far apart for the L2 to hold, 52 % of the HBM peak at n = 30), and an op that
// touches positions inside a wave row gathers 16-byte pieces.  Here a block stages the op's whole GROUP beside the wave row:
// the tile is the six low positions (the lane at load / store time, tile_lane_off: 0..4 and p5) plus the op's `kh` other
// positions, 2^(6+kh) amplitudes in LDS (kh <= 6 for Complex<f64> at two blocks per CU, 7 = 128 KiB at one), loaded and
// stored as whole wave rows exactly like a tile sweep, every input of every row of the group present once.  A lane owns 8
// rows of the tile; its output row folds the stored entries in stored order from 0 (qubit_iterators.rs:60-102, ops.rs:104-110)
// out of LDS, so the result is bit-equal to the literal kernel's.  The table is indexed by m' = the op's lane bits (ascending)
// then the tile row; `slot` = the stored column's place in the tile with the lane's other bits zero.
// Controls: outside the tile they are taken off the grid (`ins` holds them as ones), inside the lane they are a predicate.
struct SparseTileDesc {
  uint32_t kh;        // the op's positions outside the tile's low six, ascending = tile bits 6 ..
  uint32_t hpos[8];
  uint32_t p5;        // position of lane bit 5 (tile_block_base)
  uint32_t low_op;    // lane bits that are op positions
  uint32_t nlow;      // how many
  uint32_t low_ctl;   // lane bits that are controls (all must read 1 for the row to be touched)
};

// TL (r6): the op's table — `rows` x E (slot, value) entries + the row counts — is copied into LDS behind the tile by the block and
// read from there.  With wider rows a lane issues 8 x (1 + 2 E) table loads per tile through the vector memory path, which, not
// HBM, then bounds the sweep (Complex<f32>, k = 7, four entries per row: 36 % of the HBM peak; profiles/r06_sparse_tile.md); from
// LDS they cost a fraction.  Used for Complex<f32> rows of four entries when the table is at most 16 KiB (the launcher's comment
// has the cases where it measured no better).
template <typename T, int E, bool NT, bool TL>
__global__ void k_sparse_tile(amp_t<T>* __restrict__ st, uint64_t ntiles, Ins ins, SparseTileDesc d,
                              const uint32_t* __restrict__ nnz, const uint32_t* __restrict__ slot,
                              const amp_t<T>* __restrict__ val, uint32_t rows) {
  using A = amp_t<T>;
  constexpr int R = 8;  // rows of the tile per wave
  extern __shared__ __attribute__((aligned(16))) unsigned char sparse_tile_raw[];
  A* tile = reinterpret_cast<A*>(sparse_tile_raw);
  A* lval = tile + ((size_t)64 << d.kh);                          // TL: rows * E values ...
  uint32_t* lslot = reinterpret_cast<uint32_t*>(lval + (size_t)rows * E);  // ... their slots ...
  uint32_t* lnnz = lslot + (size_t)rows * E;                      // ... and the rows' counts
  const uint64_t blk = blockIdx.x + (uint64_t)blockIdx.y * gridDim.x;
  if (blk >= ntiles) return;
  const uint32_t lane = threadIdx.x & 63u, nw = blockDim.x >> 6;
  const uint32_t wave_v = threadIdx.x >> 6;                           // as the lanes see it: indexes the table with VECTOR loads (below)
  const uint32_t wave = __builtin_amdgcn_readfirstlane(wave_v);       // wave-uniform: the row-address arithmetic is scalar
  const uint64_t base = tile_block_base(blk, ins, d.p5) | tile_lane_off(lane, d.p5);
  // row i * nw + wave of the tile: the wave number fills the low kh - 3 tile-row bits, i the top three
  uint64_t offw = 0;
  for (uint32_t j = 0; j + 3 < d.kh; ++j) offw |= (uint64_t)((wave >> j) & 1u) << d.hpos[j];
  const uint64_t h0 = 1ull << d.hpos[d.kh - 3], h1 = 1ull << d.hpos[d.kh - 2], h2 = 1ull << d.hpos[d.kh - 1];
  A x[R];
  uint64_t g[R];
#pragma unroll
  for (int i = 0; i < R; ++i) {
    g[i] = base | offw | ((i & 1) ? h0 : 0ull) | ((i & 2) ? h1 : 0ull) | ((i & 4) ? h2 : 0ull);
    x[i] = ldg<NT>(st + g[i]);
  }
  uint32_t ml = 0;  // the op's lane bits, packed
  for (uint32_t b = 0, o = 0; b < 6u; ++b)
    if ((d.low_op >> b) & 1u) ml |= ((lane >> b) & 1u) << o++;
  const uint32_t keep = lane & ~d.low_op;
  const bool active = (lane & d.low_ctl) == d.low_ctl;
  // The rows' table entries do not depend on the tile.  One entry per row: fetched while the tile's loads are in flight.  Wider rows
  // are fetched after the barrier (8 rows x E entries ahead of time cost more registers than the occupancy can spare).  Always
  // through vector loads, even where a wave's 64 lanes share m — measured at n = 30, k = 6, two entries per row: vector loads after
  // the barrier 74.7 % of the HBM peak, scalar loads after it 71.2 % (they share the LDS reads' counter), scalar loads ahead 68.0 %.
  constexpr bool PRE = E == 1;
  uint32_t cnt[R], sl[R][E];
  A v[R][E];
  if constexpr (PRE) {
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const uint32_t m = ml | (((uint32_t)i * nw + wave_v) << d.nlow);
      cnt[i] = nnz[m];
      sl[i][0] = slot[m];
      v[i][0] = val[m];
    }
  }
  if constexpr (TL) {  // (the table does not depend on the tile: it arrives while the tile's loads are in flight)
    for (uint32_t i = threadIdx.x; i < rows * (uint32_t)E; i += blockDim.x) {
      lval[i] = val[i];
      lslot[i] = slot[i];
    }
    for (uint32_t i = threadIdx.x; i < rows; i += blockDim.x) lnnz[i] = nnz[i];
  }
#pragma unroll
  for (int i = 0; i < R; ++i) tile[(((uint32_t)i * nw + wave) << 6) | lane] = x[i];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const uint32_t m = ml | (((uint32_t)i * nw + wave_v) << d.nlow);
    if constexpr (!PRE) cnt[i] = TL ? lnnz[m] : nnz[m];
    A acc = czero<A>();
#pragma unroll
    for (int e = 0; e < E; ++e) {
      if constexpr (!PRE) {
        sl[i][e] = TL ? lslot[m * E + e] : slot[m * E + e];  // (slots beyond a row's count hold 0: a harmless read, not added)
        v[i][e] = TL ? lval[m * E + e] : val[m * E + e];
      }
      const A xx = tile[sl[i][e] | keep];
      if ((uint32_t)e < cnt[i]) acc = cadd(acc, cmul(v[i][e], xx));
    }
    if (active) stg<NT>(st + g[i], acc);
  }
}

// A tile sweep that stores its tiles ELSEWHERE and permuted (r4): the multi-GPU remap gathers the g leaving qubits' bit positions
// into the top g local positions before the all-to-all — a full out-of-place sweep of its own (k_pack_bits) unless the sweep
// that precedes it writes its rows straight to their packed places: destination index = the source index with the bits at
// `sel` taken out (every other bit keeps its relative order) and put on top, bit sel[t] -> position Lg + t.  The map moves
// bits, so it distributes over the disjoint parts of an address (block base | wave bits | access bits | lane offset).
// None of `sel` may be a low (lane) position: a row stays a row (two 512-byte halves, or one KiB).
struct TileStorePerm {
  uint32_t g, Lg;        // g = 0: store in place (the ordinary sweep)
  uint32_t sel[8];       // source position of destination bit Lg + t
  uint32_t sel_desc[8];  // the same positions, descending (removal order)
};
__device__ __forceinline__ uint64_t tile_packed_index(uint64_t x, const TileStorePerm& sp) {
  uint64_t top = 0;
  for (uint32_t t = 0; t < sp.g; ++t) top |= ((x >> sp.sel[t]) & 1ull) << (sp.Lg + t);
  for (uint32_t t = 0; t < sp.g; ++t) {
    const uint32_t p = sp.sel_desc[t];
    x = ((x >> (p + 1)) << p) | (x & ((1ull << p) - 1ull));
  }
  return x | top;
}
// (Measured in round 2: kTileHigh = 6 — 64-KiB tiles, 512 lanes, 2 blocks per CU — cuts the configs[1] circuit from 19
// to 15 sweeps but each sweep takes 11.8 ms instead of 6.8: 177 vs 129 ms.  Five resident blocks per CU are what
// overlaps the load / LDS / store phases; profiles/r02_tile_variants.md.)
// A gate riding along costs 0.05-0.2 ms at n = 30, a new sweep ~6 ms, so segments may grow long.  Two caps: gates that
// EXCHANGE amplitudes (dense targets, swaps) each may open a pass, and the pass table travels as a kernel argument;
// diagonal gates (QFT's 435 controlled phases) open none and only lengthen the gate list in the arena.
constexpr int kTileMaxExchGates = 64;
constexpr int kTileMaxGates = 256;

// Only the bits a gate EXCHANGES amplitudes across must lie inside the tile: the target of a dense gate, the
// two bits of a swap.  Controls, and the target of a diagonal gate, may sit on any index bit: outside the tile
// such a bit is constant for the whole block, so it is tested once against the block's base index
// (`omask` / `tpos_out`, amplitude-index space) instead of per element (`cmask` / `b0`, tile-index space).
constexpr uint32_t kTileOutside = 0xffffffffu;
template <typename T> struct TileGate {
  uint32_t kind;      // 0 = dense 1-qubit (pair update), 1 = diagonal 1-qubit (factor by target bit), 2 = bit swap,
                      // 3 = dense 2-qubit (b0 = bit of the sub-index MSB, b1 = LSB; nz = index of its 4x4 matrix),
                      // 4 = dense 3-qubit (b0, b1, tpos_out = tile bits of the sub-index MSB, middle, LSB; nz = offset of its
                      //     8x8 matrix in the matrix block, in units of 16 entries)
  uint32_t b0, b1;    // tile-index bit(s): target (kinds 0, 1; kTileOutside for a diagonal target outside the tile)
                      // or the two swapped bits (kind 2, b0 < b1).  Kind 0 keeps flags in b1:
                      //   bit 0: every matrix entry is real  -> 2 multiplies per product instead of 4 mul + 2 add
                      //   bit 1: the gate is X ([0,1;1,0])   -> the pair is exchanged, no arithmetic
                      // both give IEEE-equal results for finite amplitudes (x*1 == x, a - 0*b == a); inside a
                      // tile sweep VALU issue, not HBM, is the limit, so instructions matter here
  uint32_t cmask;     // tile-index bits that must all be 1 (controls inside the tile)
  uint32_t nz;        // kind 0: non-zero mask of the 2x2 entries
  uint32_t tpos_out;  // kind 1 with b0 == kTileOutside: amplitude-index position of the target
  uint64_t omask;     // amplitude-index bits outside the tile that must all be 1 (controls outside the tile)
  // k_tile_passes only, resolved by the host against the pass the gate belongs to — the kernel is bound by
  // instruction issue, scalar instructions included, so nothing that depends only on (gate, pass) is recomputed
  // per wave: `op` selects the code path (TileOp), `cm_reg` / `cm_lane` are cmask split into the controls on pass
  // bits (wave-uniform per element) and on lane bits (one predicate per gate)
  uint32_t op, cm_reg, cm_lane, pad_;
  amp_t<T> m[4];      // kind 0: 2x2 row-major; kind 1: m[0] = d0, m[1] = d1
};

// Code paths of k_tile_passes.  J* = index (0..2) of the pass bit the gate exchanges across / tests.
enum TileOp : uint32_t {
  TOP_DIAG_UNIFORM = 0,  // diagonal, target outside the tile, no lane-bit control: one wave-uniform factor
  TOP_DIAG_LANE,         // diagonal, target on a lane bit (or outside), no lane-bit control: per-lane factor
  TOP_DIAG_LANE_CTL,     // ... with lane-bit controls folded into the factor
  TOP_DIAG_REG0, TOP_DIAG_REG1, TOP_DIAG_REG2,  // diagonal, target = pass bit J
  TOP_DENSE0, TOP_DENSE1, TOP_DENSE2,           // dense 1-qubit on pass bit J, controls (if any) on pass bits
  TOP_DENSE_LANE0, TOP_DENSE_LANE1, TOP_DENSE_LANE2,  // ... with lane-bit controls (select per lane)
  TOP_DENSE2Q_01, TOP_DENSE2Q_02, TOP_DENSE2Q_10, TOP_DENSE2Q_12, TOP_DENSE2Q_20, TOP_DENSE2Q_21,  // JA (MSB), JB
  TOP_SWAP_01, TOP_SWAP_02, TOP_SWAP_12,
  // dense 3-qubit gate whose three targets ARE the pass's three bits: JA JB JC = pass-bit index of the sub-index MSB, middle, LSB
  TOP_DENSE3Q_012, TOP_DENSE3Q_021, TOP_DENSE3Q_102, TOP_DENSE3Q_120, TOP_DENSE3Q_201, TOP_DENSE3Q_210,
  // r5: a RUN of consecutive diagonal gates of one pass, as TileDiagItem[b1] starting at item nz of the launch's item block
  // (interpreter launches only: the host rewrites the pass's gate list, tile_merge_diag_runs; plans and generators never see it)
  TOP_DIAG_RUN,
};

// One step of a diagonal run (r5).  Every diagonal tile gate — phase / controlled phase / Rz-like, target and controls on lane
// bits, pass bits or outside the tile — is one or two of these: the amplitudes a lane holds in elements whose pass-bit
// combination c satisfies (c & reg_mask) == reg_val are multiplied by F, where
//     F = (tb & sel_mask) ? f1 : f0        (sel_mask = 0: F = f1)            the target's entry, when the target is a lane bit
//     F = ((tb & lane_mask) == lane_val) ? F : (1, 0)                          lane-bit controls folded into the factor
// provided the block's base index satisfies (base & omask) == oval (controls and targets outside the tile: wave-uniform).
// These are operation for operation the products the per-op code paths above (TOP_DIAG_*) perform — same factor selection, same
// multiplications by (1, 0) where a lane-bit control is 0, same elements, same order — so a run is bit-identical to its gates
// one by one; what goes away is the per-gate decoding: a 128-byte descriptor, a jump table and the branches of five code paths
// against 64 (Complex<f32>: 48) bytes and one loop (QFT's segments are runs of ~30 controlled phases per H).
// `emask` (host-resolved against the item's pass): bit i = element i of the lane's eight satisfies the pass-bit condition, so the
// kernel tests one bit per element; the condition itself (reg_mask / reg_val, tile-index space) is kept for the CPU replay.
// Complex<f64>: 64 bytes = ONE scalar load per step (sel_mask travels in the top byte of `emask` as bit position + 1).
template <typename T> struct alignas(16) TileDiagItem {
  amp_t<T> f0, f1;
  uint64_t omask, oval;
  uint32_t lane_mask, lane_val, emask_sel, reg_pack;  // emask_sel = emask | (sel bit + 1) << 24; reg_pack = reg_mask | reg_val << 16
};
static_assert(sizeof(TileDiagItem<double>) == 64 && sizeof(TileDiagItem<float>) == 48, "one (f64) / two scalar loads per diagonal step");

struct TileDesc {
  uint32_t ngates;
  uint32_t hpos[kTileHigh];  // amplitude-index bit positions of tile bits 6..10 (ascending)
  uint32_t p5;               // amplitude-index position of tile bit 5 (see tile_block_base)
};

// The tile is resident in LDS memory and every gate is a read-modify-write of LDS by the whole block; one
// block per tile, and the hardware overlaps the blocks' load / LDS / store phases.  Measured alternatives for
// the 256-gate circuit at n = 30 (22 segments, ~11.6 gates per sweep): this form 9.3 ms per sweep = 1256
// gates/s; persistent blocks software-pipelined over tiles (next tile's loads in flight during the LDS phase)
// 11.1 ms; tile held in registers with lane-shuffle / LDS exchange classes 14 ms (each element then evaluates
// its own matrix row: twice the f64 multiplies); a third register-to-register class: 344 VGPRs,
// 3.6x slower.
template <typename T, bool NT>
__global__ __launch_bounds__(kBlock) void k_tile_gates(amp_t<T>* __restrict__ st, Ins ins, TileDesc d,
                                                       const TileGate<T>* __restrict__ gates) {
  using A = amp_t<T>;
  extern __shared__ __attribute__((aligned(16))) unsigned char tile_raw[];
  A* tile = reinterpret_cast<A*>(tile_raw);
  constexpr int PER = (1 << kTileBits) / kBlock;  // 8 amplitudes per lane
  // `ins` opens the kTileHigh high positions; the low kTileLow bits of the shifted work index are zero
  const uint64_t base = tile_block_base(blockIdx.x + (uint64_t)blockIdx.y * gridDim.x, ins, d.p5);
  uint64_t idx[PER];
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const uint32_t t = u * kBlock + threadIdx.x;
    const uint32_t h = t >> kTileLow;
    uint64_t off = tile_lane_off(t & ((1u << kTileLow) - 1u), d.p5);
#pragma unroll
    for (int j = 0; j < kTileHigh; ++j) off |= (uint64_t)((h >> j) & 1u) << d.hpos[j];
    idx[u] = base | off;
  }
  A x[PER];
#pragma unroll
  for (int u = 0; u < PER; ++u) x[u] = ldg<NT>(st + idx[u]);
#pragma unroll
  for (int u = 0; u < PER; ++u) tile[u * kBlock + threadIdx.x] = x[u];
  __syncthreads();
  for (uint32_t gi = 0; gi < d.ngates; ++gi) {
    const TileGate<T> g = gates[gi];  // wave-uniform
    if ((base & g.omask) != g.omask) continue;  // block-uniform: an outside control is 0 for this whole tile
    if (g.kind == 0) {
      const uint32_t low = (1u << g.b0) - 1u, bit = 1u << g.b0;
#pragma unroll
      for (int k = 0; k < PER / 2; ++k) {
        const uint32_t p = k * kBlock + threadIdx.x;
        const uint32_t t0 = ((p >> g.b0) << (g.b0 + 1)) | (p & low);
        if ((t0 & g.cmask) != g.cmask) continue;
        const A a0 = tile[t0], a1 = tile[t0 | bit];
        A r0 = czero<A>(), r1 = czero<A>();
        if (g.nz & 1u) r0 = cadd(r0, cmul(g.m[0], a0));
        if (g.nz & 2u) r0 = cadd(r0, cmul(g.m[1], a1));
        if (g.nz & 4u) r1 = cadd(r1, cmul(g.m[2], a0));
        if (g.nz & 8u) r1 = cadd(r1, cmul(g.m[3], a1));
        tile[t0] = r0;
        tile[t0 | bit] = r1;
      }
    } else if (g.kind == 1) {
      const bool out = g.b0 == kTileOutside;
      const bool out_bit = out && ((base >> g.tpos_out) & 1ull);
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const uint32_t t = u * kBlock + threadIdx.x;
        if ((t & g.cmask) != g.cmask) continue;
        const bool one = out ? out_bit : (((t >> g.b0) & 1u) != 0);
        const A f = one ? g.m[1] : g.m[0];
        if (f.x == (T)1 && f.y == (T)0) continue;  // unit entries leave the amplitude untouched
        tile[t] = cmul(f, tile[t]);
      }
    } else {
      const uint32_t lowa = (1u << g.b0) - 1u;
#pragma unroll
      for (int k = 0; k < PER / 4; ++k) {
        uint32_t p = k * kBlock + threadIdx.x;         // index over the tile with bits b0 < b1 removed
        p = ((p >> g.b0) << (g.b0 + 1)) | (p & lowa);  // open b0
        const uint32_t hi_part = p >> g.b1;            // open b1 (p already has b0 opened, so b1 is final)
        p = (hi_part << (g.b1 + 1)) | (p & ((1u << g.b1) - 1u));
        if ((p & g.cmask) != g.cmask) continue;
        const uint32_t ta = p | (1u << g.b0), tb = p | (1u << g.b1);
        const A va = tile[ta], vb = tile[tb];
        tile[ta] = vb;
        tile[tb] = va;
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < PER; ++u) x[u] = tile[u * kBlock + threadIdx.x];
#pragma unroll
  for (int u = 0; u < PER; ++u) stg<NT>(st + idx[u], x[u]);
}

// ---- the same segment, LDS traffic cut by "passes" ------------------------------------------------------
// k_tile_gates above is LDS-bandwidth bound (64 KiB of LDS traffic per gate per tile; ~17 gates per sweep cost
// twice the HBM time).  Here consecutive gates are grouped into PASSES of at most three distinct exchange
// bits: each lane pulls the 8 tile elements that are closed under those bits into registers (one LDS read +
// one LDS write per element per PASS), applies every gate of the pass to them in circuit order — register
// butterflies for dense gates, element-wise factors for diagonal gates, register permutations for swaps —
// and barriers only at pass boundaries.  Per element the operations and their order are those of the
// gate-by-gate path, so circuit-order segments stay IEEE-equal to it.
//
// LDS banking (MI355X_MICROARCH.md §LDS): a 16-byte ds_read_b128 is served in four 16-lane groups over 16 slots
// of 16 B, a ds_write_b128 in eight 8-lane groups over 8 slots; lanes of a group that hit one slot at different
// addresses serialise.  With the tile stored linearly a pass on the low bits {0,1,2} makes every lane of a group
// hit the same slot (8-way: the LDS phase then costs more than the HBM phase).  Two measures make every pass
// conflict-free unless it holds both bits of a pair (j, j+S) (then 2-way):
//   * the tile is stored swizzled, slot(t) = t ^ ((t >> S) & (2^S - 1)), S = 4 for 16-byte amplitudes (5 for
//     8-byte ones, whose reads are served in 32-lane groups over 32 slots);
//   * the host chooses, per pass, which lane-id bit fills which non-pass tile bit (`lanepos`), putting lane bit
//     j < S on bit j or j+S so the low S slot bits enumerate the lanes of a group.
struct TilePass {
  uint32_t first, count;  // gates[first .. first+count)
  uint32_t pb[3];         // the pass's three exchange bits (tile-index space, distinct, ascending)
  uint32_t pad_;
  uint64_t lanepos;       // nibble k = tile-index bit filled by bit k of the thread id (the kTileLaneBits non-pass bits)
};
template <typename A> __device__ __forceinline__ uint32_t tile_slot(uint32_t t) {
  constexpr uint32_t S = sizeof(A) == 16 ? 4u : 5u;
  return t ^ ((t >> S) & ((1u << S) - 1u));
}
constexpr int kTileMaxPasses = kTileMaxExchGates;  // only a gate with exchange bits can close a pass
struct TilePassDesc {
  uint32_t npasses;
  uint32_t hpos[kTileHigh];
  uint32_t p5;  // amplitude-index position of tile bit 5 (see tile_block_base)
  uint32_t pad_;
  TilePass pass[kTileMaxPasses];
};

// Arithmetic: exactly the unfused products and sums of the gate-by-gate kernels (zero entries skipped), so that
// circuit-order tile sweeps stay IEEE-equal to them.
//
// The sweep is bound by vector-instruction issue and per-wave latency, not by LDS or HBM (rocprofv3 PMC,
// profiles/r01_tile_pmc.md), so the gate loop is written to issue as few vector instructions as possible and to
// keep five blocks resident per CU:
//   * a control on a pass bit is wave-uniform per register element (scalar branch); the host makes the controls
//     of dense gates and swaps pass bits whenever the pass has room, so the dense code has no per-lane predicate
//     (a divergent `if` around the update makes the compiler keep, and move, two copies of the lane's eight
//     amplitudes per gate).  What is left — a diagonal gate's lane-bit controls, the rare dense gate with more
//     controls than a pass holds — folds the predicate into the factor / selects per lane;
//   * a diagonal gate whose target is a pass bit has a wave-uniform factor per element (unit factors skipped by
//     a scalar branch); on a lane bit the factor is selected once per gate, not per element;
//   * X is a register exchange; gates with real entries multiply two reals per product;
//   * the reference's leading "0 +" of every row sum is dropped: 0 + p == p under IEEE == (it only turns a -0
//     into +0), the same equality the X and real-entry forms rely on;
//   * explicit FMAs for tile = 2 (which is held to 1e-12 anyway) would halve the arithmetic, but every FMA
//     variant of this kernel spilled under the occupancy bound (430 VGPRs at 96, 48 at 128): 30 ms per sweep;
//   * what remains above the arithmetic is hipcc ping-ponging the lane's eight amplitudes between two register
//     sets across the gate loop (about 16 v_mov_b64 per gate).  Tried and measured worse: the updates as in-place
//     gfx950 inline assembly with "+v" operands (operands copied in and out: 2186 vs 1602 vector instructions
//     per wave for 20 Hadamards), scalar re[8]/im[8] registers with products-before-sums ordering (1825, and the
//     X exchange if-converted into selects: 2959 vs 1256);
//   * wave-uniform conditions stay BRANCHES (QIP_KEEP_BRANCH): if-converted they become speculative arithmetic
//     blended by v_cndmask — more instructions and ~100 more live registers;
//   * global addresses are a wave-uniform base plus the lane id.
#define QIP_KEEP_BRANCH() asm volatile("")

template <typename A> __device__ __forceinline__ A tile_sel(bool take, A yes, A no) {
  A r;
  r.x = take ? yes.x : no.x;
  r.y = take ? yes.y : no.y;
  return r;
}

// one row of a 2x2 gate applied to the pair (a0, a1), zero entries skipped; REAL: every entry is (re, 0)
template <typename T, bool REAL>
__device__ __forceinline__ amp_t<T> tile_row(amp_t<T> ma, amp_t<T> mb, bool has_a, bool has_b, amp_t<T> a0, amp_t<T> a1) {
  using A = amp_t<T>;
  auto prod = [](A m, A x) {
    if constexpr (REAL) {
      A r;
      r.x = m.x * x.x;
      r.y = m.x * x.y;
      return r;
    } else {
      return cmul(m, x);
    }
  };
  if (has_a && has_b) return cadd(prod(ma, a0), prod(mb, a1));
  if (has_a) return prod(ma, a0);
  if (has_b) return prod(mb, a1);
  return czero<A>();
}

// dense 1-qubit gate on pass bit J: four register butterflies.  c[i] = the pass-bit part of element i's tile
// index (wave-uniform), cm = the gate's controls that sit on pass bits.  The common shapes (no control on a pass
// bit, all four entries non-zero) run as straight-line code; every test below is wave-uniform.
template <typename T, int J, bool REAL, bool CHECKED, int NE>
__device__ __forceinline__ void pass_dense_body(const TileGate<T>& g, amp_t<T> (&e)[NE], const uint32_t (&c)[NE], uint32_t cm) {
  using A = amp_t<T>;
  const bool h0 = CHECKED ? (g.nz & 1u) != 0 : true, h1 = CHECKED ? (g.nz & 2u) != 0 : true;
  const bool h2 = CHECKED ? (g.nz & 4u) != 0 : true, h3 = CHECKED ? (g.nz & 8u) != 0 : true;
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    if ((i >> J) & 1) continue;
    const int k = i | (1 << J);
    if constexpr (CHECKED) {
      if ((c[i] & cm) != cm) continue;  // a control on another pass bit is 0 for this pair
      QIP_KEEP_BRANCH();
    }
    const A a0 = e[i], a1 = e[k];
    e[i] = tile_row<T, REAL>(g.m[0], g.m[1], h0, h1, a0, a1);
    e[k] = tile_row<T, REAL>(g.m[2], g.m[3], h2, h3, a0, a1);
    __builtin_amdgcn_sched_barrier(0);  // one butterfly at a time: interleaving them only costs registers
  }
}

template <typename T, int J, int NE>
__device__ __forceinline__ void pass_dense(const TileGate<T>& g, amp_t<T> (&e)[NE], const uint32_t (&c)[NE], uint32_t cm) {
  using A = amp_t<T>;
  const bool is_x = (g.b1 & 2u) != 0, real = (g.b1 & 1u) != 0;
  if (is_x) {
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      if ((i >> J) & 1) continue;
      const int k = i | (1 << J);
      if ((c[i] & cm) != cm) continue;
      QIP_KEEP_BRANCH();
      const A a0 = e[i];
      e[i] = e[k];
      e[k] = a0;
    }
    return;
  }
  if (cm == 0u && g.nz == 15u) {
    if (real) pass_dense_body<T, J, true, false, NE>(g, e, c, cm);
    else pass_dense_body<T, J, false, false, NE>(g, e, c, cm);
  } else {
    if (real) pass_dense_body<T, J, true, true, NE>(g, e, c, cm);
    else pass_dense_body<T, J, false, true, NE>(g, e, c, cm);
  }
}

// Rare shape: a dense gate that still has controls on LANE bits (more controls than a pass holds).  One generic
// form, selected per lane — compactness over speed.
template <typename T, int J, int NE>
__device__ __forceinline__ void pass_dense_lane(const TileGate<T>& g, amp_t<T> (&e)[NE], const uint32_t (&c)[NE], uint32_t cm,
                                                bool lane_ok) {
  using A = amp_t<T>;
  const bool h0 = (g.nz & 1u) != 0, h1 = (g.nz & 2u) != 0, h2 = (g.nz & 4u) != 0, h3 = (g.nz & 8u) != 0;
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    if ((i >> J) & 1) continue;
    const int k = i | (1 << J);
    if ((c[i] & cm) != cm) continue;
    QIP_KEEP_BRANCH();
    const A a0 = e[i], a1 = e[k];
    const A r0 = tile_row<T, false>(g.m[0], g.m[1], h0, h1, a0, a1);
    const A r1 = tile_row<T, false>(g.m[2], g.m[3], h2, h3, a0, a1);
    e[i] = tile_sel(lane_ok, r0, a0);
    e[k] = tile_sel(lane_ok, r1, a1);
  }
}

// e[i] <- f * e[i] for the elements whose pass-bit controls are 1 (HALF >= 0: only elements with pass bit J equal
// to HALF); straight-line when the gate has no control on a pass bit
template <typename T, int J, int HALF, int NE>
__device__ __forceinline__ void pass_scale(amp_t<T> f, amp_t<T> (&e)[NE], const uint32_t (&c)[NE], uint32_t cm) {
  if (cm == 0u) {
#pragma unroll
    for (int i = 0; i < NE; ++i)
      if (HALF < 0 || ((i >> J) & 1) == HALF) e[i] = cmul(f, e[i]);
  } else {
    QIP_KEEP_BRANCH();
#pragma unroll
    for (int i = 0; i < NE; ++i)
      if ((HALF < 0 || ((i >> J) & 1) == HALF) && (c[i] & cm) == cm) {
        QIP_KEEP_BRANCH();
        e[i] = cmul(f, e[i]);
      }
  }
}

// diagonal 1-qubit gate whose target is pass bit J: the factor of element i is m[(i >> J) & 1], known at
// compile time; unit factors (wave-uniform test) leave their four elements untouched.  With lane-bit controls
// the lanes whose controls are 0 multiply by (1, 0) instead: x*1 - y*0 == x for finite amplitudes.
template <typename T, int J, int NE>
__device__ __forceinline__ void pass_diag(const TileGate<T>& g, amp_t<T> (&e)[NE], const uint32_t (&c)[NE], uint32_t cm,
                                          bool lane_ctl, bool lane_ok) {
  using A = amp_t<T>;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    A f = g.m[half];
    if (f.x == (T)1 && f.y == (T)0) continue;
    QIP_KEEP_BRANCH();
    if (lane_ctl) {
      QIP_KEEP_BRANCH();
      f.x = lane_ok ? f.x : (T)1;
      f.y = lane_ok ? f.y : (T)0;
    }
    if (half == 0) pass_scale<T, J, 0, NE>(f, e, c, cm);
    else pass_scale<T, J, 1, NE>(f, e, c, cm);
  }
}

template <typename T, int J0, int J1, int NE>
__device__ __forceinline__ void pass_swap(amp_t<T> (&e)[NE], const uint32_t (&c)[NE], uint32_t cm, bool lane_ctl, bool lane_ok) {
  using A = amp_t<T>;
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    if (!(((i >> J0) & 1) == 1 && ((i >> J1) & 1) == 0)) continue;  // i has (J0,J1) = (1,0); partner (0,1)
    const int k = (i & ~(1 << J0)) | (1 << J1);
    if ((c[i] & cm) != cm) continue;  // controls are never J0/J1, so both elements agree
    QIP_KEEP_BRANCH();
    const A a = e[i], b = e[k];
    if (lane_ctl) {
      QIP_KEEP_BRANCH();
      e[i] = tile_sel(lane_ok, b, a);
      e[k] = tile_sel(lane_ok, a, b);
    } else {
      e[i] = b;
      e[k] = a;
    }
  }
}

// dense 2-qubit gate on pass bits JA (sub-index MSB) and JB: the lane's elements are NE / 4 quads (one per value of the other
// register bits); out[r] = sum_c M[r][c] * in[c] with the fold order of k_gate_kq (all 16 products, columns
// ascending), so circuit-order sweeps stay IEEE-equal to the gate-by-gate path.
template <typename T, int JA, int JB, int NE>
__device__ __forceinline__ void pass_dense2(const amp_t<T>* __restrict__ M, amp_t<T> (&e)[NE], const uint32_t (&c)[NE], uint32_t cm,
                                            bool lane_ctl, bool lane_ok) {
  using A = amp_t<T>;
#pragma unroll
  for (int base = 0; base < NE; ++base) {
    if (((base >> JA) & 1) || ((base >> JB) & 1)) continue;  // one quad per value of the OTHER register bits
    if ((c[base] & cm) != cm) continue;  // controls never sit on JA / JB: one test per quad
    QIP_KEEP_BRANCH();
    A x[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) x[s] = e[base | ((s >> 1) << JA) | ((s & 1) << JB)];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      A acc = cmul(M[r * 4], x[0]);
#pragma unroll
      for (int s = 1; s < 4; ++s) acc = cadd(acc, cmul(M[r * 4 + s], x[s]));
      const int i = base | ((r >> 1) << JA) | ((r & 1) << JB);
      e[i] = lane_ctl ? tile_sel(lane_ok, acc, x[r]) : acc;
    }
  }
}

// dense 3-qubit gate on three register bits: the lane's elements are NE / 8 groups (one with the product's three-bit passes);
// out[r] = sum_c M[r][c] * in[c] with the fold order of k_gate_kq (all 64 products, columns ascending), so circuit-order sweeps
// stay IEEE-equal to the gate-by-gate path.  `c` / `cm`: controls on the OTHER register bits (wide tiles only; a three-bit
// pass has none: its bits are the targets).
template <typename T, int JA, int JB, int JC, int NE>
__device__ __forceinline__ void pass_dense3w(const amp_t<T>* __restrict__ M, amp_t<T> (&e)[NE], const uint32_t (&c)[NE], uint32_t cm,
                                             bool lane_ctl, bool lane_ok) {
  using A = amp_t<T>;
#pragma unroll
  for (int base = 0; base < NE; ++base) {
    if (((base >> JA) & 1) || ((base >> JB) & 1) || ((base >> JC) & 1)) continue;
    if ((c[base] & cm) != cm) continue;
    QIP_KEEP_BRANCH();
    A x[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) x[s] = e[base | (((s >> 2) & 1) << JA) | (((s >> 1) & 1) << JB) | ((s & 1) << JC)];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      A acc = cmul(M[r * 8], x[0]);
#pragma unroll
      for (int s = 1; s < 8; ++s) acc = cadd(acc, cmul(M[r * 8 + s], x[s]));
      const int i = base | (((r >> 2) & 1) << JA) | (((r >> 1) & 1) << JB) | ((r & 1) << JC);
      e[i] = lane_ctl ? tile_sel(lane_ok, acc, x[r]) : acc;
      __builtin_amdgcn_sched_barrier(0);  // one output row at a time: interleaving rows only costs registers
    }
  }
}
template <typename T, int JA, int JB, int JC>
__device__ __forceinline__ void pass_dense3(const amp_t<T>* __restrict__ M, amp_t<T> (&e)[8], bool lane_ctl, bool lane_ok) {
  const uint32_t c0[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  pass_dense3w<T, JA, JB, JC, 8>(M, e, c0, 0u, lane_ctl, lane_ok);
}

// __launch_bounds__(kBlock, 5): five waves per SIMD = the five 32-KiB tiles that fit a CU's LDS.  The sweep is
// latency-bound per wave (scalar gate fetch -> branch -> short VALU body, per gate), so resident blocks are what
// hide it; left alone the compiler spent 170 registers (VGPR + AGPR) on scheduling freedom = 2 blocks per CU.
// (f32: the same bound holds without spills once the SLP vectorizer is off — rustqip_amd/build.py; with it the pass
// packs f32 products into v_pk_* pairs and the kernel needs 180 registers.)
template <typename T, bool NT, bool FOLD = false>
__global__ __launch_bounds__(kTileBlock, 5) void k_tile_passes(amp_t<T>* __restrict__ st, Ins ins, TilePassDesc d,
                                                                              const TileGate<T>* __restrict__ gates,
                                                                              const amp_t<T>* __restrict__ mats,
                                                                              amp_t<T>* __restrict__ out = nullptr,
                                                                              TileStorePerm sp = TileStorePerm(),
                                                                              const TileDiagItem<T>* __restrict__ diag = nullptr) {
  using A = amp_t<T>;
  extern __shared__ __attribute__((aligned(16))) unsigned char tile_raw[];
  A* tile = reinterpret_cast<A*>(tile_raw);
  constexpr int PER = (1 << kTileBits) / kTileBlock;
  static_assert(PER == 8 && kTileLow == 6, "tile index = (u << kTileLaneBits) | (wave << 6) | lane");
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // everything but the lane id is wave-uniform: tile bits 6 .. 6 + kTileWaveBits - 1 = wave id, the top three = u
  uint64_t wbase = tile_block_base(blockIdx.x + (uint64_t)blockIdx.y * gridDim.x, ins, d.p5);
  const uint64_t base = wbase;
#pragma unroll
  for (int j = 0; j < kTileWaveBits; ++j) wbase |= (uint64_t)((wave >> j) & 1u) << d.hpos[j];
  const uint32_t slot_tid = tile_slot<A>(tid);
  const uint32_t lane_off = tile_lane_off(lane, d.p5);
  {
    A x[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const uint64_t ub = wbase | ((uint64_t)(u & 1) << d.hpos[kTileWaveBits]) | ((uint64_t)((u >> 1) & 1) << d.hpos[kTileWaveBits + 1]) |
                          ((uint64_t)((u >> 2) & 1) << d.hpos[kTileWaveBits + 2]);
      x[u] = ldg<NT>(st + ub + lane_off);
    }
#pragma unroll
    for (int u = 0; u < PER; ++u) tile[slot_tid ^ tile_slot<A>((uint32_t)u << kTileLaneBits)] = x[u];
  }
  __syncthreads();
  for (uint32_t pi = 0; pi < d.npasses; ++pi) {
    const TilePass ps = d.pass[pi];
    // this lane's group: the lane id's bits spread over the 8 non-pass tile bits (tb), and the 8 combinations
    // of the pass bits (c[i], wave-uniform); tb and c[i] have no bit in common, so slot(tb | c) = slot(tb) ^ slot(c)
    uint32_t tb = 0;
    const uint32_t lp_lo = (uint32_t)ps.lanepos, lp_hi = (uint32_t)(ps.lanepos >> 32);
#pragma unroll
    for (int k = 0; k < kTileLaneBits; ++k) tb |= ((tid >> k) & 1u) << (((k < 8 ? lp_lo : lp_hi) >> (4 * (k & 7))) & 15u);
    const uint32_t slot_tb = tile_slot<A>(tb);
    uint32_t c[8];
    A e[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      c[i] = ((uint32_t)(i & 1) << ps.pb[0]) | ((uint32_t)((i >> 1) & 1) << ps.pb[1]) |
             ((uint32_t)((i >> 2) & 1) << ps.pb[2]);
      e[i] = tile[slot_tb ^ tile_slot<A>(c[i])];
    }
    const TileGate<T>* gp = gates + ps.first;
    const TileGate<T>* const gend = gp + ps.count;
    for (; gp != gend; ++gp) {
      const TileGate<T> g = *gp;  // wave-uniform (prefetching the next descriptor was measured: no gain)
      if ((base & g.omask) != g.omask) continue;  // an outside control is 0 for this whole tile
      const uint32_t cm_reg = g.cm_reg;
      switch (g.op) {
        case TOP_DIAG_RUN: {
          const TileDiagItem<T>* ip = diag + g.nz;
          const TileDiagItem<T>* const iend = ip + g.b1;
// Measured on MI355X (QFT n = 30 through the interpreter, tools/build_variants.sh A/B inside one GPU call, two rounds each, ms):
//   per-gate code paths 125.3 | runs 90.3 | + in-place products (QIP_DIAG_ASM) 85.1 | + descriptor one step ahead (QIP_DIAG_PREFETCH) 96.4
// the prefetch costs more scalar moves than the load latency it hides (five waves per SIMD already cover it): off.
#ifndef QIP_DIAG_PREFETCH
#define QIP_DIAG_PREFETCH 0
#endif
#ifndef QIP_DIAG_ASM
#define QIP_DIAG_ASM 1
#endif
#if QIP_DIAG_PREFETCH
          TileDiagItem<T> nxt = *ip;  // wave-uniform: scalar loads, one step ahead of their use
#endif
          for (; ip != iend;) {
#if QIP_DIAG_PREFETCH
            const TileDiagItem<T> it = nxt;
            ++ip;
            if (ip != iend) nxt = *ip;
#else
            const TileDiagItem<T> it = *ip;
            ++ip;
#endif
            if ((base & it.omask) != it.oval) continue;
            A f = it.f1;
            const uint32_t selp = it.emask_sel >> 24;
            if (selp != 0u) {
              QIP_KEEP_BRANCH();
              f = tile_sel(((tb >> (selp - 1u)) & 1u) != 0u, it.f1, it.f0);
            }
            if (it.lane_mask != 0u) {
              QIP_KEEP_BRANCH();
              const bool lane_ok = (tb & it.lane_mask) == it.lane_val;
              f.x = lane_ok ? f.x : (T)1;
              f.y = lane_ok ? f.y : (T)0;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
              if ((it.emask_sel >> i) & 1u) {
                QIP_KEEP_BRANCH();
#if QIP_DIAG_ASM
                cscale_inplace(f, e[i]);
#else
                e[i] = cmul(f, e[i]);
#endif
              }
          }
          break;
        }
        case TOP_DIAG_UNIFORM: {
          const A f = ((base >> g.tpos_out) & 1ull) ? g.m[1] : g.m[0];
          if (f.x == (T)1 && f.y == (T)0) break;  // unit entries leave the amplitude untouched
          QIP_KEEP_BRANCH();
          pass_scale<T, 0, -1>(f, e, c, cm_reg);
          break;
        }
        case TOP_DIAG_LANE:
        case TOP_DIAG_LANE_CTL: {
          // per-lane factor: the target bit's entry (a unit entry multiplies exactly), (1, 0) where a lane-bit
          // control is 0
          const bool outside = g.b0 == kTileOutside;
          const bool one = outside ? ((base >> g.tpos_out) & 1ull) != 0 : ((tb >> g.b0) & 1u) != 0;
          A f = tile_sel(one, g.m[1], g.m[0]);
          if (g.op == TOP_DIAG_LANE_CTL) {
            QIP_KEEP_BRANCH();
            const bool lane_ok = (tb & g.cm_lane) == g.cm_lane;
            f.x = lane_ok ? f.x : (T)1;
            f.y = lane_ok ? f.y : (T)0;
          }
          pass_scale<T, 0, -1>(f, e, c, cm_reg);
          break;
        }
        case TOP_DIAG_REG0: pass_diag<T, 0>(g, e, c, cm_reg, g.cm_lane != 0u, (tb & g.cm_lane) == g.cm_lane); break;
        case TOP_DIAG_REG1: pass_diag<T, 1>(g, e, c, cm_reg, g.cm_lane != 0u, (tb & g.cm_lane) == g.cm_lane); break;
        case TOP_DIAG_REG2: pass_diag<T, 2>(g, e, c, cm_reg, g.cm_lane != 0u, (tb & g.cm_lane) == g.cm_lane); break;
        case TOP_DENSE0: pass_dense<T, 0>(g, e, c, cm_reg); break;
        case TOP_DENSE1: pass_dense<T, 1>(g, e, c, cm_reg); break;
        case TOP_DENSE2: pass_dense<T, 2>(g, e, c, cm_reg); break;
        case TOP_DENSE_LANE0: pass_dense_lane<T, 0>(g, e, c, cm_reg, (tb & g.cm_lane) == g.cm_lane); break;
        case TOP_DENSE_LANE1: pass_dense_lane<T, 1>(g, e, c, cm_reg, (tb & g.cm_lane) == g.cm_lane); break;
        case TOP_DENSE_LANE2: pass_dense_lane<T, 2>(g, e, c, cm_reg, (tb & g.cm_lane) == g.cm_lane); break;
#define QIP_D2Q(JA, JB) \
  pass_dense2<T, JA, JB>(mats + 16u * g.nz, e, c, cm_reg, g.cm_lane != 0u, (tb & g.cm_lane) == g.cm_lane)
        case TOP_DENSE2Q_01: QIP_D2Q(0, 1); break;
        case TOP_DENSE2Q_02: QIP_D2Q(0, 2); break;
        case TOP_DENSE2Q_10: QIP_D2Q(1, 0); break;
        case TOP_DENSE2Q_12: QIP_D2Q(1, 2); break;
        case TOP_DENSE2Q_20: QIP_D2Q(2, 0); break;
        case TOP_DENSE2Q_21: QIP_D2Q(2, 1); break;
#undef QIP_D2Q
        case TOP_SWAP_01: pass_swap<T, 0, 1>(e, c, cm_reg, g.cm_lane != 0u, (tb & g.cm_lane) == g.cm_lane); break;
        case TOP_SWAP_02: pass_swap<T, 0, 2>(e, c, cm_reg, g.cm_lane != 0u, (tb & g.cm_lane) == g.cm_lane); break;
        case TOP_SWAP_12: pass_swap<T, 1, 2>(e, c, cm_reg, g.cm_lane != 0u, (tb & g.cm_lane) == g.cm_lane); break;
#define QIP_D3Q(JA, JB, JC) pass_dense3<T, JA, JB, JC>(mats + 16u * g.nz, e, g.cm_lane != 0u, (tb & g.cm_lane) == g.cm_lane)
        case TOP_DENSE3Q_012: QIP_D3Q(0, 1, 2); break;
        case TOP_DENSE3Q_021: QIP_D3Q(0, 2, 1); break;
        case TOP_DENSE3Q_102: QIP_D3Q(1, 0, 2); break;
        case TOP_DENSE3Q_120: QIP_D3Q(1, 2, 0); break;
        case TOP_DENSE3Q_201: QIP_D3Q(2, 0, 1); break;
        case TOP_DENSE3Q_210: QIP_D3Q(2, 1, 0); break;
#undef QIP_D3Q
        default: break;
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) tile[slot_tb ^ tile_slot<A>(c[i])] = e[i];
    __syncthreads();
  }
  {
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const uint64_t ub = wbase | ((uint64_t)(u & 1) << d.hpos[kTileWaveBits]) | ((uint64_t)((u >> 1) & 1) << d.hpos[kTileWaveBits + 1]) |
                          ((uint64_t)((u >> 2) & 1) << d.hpos[kTileWaveBits + 2]);
      if constexpr (FOLD) stg<NT>(out + tile_packed_index(ub, sp) + tile_packed_index(lane_off, sp), tile[slot_tid ^ tile_slot<A>((uint32_t)u << kTileLaneBits)]);
      else stg<NT>(st + ub + lane_off, tile[slot_tid ^ tile_slot<A>((uint32_t)u << kTileLaneBits)]);
    }
  }
}

// ---- dense 4- / 5-qubit gate on the matrix cores through an LDS-resident tile ----------------------------------------------
// k_gate_kq_mfma reads its operands where they lie: 16 groups x 4 sub-indices per load instruction = four 256-B runs, 65 %
// of peak for k = 4 and k = 5 whatever the target bits.  Here a block stages the tile the one-op sweeps use — index bits 0..5
// + the gate's targets above them + free positions from 11 upwards, whole (split) rows on both global sides — and the wave
// takes its matrix-core operands from LDS: same lane mapping (lane = (group j, q), c~ = 4 m + q over the targets in ascending
// position order), same host-built A fragments (16 values per lane for k = 4, 64 for k = 5), same fma chains (1e-12 bar),
// results written back over the operands in LDS, then the tile is stored row by row.  The tile holds 2^(11-k) groups: eight
// items of 16 for k = 4 (two per wave), four for k = 5 (one per wave).  Row blocks go through the matrix pipe two at a time
// (two independent accumulator chains over the same B operands).
// k = 5 (r6, Complex<f64>; profiles/r06_dense_k5.md): 8 flop/B — at the HBM rate the f64 matrix pipe is 67 % busy (64 matrix
// instructions = 1.7 us per tile and SIMD against 2.5 us of HBM time per tile and CU), and with 128 registers of A per lane only
// two blocks fit a CU.  One tile per block ran 8.4 ms at n = 30 (load, barrier, 1.7 us of matrix pipe, barrier, store: nothing
// overlaps; the direct kernel: 6.6).  LOOP: a block walks `pipe` consecutive tiles with the NEXT tile's rows in flight (in
// registers) while the current one is in the matrix pipe: 8 tiles 6.0 ms, 32 5.7, 256 5.6 = 76 % of 8 TB/s (direct: 63 - 65 %).
struct TileMfmaDesc {
  uint32_t hpos[kTileHigh];  // amplitude-index positions of tile bits 6..10 (ascending)
  uint32_t tb[5];            // tile-bit index of the targets, ascending
  uint32_t nb[kTileBits - 4];  // the other tile bits, ascending: nb[0..3] = group, the rest = item
  uint32_t p5;                 // amplitude-index position of tile bit 5 (see tile_block_base)
};

template <typename T, int K, bool NT, bool LOOP, int NP>  // NP = 0: the four-product real form (k = 4); 3 / 1: mfma3_item (k = 5)
__global__ __launch_bounds__(kTileBlock, K == 4 ? 5 : 2) void k_gate_tile_mfma(amp_t<T>* __restrict__ st, Ins ins, TileMfmaDesc d,
                                                                              const T* __restrict__ afrag, uint32_t pipe) {
  const uint32_t PIPE = LOOP ? pipe : 1u;  // tiles per block
  using A = amp_t<T>;
  using V4 = typename Acc4<T>::type;
  constexpr int S = 1 << K, TT = S / 8, KS = S / 2, NA = S / 4;
  constexpr int IB = kTileBits - K - 4;          // item bits of the tile
  constexpr int IPW = (1 << IB) / (kTileBlock / 64);  // items per wave
  static_assert(IPW >= 1 && TT % 2 == 0, "tile shape");
  extern __shared__ __attribute__((aligned(16))) unsigned char tile_raw[];
  A* tile = reinterpret_cast<A*>(tile_raw);
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint64_t first = (blockIdx.x + (uint64_t)blockIdx.y * gridDim.x) * PIPE;
  auto base_of = [&](uint64_t t) {
    uint64_t w = tile_block_base(t, ins, d.p5);
#pragma unroll
    for (int j = 0; j < kTileWaveBits; ++j) w |= (uint64_t)((wave >> j) & 1u) << d.hpos[j];
    return w;
  };
  auto row_of = [&](int u) {
    return ((uint64_t)(u & 1) << d.hpos[kTileWaveBits]) | ((uint64_t)((u >> 1) & 1) << d.hpos[kTileWaveBits + 1]) | ((uint64_t)((u >> 2) & 1) << d.hpos[kTileWaveBits + 2]);
  };
  const uint32_t slot_tid = tile_slot<A>(tid);
  const uint32_t lane_off = tile_lane_off(lane, d.p5);
  uint64_t wbase = base_of(first);
  A r[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) r[u] = ldg<NT>(st + (wbase | row_of(u)) + lane_off);
  // the gate's fragments while the rows are on their way
  constexpr int NPA = NP ? NP : 1, RB3 = NP ? S / 16 : 1, KS3 = NP ? S / 4 : 1;
  T a[NP ? 1 : TT][NP ? 1 : KS];
  T a3[RB3][NPA][KS3];
  if constexpr (NP == 0) {
#pragma unroll
    for (int rb = 0; rb < TT; ++rb)
#pragma unroll
      for (int s = 0; s < KS; ++s) a[rb][s] = afrag[(rb * KS + s) * 64 + lane];
  } else {
#pragma unroll
    for (int rb = 0; rb < RB3; ++rb)
#pragma unroll
      for (int pt = 0; pt < NPA; ++pt)
#pragma unroll
        for (int s = 0; s < KS3; ++s) a3[rb][pt][s] = afrag[((rb * NPA + pt) * KS3 + s) * 64 + lane];
  }
  // lane = (group j, q): j fills the four lowest non-target tile bits, q the two lowest targets
  const uint32_t j = lane & 15u, q = lane >> 4;
  uint32_t t_lane = ((q & 1u) << d.tb[0]) | ((q >> 1) << d.tb[1]);
#pragma unroll
  for (int b = 0; b < 4; ++b) t_lane |= ((j >> b) & 1u) << d.nb[b];
  const uint32_t slot_lane = tile_slot<A>(t_lane);
  // a block moves PIPE consecutive tiles: the rows of tile p + 1 are in flight while tile p goes through the matrix pipe
#pragma unroll 1
  for (uint32_t p = 0; p < PIPE; ++p) {
#pragma unroll
    for (int u = 0; u < 8; ++u) tile[slot_tid ^ tile_slot<A>((uint32_t)u << kTileLaneBits)] = r[u];
    __syncthreads();
    const uint64_t wcur = wbase;
    if (p + 1 < PIPE) {
      wbase = base_of(first + (uint64_t)p + 1);
#pragma unroll
      for (int u = 0; u < 8; ++u) r[u] = ldg<NT>(st + (wbase | row_of(u)) + lane_off);
    }
#pragma unroll
    for (int it = 0; it < IPW; ++it) {
      const uint32_t item = wave * (uint32_t)IPW + (uint32_t)it;  // (wave-uniform)
      uint32_t t_item = 0;
#pragma unroll
      for (int b = 0; b < IB; ++b) t_item |= ((item >> b) & 1u) << d.nb[4 + b];
      uint32_t slot[NA];
      A x[NA];
#pragma unroll
      for (int m = 0; m < NA; ++m) {
        uint32_t t_m = t_item;
#pragma unroll
        for (int b = 0; b < K - 2; ++b) t_m |= (((uint32_t)m >> b) & 1u) << d.tb[2 + b];
        slot[m] = slot_lane ^ tile_slot<A>(t_m);
        x[m] = tile[slot[m]];
      }
      if constexpr (NP != 0) {
        A y[NA];
        mfma3_item<T, K, NPA>(a3, x, y);
#pragma unroll
        for (int m = 0; m < NA; ++m) tile[slot[m]] = y[m];  // every amplitude of the tile belongs to exactly one lane: in place
      } else {
#pragma unroll
        for (int rb = 0; rb < TT; rb += 2) {
          V4 acc0 = {(T)0, (T)0, (T)0, (T)0}, acc1 = {(T)0, (T)0, (T)0, (T)0};
#pragma unroll
          for (int s = 0; s < KS; ++s) {
            const T b = (s & 1) ? x[s >> 1].y : x[s >> 1].x;
            acc0 = mfma16(a[rb][s], b, acc0);
            acc1 = mfma16(a[rb + 1][s], b, acc1);
          }
          A y0, y1, y2, y3;
          y0.x = acc0[0];
          y0.y = acc0[1];
          y1.x = acc0[2];
          y1.y = acc0[3];
          y2.x = acc1[0];
          y2.y = acc1[1];
          y3.x = acc1[2];
          y3.y = acc1[3];
          tile[slot[2 * rb]] = y0;      // every amplitude of the tile belongs to exactly one lane: in place
          tile[slot[2 * rb + 1]] = y1;
          tile[slot[2 * rb + 2]] = y2;
          tile[slot[2 * rb + 3]] = y3;
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 8; ++u) stg<NT>(st + (wcur | row_of(u)) + lane_off, tile[slot_tid ^ tile_slot<A>((uint32_t)u << kTileLaneBits)]);
    if (p + 1 < PIPE) __syncthreads();
  }
}

// ---- literal fallback: one output row per lane, out of place ------------------------------
// The gather formulation of the reference, variant by variant (matrix_ops.rs:62-94,
// ops.rs:100-156, qubit_iterators.rs).  Correct for every descriptor the reference accepts,
// including window offsets and accumulate; used for SparseMatrix, large dense matrices,
// descriptors with repeated indices, and the host-pointer twin of apply_op.
struct GatherDesc {
  uint32_t n;
  uint32_t k_all;       // indices of the outer op
  uint32_t n_control;   // flattened controls (0 when not a Control)
  uint32_t n_op;        // indices the innermost iterator is built with
  int32_t inner_kind;   // QIP_OP_MATRIX / SPARSE / SWAP
  int32_t accumulate;
  uint64_t in_len, out_len, in_off, out_off;
  uint32_t pos[kMaxIns];  // pos[j] = n-1-indices[j]
};

__device__ __forceinline__ uint64_t g_full_to_sub(const GatherDesc& d, uint64_t full) {
  uint64_t acc = 0;
  for (uint32_t j = 0; j < d.k_all; ++j) acc |= ((full >> d.pos[j]) & 1ull) << (d.k_all - 1 - j);
  return acc;
}
__device__ __forceinline__ uint64_t g_sub_to_full(const GatherDesc& d, uint64_t sub, uint64_t base) {
  uint64_t acc = base;
  for (uint32_t j = 0; j < d.k_all; ++j) {
    const uint64_t bit = (sub >> (d.k_all - 1 - j)) & 1ull;
    acc = (acc & ~(1ull << d.pos[j])) | (bit << d.pos[j]);
  }
  return acc;
}

template <typename T>
__device__ __forceinline__ amp_t<T> g_term(const GatherDesc& d, uint64_t row, uint64_t col,
                                           amp_t<T> val, const amp_t<T>* __restrict__ in) {
  using A = amp_t<T>;
  const uint64_t colbits = g_sub_to_full(d, col, row);
  if (colbits < d.in_off) return czero<A>();
  const uint64_t vecrow = colbits - d.in_off;
  if (vecrow >= d.in_len) return czero<A>();
  return cmul(val, in[vecrow]);
}

// ---- dense k = 5..10 on a state too small for the matrix-core kernels (fewer than 16 groups: n < k + controls + 4) ----------
// The reference's own bench shape (qip/benches/state_bench.rs:118-139: n = 8, one dense 8-qubit gate) is ONE 256 x 256 complex
// matrix-vector product.  The literal kernel runs it as 256 lanes x 256 sequential terms with the reference's index loops per
// term (172 us); the matrix-core kernels need 16 groups per wave.  Here a block of 256 lanes owns 16 rows of one group: lane
// (chunk, r) folds the columns [chunk * S/16, (chunk + 1) * S/16) of row 16 * rb + r in increasing column order — the group's
// amplitudes staged in LDS, the matrix read TRANSPOSED (mt[c * S + row]: 16 consecutive rows = 256 contiguous bytes) — and lane
// (0, r) adds the 16 partial sums in chunk order.  Out of place (in -> out); rows outside the control subspace are copied by
// the launcher first.  Products and sums are unfused; the order of the additions differs from the reference's single fold:
// the 1e-12 bar of dense k >= 3 gates, not bit equality (option force_generic keeps the literal fold).
struct DenseSmallDesc {
  uint32_t k;
  uint32_t tpos[10];  // tpos[b] = index position of sub-index bit b (bit 0 = the LAST op index, matrix_ops.rs:12-21)
};
template <typename T>
__global__ __launch_bounds__(kBlock) void k_dense_small(const amp_t<T>* __restrict__ in, amp_t<T>* __restrict__ out, Ins ins,
                                                        DenseSmallDesc d, const amp_t<T>* __restrict__ mt) {
  using A = amp_t<T>;
  extern __shared__ __attribute__((aligned(16))) unsigned char dense_small_lds[];
  A* x = reinterpret_cast<A*>(dense_small_lds);  // [S]
  const uint32_t S = 1u << d.k;
  A* partial = x + S;                            // [16][16]
  const uint32_t tid = threadIdx.x, r = tid & 15u, chunk = tid >> 4;
  const uint64_t base = insert_bits<-1>((uint64_t)blockIdx.y, ins);
  auto spread = [&](uint32_t c) {
    uint64_t o = 0;
    for (uint32_t b = 0; b < d.k; ++b) o |= (uint64_t)((c >> b) & 1u) << d.tpos[b];
    return o;
  };
  for (uint32_t c = tid; c < S; c += kBlock) x[c] = in[base | spread(c)];
  __syncthreads();
  const uint32_t row = 16u * blockIdx.x + r, per = S >> 4, c0 = chunk * per;
  A acc = czero<A>();
  const A* col = mt + (size_t)c0 * S + row;
#pragma unroll 4
  for (uint32_t j = 0; j < per; ++j) acc = cadd(acc, cmul(col[(size_t)j * S], x[c0 + j]));
  partial[chunk * 16u + r] = acc;
  __syncthreads();
  if (chunk == 0) {
    A tot = partial[r];
    for (uint32_t ch = 1; ch < 16u; ++ch) tot = cadd(tot, partial[ch * 16u + r]);
    out[base | spread(row)] = tot;
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_gather_generic(const amp_t<T>* __restrict__ in,
                                                           amp_t<T>* __restrict__ out, GatherDesc d,
                                                           const amp_t<T>* __restrict__ dense,
                                                           const uint64_t* __restrict__ rowptr,
                                                           const uint64_t* __restrict__ cols,
                                                           const amp_t<T>* __restrict__ vals) {
  using A = amp_t<T>;
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t r = (uint64_t)blockIdx.x * kBlock + threadIdx.x; r < d.out_len; r += stride) {
    const uint64_t row = d.out_off + r;
    const uint64_t matrow = g_full_to_sub(d, row);
    A acc = czero<A>();
    A one;
    one.x = 1;
    one.y = 0;
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
      acc = cadd(acc, g_term<T>(d, row, matrow, one, in));
    } else if (d.inner_kind == 0) {  // MATRIX
      const uint64_t side = 1ull << d.n_op;
      const A* rowdata = dense + irow * side;
      for (uint64_t c = 0; c < side; ++c) {
        const A v = rowdata[c];
        if (!(v.x == (T)0 && v.y == (T)0)) acc = cadd(acc, g_term<T>(d, row, c + shift, v, in));
      }
    } else if (d.inner_kind == 1) {  // SPARSE
      for (uint64_t p = rowptr[irow]; p < rowptr[irow + 1]; ++p)
        acc = cadd(acc, g_term<T>(d, row, cols[p] + shift, vals[p], in));
    } else {  // SWAP
      const uint32_t half_n = d.n_op >> 1;
      const uint64_t lower_mask = ~(~0ull << half_n);
      const uint64_t col = ((irow & lower_mask) << half_n) + (irow >> half_n);
      acc = cadd(acc, g_term<T>(d, row, col + shift, one, in));
    }
    if (d.accumulate) {
      out[r] = cadd(out[r], acc);
    } else {
      out[r] = acc;
    }
  }
}

// out[i] += in[i]  (accumulate leg of the host twin when a fast in-place kernel produced `in`)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_add_into(amp_t<T>* __restrict__ out,
                                                     const amp_t<T>* __restrict__ in, uint64_t len) {
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < len; i += stride)
    out[i] = cadd(out[i], in[i]);
}

// ---- measurement (qip/src/state_ops/measurement_ops.rs) -------------------------------------

__device__ __forceinline__ double block_reduce_sum(double v, double* smem) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) smem[wv] = v;
  __syncthreads();
  double t = 0;
  if (threadIdx.x == 0)
    for (int i = 0; i < kBlock / 64; ++i) t += smem[i];
  __syncthreads();  // smem may be reused by the next call
  return t;  // valid in thread 0
}

// partial[m * gridDim.x + blockIdx.x] = sum of |amp|^2 over this block's slice of the
// sub-space whose measured bits read m (measure_prob_fn, measurement_ops.rs:65-112).
// `ins` opens the measured bit positions; mbits[m] are precomputed by the host? no:
// bit i of m goes to position mpos[i] (LSB-first, :72-79).
struct MeasDesc {
  uint32_t k;
  uint32_t mpos[kMaxIns];
};

__device__ __forceinline__ uint64_t meas_template(const MeasDesc& md, uint64_t m) {
  uint64_t t = 0;
  for (uint32_t i = 0; i < md.k; ++i) t |= ((m >> i) & 1ull) << md.mpos[i];
  return t;
}

// |amp|^2 in the state's own precision (measurement_ops.rs:65-112 sums P values), widened for the accumulation.  A packed
// f32 element holds two adjacent amplitudes (index bit 0 inside the element): `lo` / `hi` are theirs.
__device__ __forceinline__ double prob_of(amp_t<double> x) { return x.x * x.x + x.y * x.y; }
__device__ __forceinline__ double prob_of(amp_t<float> x) { return (double)(x.x * x.x + x.y * x.y); }
__device__ __forceinline__ double prob_lo(f32x4 x) { return (double)(x.x * x.x + x.y * x.y); }
__device__ __forceinline__ double prob_hi(f32x4 x) { return (double)(x.z * x.z + x.w * x.w); }
__device__ __forceinline__ double prob_of(f32x4 x) { return prob_lo(x) + prob_hi(x); }  // both amplitudes of the element

template <typename T>
__global__ __launch_bounds__(kBlock) void k_measure_probs(const amp_t<T>* __restrict__ st,
                                                          uint64_t count, Ins ins, MeasDesc md,
                                                          uint64_t m_first,
                                                          double* __restrict__ partial) {
  __shared__ double smem[kBlock / 64];
  const uint64_t m = m_first + blockIdx.y;
  const uint64_t templ = meas_template(md, m);
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  double s = 0;
  for (uint64_t w = (uint64_t)blockIdx.x * kBlock + threadIdx.x; w < count; w += stride) {
    const amp_t<T> x = st[insert_bits<-1>(w, ins) | templ];
    s += (double)(x.x * x.x + x.y * x.y);
  }
  const double t = block_reduce_sum(s, smem);
  if (threadIdx.x == 0) partial[(uint64_t)blockIdx.y * gridDim.x + blockIdx.x] = t;
}

// few outcomes (k <= 4): ONE fully coalesced pass over the vector; every lane keeps 2^K running sums
// (select-by-compare, no dynamic register indexing) and blocks write partial[blockIdx.x * 2^K + m].
// r4: four independent 16-byte loads in flight per lane, 32 KiB apart (`work_index`'s spacing: 6.2 vs 5.3 TB/s for the gate
// kernels; a lone load per iteration left this pass at 74 %), and Complex<f32> states read as 16-byte elements of two
// amplitudes (E = f32x4, positions in units of elements: bit 0 of the amplitude index is the half of the element and is
// measured through `bit0` = its outcome bit, or -1) — an 8-byte access per lane runs at 0.54 - 0.70x the 16-byte rate.
template <typename T, int K, typename E = amp_t<T>>
__global__ __launch_bounds__(kBlock) void k_measure_probs_small(const E* __restrict__ st,
                                                                uint64_t nelems, MeasDesc md, int bit0,
                                                                double* __restrict__ partial) {
  constexpr int M = 1 << K;
  constexpr bool PACKED = !SameT<E, amp_t<T>>::v;
  __shared__ double smem[kBlock / 64];
  double acc[M];
#pragma unroll
  for (int m = 0; m < M; ++m) acc[m] = 0.0;
  auto take = [&](uint64_t i, E x) {  // `i`: element index
    uint32_t mine = 0;
#pragma unroll
    for (int b = 0; b < K; ++b)
      if (!(PACKED && b == bit0)) mine |= (uint32_t)((i >> md.mpos[b]) & 1ull) << b;
    if constexpr (PACKED) {
      const double p0 = prob_lo(x), p1 = prob_hi(x);
      const uint32_t hi_bit = bit0 >= 0 ? 1u << bit0 : 0u;
#pragma unroll
      for (int m = 0; m < M; ++m) acc[m] += (mine == (uint32_t)m ? p0 : 0.0) + ((mine | hi_bit) == (uint32_t)m ? p1 : 0.0);
    } else {
      const double p = prob_of(x);
#pragma unroll
      for (int m = 0; m < M; ++m) acc[m] += (mine == (uint32_t)m) ? p : 0.0;
    }
  };
  constexpr int U = 4;
  constexpr uint64_t kSpan = (uint64_t)U << kStrideShift;  // elements one round of a block covers per 256-lane row set
  const uint64_t rounds = nelems / kSpan;                  // whole spans: blocks take them round-robin, 8 rows of 256 each
  // span r = elements [r * kSpan, (r + 1) * kSpan): row q (0..7) of access u sits at r * kSpan + (u << kStrideShift) + q * 256
  for (uint64_t r = blockIdx.x; r < rounds; r += gridDim.x) {
#pragma unroll 1
    for (uint32_t qrow = 0; qrow < (1u << (kStrideShift - 8)); ++qrow) {
      const uint64_t i0 = r * kSpan + (uint64_t)qrow * kBlock + threadIdx.x;
      E x[U];
#pragma unroll
      for (int u = 0; u < U; ++u) x[u] = __builtin_nontemporal_load(st + i0 + ((uint64_t)u << kStrideShift));
#pragma unroll
      for (int u = 0; u < U; ++u) take(i0 + ((uint64_t)u << kStrideShift), x[u]);
    }
  }
  for (uint64_t i = rounds * kSpan + (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < nelems; i += (uint64_t)gridDim.x * kBlock)
    take(i, st[i]);
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const double t = block_reduce_sum(acc[m], smem);
    if (threadIdx.x == 0) partial[(uint64_t)blockIdx.x * M + m] = t;
  }
}

// 5 <= k <= ~23 outcome bits without any atomic.  The measured positions split three ways:
//   lane bits   positions < 8: a function of the lane id alone (kl of them);
//   grid bits   the block only visits indices that read `mg` there (`ins` opens them): blockIdx.x / gx = mg;
//   step bits   up to three of the measured positions >= 8 (the lowest ones, taken off the grid when the grid would
//               otherwise have far more blocks than the chip needs): the lane walks their 2^KI values in a fully
//               unrolled inner loop, one running sum per value (static register index, independent loads).
// So a lane accumulates 2^KI running sums over fully coalesced 4-KiB block rows, and at the end the 256 lane sums of
// each are folded by lane outcome through wave shuffles and LDS.  Outcome index o = (mg << (KI + kl)) | (c << kl) | lane
// outcome; partial[bx * nout + o] (a block's results are contiguous); k_sum_partials adds the gx blocks of an outcome.
struct MeasGridDesc {
  uint32_t kg, kl;
  uint32_t gpos[kMaxIns];   // measured positions on the grid, in outcome-bit order of the grid part
  uint32_t lpos[8];         // measured positions < 8
  uint32_t spos[3];         // step positions (bit i of c)
};
// (E = f32x4: a Complex<f32> state read as 16-byte elements of two amplitudes when index bit 0 is not measured — positions in
// units of elements, an element contributes both of its amplitudes)
// B0 (packed elements only, r4): index bit 0 IS measured — the two halves of an element go to two outcomes; the outcome index gets
// one more bit below the lane outcome: o' = 2 o + half.
template <typename T, int KI, typename E = amp_t<T>, bool B0 = false>
__global__ __launch_bounds__(kBlock) void k_measure_probs_grid(const E* __restrict__ st, uint64_t count, Ins ins,
                                                              MeasGridDesc md, uint32_t gx, uint64_t nout,
                                                              double* __restrict__ partial) {
  constexpr int NC = 1 << KI;
  constexpr int H = B0 ? 2 : 1;
  __shared__ double lane_sum[kBlock];
  // 1-D grid of (outcomes on the grid) x gx blocks: HIP caps gridDim.y at 65535 and kg may reach 20
  const uint64_t mg = blockIdx.x / gx;
  const uint32_t bx = blockIdx.x % gx;
  uint64_t templ = 0;
  for (uint32_t i = 0; i < md.kg; ++i) templ |= ((mg >> i) & 1ull) << md.gpos[i];
  uint64_t coff[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    uint64_t o = 0;
#pragma unroll
    for (int i = 0; i < KI; ++i)
      if ((c >> i) & 1) o |= 1ull << md.spos[i];
    coff[c] = o | templ;
  }
  // `count` (a power of two) work items per (mg, c); gx blocks stride over them; >= 4 independent rows in flight
  constexpr int UN = NC >= 4 ? 1 : 4 / NC;
  const uint64_t stride = (uint64_t)gx * kBlock;
  double acc[H][NC][UN];
#pragma unroll
  for (int h = 0; h < H; ++h)
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int u = 0; u < UN; ++u) acc[h][c][u] = 0.0;
  auto add = [&](int c, int u, E a) {
    if constexpr (B0) {
      acc[0][c][u] += prob_lo(a);
      acc[H - 1][c][u] += prob_hi(a);
    } else {
      acc[0][c][u] += prob_of(a);
    }
  };
  uint64_t w = (uint64_t)bx * kBlock + threadIdx.x;
  for (; w + (UN - 1) * stride < count; w += UN * stride) {
    E a[NC][UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const uint64_t base = insert_bits<-1>(w + u * stride, ins);
#pragma unroll
      for (int c = 0; c < NC; ++c) a[c][u] = __builtin_nontemporal_load(st + (base | coff[c]));
    }
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int u = 0; u < UN; ++u) add(c, u, a[c][u]);
  }
  for (; w < count; w += stride) {
    const uint64_t base = insert_bits<-1>(w, ins);
#pragma unroll
    for (int c = 0; c < NC; ++c) add(c, 0, __builtin_nontemporal_load(st + (base | coff[c])));
  }
  // fold the 256 lane sums by lane outcome: add across every lane-id bit that is NOT measured (wave shuffles for
  // bits 0..5, LDS for the two wave-id bits); the lanes whose unmeasured bits are all zero then hold the totals
  uint32_t lmask = 0;
  for (uint32_t i = 0; i < md.kl; ++i) lmask |= 1u << md.lpos[i];
  uint32_t lo = 0;  // bit i of the lane outcome <-> lane-id bit lpos[i]
  for (uint32_t i = 0; i < md.kl; ++i) lo |= ((threadIdx.x >> md.lpos[i]) & 1u) << i;
  const bool writer = (threadIdx.x & ~lmask & 255u) == 0u;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
#pragma unroll
    for (int h = 0; h < H; ++h) {
      double v = 0;
#pragma unroll
      for (int u = 0; u < UN; ++u) v += acc[h][c][u];
#pragma unroll
      for (int b = 0; b < 6; ++b)
        if (!((lmask >> b) & 1u)) v += __shfl_xor(v, 1 << b, 64);  // wave-uniform condition
      __syncthreads();  // lane_sum is reused per (c, h)
      lane_sum[threadIdx.x] = v;
      __syncthreads();
      if (!((lmask >> 6) & 1u)) v += lane_sum[threadIdx.x ^ 64u];
      __syncthreads();
      lane_sum[threadIdx.x] = v;
      __syncthreads();
      if (!((lmask >> 7) & 1u)) v += lane_sum[threadIdx.x ^ 128u];
      if (writer) partial[(uint64_t)bx * nout + (((((mg << KI) | (uint64_t)c) << md.kl) | lo) * H + h)] = v;
    }
  }
}

// out[o] = sum of the gx per-block partial sums of outcome o (partial[b * nout + o]); one wave per outcome
// (a template only so that the header can be included by several translation units)
template <int WAVE = 64>
__global__ __launch_bounds__(kBlock) void k_sum_partials(const double* __restrict__ partial, uint32_t gx, uint64_t nout,
                                                        double* __restrict__ out) {
  static_assert(WAVE == 64, "gfx950 wavefronts");
  const uint64_t o = (uint64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (o >= nout) return;
  double t = 0;
  for (uint32_t b = threadIdx.x & 63u; b < gx; b += 64) t += partial[(uint64_t)b * nout + o];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
  if ((threadIdx.x & 63u) == 0) out[o] = t;
}

// probabilities of many outcomes: every amplitude adds |amp|^2 to out[its outcome]
template <typename T>
__global__ __launch_bounds__(kBlock) void k_measure_probs_scatter(const amp_t<T>* __restrict__ st,
                                                                  uint64_t namps, MeasDesc md, uint64_t offset,
                                                                  double* __restrict__ out) {
  // `offset`: st[0] is amplitude `offset` of the full vector (the input_offset window of measurement_ops.rs:17-19)
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t w = (uint64_t)blockIdx.x * kBlock + threadIdx.x; w < namps; w += stride) {
    const amp_t<T> x = st[w];
    if (x.x == (T)0 && x.y == (T)0) continue;  // measurement_ops.rs:98-99
    const uint64_t i = w + offset;
    uint64_t m = 0;
    for (uint32_t b = 0; b < md.k; ++b) m |= ((i >> md.mpos[b]) & 1ull) << b;
    atomicAdd(&out[m], (double)(x.x * x.x + x.y * x.y));
  }
}

// partial[b] = sum |amp|^2 over the b-th contiguous chunk of `chunk` amplitudes (elements, for a packed f32 state).
// r4: four independent 16-byte loads per lane 32 KiB apart where the chunk allows it, packed elements for Complex<f32>.
template <typename T, typename E = amp_t<T>>
__global__ __launch_bounds__(kBlock) void k_chunk_norms(const E* __restrict__ st,
                                                        uint64_t nelems, uint64_t chunk,
                                                        double* __restrict__ partial) {
  constexpr bool PACKED = !SameT<E, amp_t<T>>::v;
  __shared__ double smem[kBlock / 64];
  const uint64_t lo = (uint64_t)blockIdx.x * chunk;
  const uint64_t hi = lo + chunk < nelems ? lo + chunk : nelems;
  auto prob = [](E x) {
    if constexpr (PACKED) return prob_lo(x) + prob_hi(x);
    else return prob_of(x);
  };
  double s = 0;
  constexpr int U = 4;
  constexpr uint64_t kSpan = (uint64_t)U << kStrideShift;
  uint64_t i = lo;
  for (; i + kSpan <= hi; i += kSpan) {
#pragma unroll 1
    for (uint32_t qrow = 0; qrow < (1u << (kStrideShift - 8)); ++qrow) {
      const uint64_t i0 = i + (uint64_t)qrow * kBlock + threadIdx.x;
      E x[U];
#pragma unroll
      for (int u = 0; u < U; ++u) x[u] = __builtin_nontemporal_load(st + i0 + ((uint64_t)u << kStrideShift));
#pragma unroll
      for (int u = 0; u < U; ++u) s += prob(x[u]);
    }
  }
  for (i += threadIdx.x; i < hi; i += kBlock) s += prob(st[i]);
  const double t = block_reduce_sum(s, smem);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// Two states compared amplitude by amplitude over the whole vector (qip_hip_state_max_abs_diff: how far a state is from a
// reference copy): partial[2 b] = max |a_i - b_i| over block b's share, partial[2 b + 1] = number of amplitudes whose
// components are not IEEE-equal (NaNs count as different).  Grid-stride, one coalesced pass over both vectors.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_max_abs_diff(const amp_t<T>* __restrict__ a, const amp_t<T>* __restrict__ b,
                                                         uint64_t namps, double* __restrict__ partial) {
  __shared__ double smem[kBlock / 64];
  double worst = 0, differ = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < namps; i += (uint64_t)gridDim.x * kBlock) {
    const amp_t<T> x = a[i], y = b[i];
    if (!(x.x == y.x && x.y == y.y)) {
      differ += 1;
      const double dx = (double)x.x - (double)y.x, dy = (double)x.y - (double)y.y;
      const double d = sqrt(dx * dx + dy * dy);
      worst = (d > worst || d != d) ? d : worst;  // a NaN distance sticks
    }
  }
  const double cnt = block_reduce_sum(differ, smem);
  __syncthreads();
  // max over the block: wave shuffles, then the four wave maxima through LDS
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double o = __shfl_down(worst, off, 64);
    worst = (o > worst || o != o) ? o : worst;
  }
  if ((threadIdx.x & 63u) == 0) smem[threadIdx.x >> 6] = worst;
  __syncthreads();
  if (threadIdx.x == 0) {
    double m = smem[0];
    for (int w = 1; w < kBlock / 64; ++w) m = (smem[w] > m || smem[w] != smem[w]) ? smem[w] : m;
    partial[2 * blockIdx.x] = m;
    partial[2 * blockIdx.x + 1] = cnt;
  }
}

// Amplitudes picked by an explicit index list (qip_hip_state_download_indices: the parity checks' download of a sub-cube whose
// index bits are scattered — a logical window of a relabelled or sharded state).  out[i] = st[idx[i]]; the indices of a sub-cube come
// in runs, so neighbouring lanes mostly read neighbouring amplitudes.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_gather_indices(const amp_t<T>* __restrict__ st, const uint64_t* __restrict__ idx,
                                                           uint64_t count, amp_t<T>* __restrict__ out) {
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < count; i += (uint64_t)gridDim.x * kBlock) out[i] = st[idx[i]];
}

// soft_measure's sequential scan (measurement_ops.rs:167-173) inside ONE chunk: find the first index at
// which r - sum_{j<=i} |amp_j|^2 <= 0.  One block, two levels: every lane sums its contiguous segment (in double), lane 0
// walks the 256 segment sums to a segment that may bring r to <= 0 (with a rounding margin); the block then splits THAT segment
// into 256 sub-segments the same way, and only the sub-segment that may cross is replayed element by element, in the
// precision and order of the reference, by the lane that owns it (r4: the round-3 form replayed a whole segment — 1024 dependent
// loads of one lane at n = 30, most of soft_measure's time beyond the norm pass).  A (sub-)segment that came close without
// crossing hands its exact remainder on.  *r_io / *found live in shared memory, set by the caller (found = ~0).
__device__ __forceinline__ int walk_sums(const double* v, int start, int cnt, double& r) {
  for (int t = start; t < cnt; ++t) {
    if (r - v[t] <= 1e-9 * (1.0 + v[t])) return t;  // may cross (or come within rounding of it): look inside
    r -= v[t];
  }
  return -1;
}

template <typename T>
__device__ void find_crossing_block(const amp_t<T>* __restrict__ st, uint64_t lo, uint64_t len, double* r_io,
                                    unsigned long long* found) {
  __shared__ double seg_sum[kBlock], sub_sum[kBlock];
  __shared__ int owner, owner2;
  const uint32_t tid = threadIdx.x;
  const uint64_t end = lo + len;
  auto sum_range = [&](uint64_t a, uint64_t b) {
    double acc = 0;  // double for both dtypes: the skip margin is far tighter than f32 summation error
    for (uint64_t i = a; i < b; ++i) {
      const amp_t<T> x = st[i];
      acc += (double)(x.x * x.x + x.y * x.y);
    }
    return acc;
  };
  const uint64_t seg = (len + kBlock - 1) / kBlock;
  // the 256 segment sums, a wave per segment in turn: 64 lanes read consecutive amplitudes (one lane per segment would touch 64
  // different lines per load instruction — measured: the larger part of the crossing search's time)
  for (uint32_t sg = tid >> 6; sg < (uint32_t)kBlock; sg += kBlock / 64) {
    const uint64_t a = lo + (uint64_t)sg * seg < end ? lo + (uint64_t)sg * seg : end;
    const uint64_t b = a + seg < end ? a + seg : end;
    double acc = 0;
    for (uint64_t i = a + (tid & 63u); i < b; i += 64) {
      const amp_t<T> x = st[i];
      acc += (double)(x.x * x.x + x.y * x.y);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((tid & 63u) == 0) seg_sum[sg] = acc;
  }
  __syncthreads();
  int start = 0;
  while (true) {
    if (tid == 0) {
      double r = *r_io;
      owner = walk_sums(seg_sum, start, kBlock, r);
      *r_io = r;
    }
    __syncthreads();
    const int own = owner;
    if (own < 0) return;  // the chunk never crosses: *r_io is the remainder after it
    const uint64_t sa = lo + (uint64_t)own * seg < end ? lo + (uint64_t)own * seg : end;
    const uint64_t sb = sa + seg < end ? sa + seg : end;
    const uint64_t sub = (sb - sa + kBlock - 1) / kBlock;
    const uint64_t a2 = sa + (uint64_t)tid * sub < sb ? sa + (uint64_t)tid * sub : sb;
    const uint64_t b2 = a2 + sub < sb ? a2 + sub : sb;
    sub_sum[tid] = sum_range(a2, b2);
    __syncthreads();
    int start2 = 0;
    while (true) {
      if (tid == 0) {
        double r = *r_io;
        owner2 = walk_sums(sub_sum, start2, kBlock, r);
        *r_io = r;
      }
      __syncthreads();
      const int own2 = owner2;
      if (own2 < 0) break;  // this segment came close but did not cross: carry on after it
      if ((int)tid == own2) {
        T r = (T)*r_io;
        for (uint64_t i = a2; i < b2; ++i) {
          const amp_t<T> x = st[i];
          r -= x.x * x.x + x.y * x.y;  // same running subtraction, same precision as the reference
          if (r <= (T)0) {
            *found = i;
            break;
          }
        }
        *r_io = (double)r;
      }
      __syncthreads();
      if (*found != ~0ull) return;
      start2 = own2 + 1;
      __syncthreads();
    }
    start = own + 1;
    __syncthreads();
  }
}

// result[0] = index (or ~0 when the chunk never crosses), result[1] = bits of the remaining r (double).
template <typename T>
__global__ __launch_bounds__(kBlock) void k_find_crossing(const amp_t<T>* __restrict__ st, uint64_t lo,
                                                          uint64_t len, double r0, uint64_t* __restrict__ result) {
  __shared__ double r_cur;
  __shared__ unsigned long long found_at;
  if (threadIdx.x == 0) {
    r_cur = r0;
    found_at = ~0ull;
  }
  __syncthreads();
  find_crossing_block<T>(st, lo, len, &r_cur, &found_at);
  __syncthreads();
  if (threadIdx.x == 0) {
    result[0] = found_at;
    result[1] = (uint64_t)__double_as_longlong(r_cur);
  }
}

// soft_measure in ONE launch (r4): the chunk sums of k_chunk_norms, and the block that finishes LAST (a ticket counter) does
// what the host did between two launches — walks the chunk sums (lanes first add up groups of chunks, lane 0 walks the 256
// group sums, then the chunks of the group that may cross) and runs find_crossing_block inside every chunk that may bring the
// remainder to <= 0.  No amplitude and no chunk sum leaves the device; the host reads 16 bytes.
// `counter` must be zero at launch (the last block leaves it zero again).  result as k_find_crossing (index over the whole vector).
template <typename T, typename E = amp_t<T>>
__global__ __launch_bounds__(kBlock) void k_chunk_norms_cross(const E* __restrict__ st, uint64_t nelems, uint64_t chunk,
                                                              double* __restrict__ partial, unsigned int* __restrict__ counter,
                                                              const amp_t<T>* __restrict__ amps, uint64_t namps, uint64_t chunk_amps,
                                                              double r0, uint64_t* __restrict__ result) {
  constexpr bool PACKED = !SameT<E, amp_t<T>>::v;
  __shared__ double smem[kBlock];
  __shared__ int last, cand_g, cand_c;
  __shared__ double r_cur;
  __shared__ unsigned long long found_at;
  {
    const uint64_t lo = (uint64_t)blockIdx.x * chunk;
    const uint64_t hi = lo + chunk < nelems ? lo + chunk : nelems;
    auto prob = [](E x) {
      if constexpr (PACKED) return prob_lo(x) + prob_hi(x);
      else return prob_of(x);
    };
    double s = 0;
    constexpr int U = 4;
    constexpr uint64_t kSpan = (uint64_t)U << kStrideShift;
    uint64_t i = lo;
    for (; i + kSpan <= hi; i += kSpan) {
#pragma unroll 1
      for (uint32_t qrow = 0; qrow < (1u << (kStrideShift - 8)); ++qrow) {
        const uint64_t i0 = i + (uint64_t)qrow * kBlock + threadIdx.x;
        E x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = __builtin_nontemporal_load(st + i0 + ((uint64_t)u << kStrideShift));
#pragma unroll
        for (int u = 0; u < U; ++u) s += prob(x[u]);
      }
    }
    for (i += threadIdx.x; i < hi; i += kBlock) s += prob(st[i]);
    const double t = block_reduce_sum(s, smem);
    if (threadIdx.x == 0) {
      __hip_atomic_store(&partial[blockIdx.x], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __threadfence();
      const unsigned int ticket = atomicAdd(counter, 1u);
      last = ticket == gridDim.x - 1u;
    }
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  const uint32_t tid = threadIdx.x;
  const uint32_t nchunks = gridDim.x;
  const uint32_t G = (nchunks + kBlock - 1) / kBlock;  // chunks per group
  auto chunk_sum = [&](uint32_t c) { return __hip_atomic_load(&partial[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  {
    double gs = 0;
    for (uint32_t c = tid * G; c < (tid + 1) * G && c < nchunks; ++c) gs += chunk_sum(c);
    smem[tid] = gs;
  }
  if (tid == 0) {
    r_cur = r0;
    found_at = ~0ull;
  }
  __syncthreads();
  int gstart = 0;
  while (true) {
    if (tid == 0) {
      double r = r_cur;
      cand_g = walk_sums(smem, gstart, kBlock, r);
      r_cur = r;
    }
    __syncthreads();
    const int cg = cand_g;
    if (cg < 0) break;
    const uint32_t c1 = ((uint32_t)cg + 1u) * G < nchunks ? ((uint32_t)cg + 1u) * G : nchunks;
    uint32_t cstart = (uint32_t)cg * G;
    while (true) {
      if (tid == 0) {
        double r = r_cur;
        int c = -1;
        for (uint32_t t = cstart; t < c1; ++t) {
          const double v = chunk_sum(t);
          if (r - v <= 1e-9 * (1.0 + v)) {
            c = (int)t;
            break;
          }
          r -= v;
        }
        cand_c = c;
        r_cur = r;
      }
      __syncthreads();
      const int cc = cand_c;
      if (cc < 0) break;
      const uint64_t lo = (uint64_t)cc * chunk_amps;
      find_crossing_block<T>(amps, lo, lo + chunk_amps < namps ? chunk_amps : namps - lo, &r_cur, &found_at);
      __syncthreads();
      if (found_at != ~0ull) break;
      cstart = (uint32_t)cc + 1u;
      __syncthreads();
    }
    if (found_at != ~0ull) break;
    gstart = cg + 1;
    __syncthreads();
  }
  if (tid == 0) {
    result[0] = found_at;
    result[1] = (uint64_t)__double_as_longlong(r_cur);
    *counter = 0u;  // ready for the next launch (stream order; the handle's own allocation, zeroed when made)
  }
}

// measure_state (measurement_ops.rs:220-269): zero what disagrees with the outcome,
// scale the rest by 1/sqrt(p) (Complex * real = (re*p, im*p)).
template <typename T>
__global__ __launch_bounds__(kBlock) void k_collapse(amp_t<T>* __restrict__ st, uint64_t namps,
                                                     uint64_t row_mask, uint64_t measured_mask,
                                                     T p_mult) {
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < namps; i += stride) {
    amp_t<T> x;
    if (((i & row_mask) ^ measured_mask) != 0) {
      x.x = 0;
      x.y = 0;
    } else {
      x = st[i];
      x.x = x.x * p_mult;
      x.y = x.y * p_mult;
    }
    st[i] = x;
  }
}

}  // namespace qipk
