// tune_gate1q.hip — standalone A/B harness for the 1-qubit sweep (not part of the product).
// Interleaves variants of the pair kernel in one process (rounds x variants) and prints GB/s
// (32 * 2^n bytes per launch) per target bit position.  Build:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/tune_gate1q.hip -o tools/tune_gate1q
#include <hip/hip_runtime.h>
#include "../rustqip_amd/csrc/qip_kernels.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct G { d2 m[4]; };

__device__ __forceinline__ d2 cmul(d2 a, d2 b) { d2 r; r.x = a.x * b.x - a.y * b.y; r.y = a.x * b.y + a.y * b.x; return r; }
__device__ __forceinline__ d2 cadd(d2 a, d2 b) { d2 r; r.x = a.x + b.x; r.y = a.y + b.y; return r; }

template <bool NT> __device__ __forceinline__ d2 ld(const d2* p) { if (NT) return __builtin_nontemporal_load(p); return *p; }
template <bool NT> __device__ __forceinline__ void st(d2* p, d2 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// direct mapping: one block = BLOCK*U consecutive pairs
template <int BLOCK, int U, bool NTL, bool NTS>
__global__ __launch_bounds__(BLOCK) void k_pair(d2* __restrict__ s, uint64_t npairs, uint32_t b, G g) {
  const uint64_t base = (uint64_t)blockIdx.x * (BLOCK * U) + threadIdx.x;
  const uint64_t lowmask = (1ull << b) - 1, tmask = 1ull << b;
  uint64_t i0[U]; d2 a0[U], a1[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const uint64_t w = base + (uint64_t)u * BLOCK;
    i0[u] = ((w >> b) << (b + 1)) | (w & lowmask);
    a0[u] = ld<NTL>(s + i0[u]); a1[u] = ld<NTL>(s + (i0[u] | tmask));
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    d2 r0 = cadd(cmul(g.m[0], a0[u]), cmul(g.m[1], a1[u]));
    d2 r1 = cadd(cmul(g.m[2], a0[u]), cmul(g.m[3], a1[u]));
    st<NTS>(s + i0[u], r0); st<NTS>(s + (i0[u] | tmask), r1);
  }
}

// U iterations of a lane are 2^SHIFT pairs apart (instead of adjacent 4-KiB runs)
template <int BLOCK, int U, int LOGU, int SHIFT, bool NTL, bool NTS>
__global__ __launch_bounds__(BLOCK) void k_xs_str(d2* __restrict__ s, uint64_t namps, uint32_t b, G g, int mode) {
  constexpr int S = SHIFT - 8;
  const uint64_t blk = blockIdx.x;
  const uint64_t hi = blk >> S, sub = blk & ((1u << S) - 1);
  const bool hib = (threadIdx.x >> b) & 1u;
  const d2 mlo = hib ? g.m[2] : g.m[0], mhi = hib ? g.m[3] : g.m[1];
  uint64_t idx[U]; d2 own[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    idx[u] = (hi << (SHIFT + LOGU)) | ((uint64_t)u << SHIFT) | (sub << 8) | threadIdx.x;
    own[u] = ld<NTL>(s + idx[u]);
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if (mode == 0) { st<NTS>(s + idx[u], cmul(g.m[0], own[u])); continue; }
    d2 o; o.x = __shfl_xor(own[u].x, 1 << b, 64); o.y = __shfl_xor(own[u].y, 1 << b, 64);
    d2 lo = hib ? o : own[u], hv = hib ? own[u] : o;
    st<NTS>(s + idx[u], cadd(cmul(mlo, lo), cmul(mhi, hv)));
  }
}

template <int BLOCK, int U, int LOGU, int SHIFT, bool NTL, bool NTS>
__global__ __launch_bounds__(BLOCK) void k_pair_str(d2* __restrict__ s, uint64_t npairs, uint32_t b, G g) {
  constexpr int S = SHIFT - 8;  // BLOCK = 256
  const uint64_t blk = blockIdx.x;
  const uint64_t hi = blk >> S, sub = blk & ((1u << S) - 1);
  const uint64_t lowmask = (1ull << b) - 1, tmask = 1ull << b;
  uint64_t i0[U]; d2 a0[U], a1[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const uint64_t w = (hi << (SHIFT + LOGU)) | ((uint64_t)u << SHIFT) | (sub << 8) | threadIdx.x;
    i0[u] = ((w >> b) << (b + 1)) | (w & lowmask);
    a0[u] = ld<NTL>(s + i0[u]); a1[u] = ld<NTL>(s + (i0[u] | tmask));
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    d2 r0 = cadd(cmul(g.m[0], a0[u]), cmul(g.m[1], a1[u]));
    d2 r1 = cadd(cmul(g.m[2], a0[u]), cmul(g.m[3], a1[u]));
    st<NTS>(s + i0[u], r0); st<NTS>(s + (i0[u] | tmask), r1);
  }
}

// persistent grid-stride: grid = CUs * k, each iteration as above
template <int BLOCK, int U, bool NTL, bool NTS>
__global__ __launch_bounds__(BLOCK) void k_pair_gs(d2* __restrict__ s, uint64_t npairs, uint32_t b, G g) {
  const uint64_t lowmask = (1ull << b) - 1, tmask = 1ull << b;
  for (uint64_t blk = blockIdx.x; blk * (BLOCK * U) < npairs; blk += gridDim.x) {
    const uint64_t base = blk * (BLOCK * U) + threadIdx.x;
    uint64_t i0[U]; d2 a0[U], a1[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint64_t w = base + (uint64_t)u * BLOCK;
      i0[u] = ((w >> b) << (b + 1)) | (w & lowmask);
      a0[u] = ld<NTL>(s + i0[u]); a1[u] = ld<NTL>(s + (i0[u] | tmask));
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      d2 r0 = cadd(cmul(g.m[0], a0[u]), cmul(g.m[1], a1[u]));
      d2 r1 = cadd(cmul(g.m[2], a0[u]), cmul(g.m[3], a1[u]));
      st<NTS>(s + i0[u], r0); st<NTS>(s + (i0[u] | tmask), r1);
    }
  }
}

// amplitude per lane + cross-lane partner (b < 6)
template <int BLOCK, int U, bool NTL, bool NTS>
__global__ __launch_bounds__(BLOCK) void k_xlane(d2* __restrict__ s, uint64_t namps, uint32_t b, G g) {
  const uint64_t base = (uint64_t)blockIdx.x * (BLOCK * U) + threadIdx.x;
  const bool hi = (threadIdx.x >> b) & 1u;
  const d2 mlo = hi ? g.m[2] : g.m[0], mhi = hi ? g.m[3] : g.m[1];
  d2 own[U];
#pragma unroll
  for (int u = 0; u < U; ++u) own[u] = ld<NTL>(s + base + (uint64_t)u * BLOCK);
#pragma unroll
  for (int u = 0; u < U; ++u) {
    d2 o; o.x = __shfl_xor(own[u].x, 1 << b, 64); o.y = __shfl_xor(own[u].y, 1 << b, 64);
    d2 lo = hi ? o : own[u], hv = hi ? own[u] : o;
    st<NTS>(s + base + (uint64_t)u * BLOCK, cadd(cmul(mlo, lo), cmul(mhi, hv)));
  }
}

// plain copy-in-place scale (upper bound for an in-place RMW stream): x *= c
template <int BLOCK, int U, bool NTL, bool NTS>
__global__ __launch_bounds__(BLOCK) void k_scale(d2* __restrict__ s, uint64_t namps, uint32_t b, G g) {
  const uint64_t base = (uint64_t)blockIdx.x * (BLOCK * U) + threadIdx.x;
  d2 x[U];
#pragma unroll
  for (int u = 0; u < U; ++u) x[u] = ld<NTL>(s + base + (uint64_t)u * BLOCK);
#pragma unroll
  for (int u = 0; u < U; ++u) st<NTS>(s + base + (uint64_t)u * BLOCK, cmul(g.m[0], x[u]));
}

// out-of-place copy (the guide's 6.29 TB/s figure): dst = src
template <int BLOCK, int U>
__global__ __launch_bounds__(BLOCK) void k_copy(d2* __restrict__ dst, const d2* __restrict__ src) {
  const uint64_t base = (uint64_t)blockIdx.x * (BLOCK * U) + threadIdx.x;
  d2 x[U];
#pragma unroll
  for (int u = 0; u < U; ++u) x[u] = src[base + (uint64_t)u * BLOCK];
#pragma unroll
  for (int u = 0; u < U; ++u) dst[base + (uint64_t)u * BLOCK] = x[u];
}

struct Variant { const char* name; void (*launch)(d2*, uint64_t, uint32_t, G, hipStream_t); bool lowbit_only; bool any_bit; };

template <int BLOCK, int U, bool NTL, bool NTS> void L_pair(d2* s, uint64_t N, uint32_t b, G g, hipStream_t st_) {
  const uint64_t np = N / 2; hipLaunchKernelGGL((k_pair<BLOCK, U, NTL, NTS>), dim3(np / (BLOCK * U)), dim3(BLOCK), 0, st_, s, np, b, g);
}
template <int U, int LOGU, int SHIFT, bool NT> void L_pair_str(d2* s, uint64_t N, uint32_t b, G g, hipStream_t st_) {
  const uint64_t np = N / 2; hipLaunchKernelGGL((k_pair_str<256, U, LOGU, SHIFT, NT, NT>), dim3(np / (256 * U)), dim3(256), 0, st_, s, np, b, g);
}
template <int U, int LOGU, int SHIFT, int MODE> void L_xs_str(d2* s, uint64_t N, uint32_t b, G g, hipStream_t st_) {
  hipLaunchKernelGGL((k_xs_str<256, U, LOGU, SHIFT, true, true>), dim3(N / (256 * U)), dim3(256), 0, st_, s, N, b, g, MODE);
}
// the product kernels themselves (rustqip_amd/csrc/qip_kernels.h), for A/B against the variants here
template <int U> void L_prod_pair(d2* s, uint64_t N, uint32_t b, G g, hipStream_t st_) {
  qipk::Ins ins; memset(&ins, 0, sizeof ins); ins.npos = 1; ins.pos[0] = b;
  qipk::Mat2<double> m; for (int e = 0; e < 4; ++e) m.m[e] = g.m[e]; m.nz = 15;
  const uint64_t np = N / 2;
  hipLaunchKernelGGL((qipk::k_gate1q_pair<double, U, false, true, 1>), dim3(np / (256 * U)), dim3(256), 0, st_, (qipk::amp_t<double>*)s, np, ins, 1ull << b, m);
}
template <int U> void L_prod_xlane(d2* s, uint64_t N, uint32_t b, G g, hipStream_t st_) {
  qipk::Ins ins; memset(&ins, 0, sizeof ins);
  qipk::Mat2<double> m; for (int e = 0; e < 4; ++e) m.m[e] = g.m[e]; m.nz = 15;
  hipLaunchKernelGGL((qipk::k_gate1q_xlane<double, U, false, true, 0>), dim3(N / (256 * U)), dim3(256), 0, st_, (qipk::amp_t<double>*)s, N, ins, b, m);
}
template <int BLOCK, int U, bool NTL, bool NTS, int PER_CU> void L_pair_gs(d2* s, uint64_t N, uint32_t b, G g, hipStream_t st_) {
  const uint64_t np = N / 2; hipLaunchKernelGGL((k_pair_gs<BLOCK, U, NTL, NTS>), dim3(256 * PER_CU), dim3(BLOCK), 0, st_, s, np, b, g);
}
template <int BLOCK, int U, bool NTL, bool NTS> void L_xlane(d2* s, uint64_t N, uint32_t b, G g, hipStream_t st_) {
  hipLaunchKernelGGL((k_xlane<BLOCK, U, NTL, NTS>), dim3(N / (BLOCK * U)), dim3(BLOCK), 0, st_, s, N, b, g);
}
template <int BLOCK, int U, bool NTL, bool NTS> void L_scale(d2* s, uint64_t N, uint32_t b, G g, hipStream_t st_) {
  hipLaunchKernelGGL((k_scale<BLOCK, U, NTL, NTS>), dim3(N / (BLOCK * U)), dim3(BLOCK), 0, st_, s, N, b, g);
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 30;
  const int rounds = argc > 2 ? atoi(argv[2]) : 5;
  const uint64_t N = 1ull << n;
  d2* s; CK(hipMalloc(&s, N * sizeof(d2)));
  d2* s2 = nullptr;
  CK(hipMemset(s, 0, N * sizeof(d2)));
  // non-trivial data: fill via a few H-like sweeps from a constant
  std::vector<d2> seed(1 << 20);
  for (size_t i = 0; i < seed.size(); ++i) { seed[i].x = 1e-3 * ((i * 2654435761u) % 1000) - 0.5; seed[i].y = 1e-3 * ((i * 40503u) % 1000) - 0.5; }
  for (uint64_t off = 0; off < N; off += seed.size()) CK(hipMemcpy(s + off, seed.data(), seed.size() * sizeof(d2), hipMemcpyHostToDevice));
  hipStream_t st_; CK(hipStreamCreate(&st_));
  const double h = 0.70710678118654752;
  G g; g.m[0] = (d2){h, 0}; g.m[1] = (d2){h, 0}; g.m[2] = (d2){h, 0}; g.m[3] = (d2){-h, 0};
  std::vector<Variant> vs = {
      {"pair u2 str32K nt      ", L_pair_str<2, 1, 11, true>, false, false},
      {"PRODUCT pair u2        ", L_prod_pair<2>, false, false},
      {"PRODUCT pair u4        ", L_prod_pair<4>, false, false},
      {"PRODUCT pair u8        ", L_prod_pair<8>, false, false},
      {"pair u8 str32K nt      ", L_pair_str<8, 3, 11, true>, false, false},
      {"xlane u4 str32K nt     ", L_xs_str<4, 2, 11, 1>, true, false},
      {"PRODUCT xlane u4       ", L_prod_xlane<4>, true, false},
      {"PRODUCT xlane u2       ", L_prod_xlane<2>, true, false},
  };
  const std::vector<int> bits = {0, 2, 5, 6, 8, 10, 11, 12, 13, 16, 20, 24, n - 1};
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("n=%d, %d rounds, GB/s = 32*2^n / t (median of rounds)\n%-24s", n, rounds, "variant \\ bit");
  for (int b : bits) printf("%7d", b);
  printf("\n");
  std::vector<std::vector<std::vector<float>>> t(vs.size(), std::vector<std::vector<float>>(bits.size()));
  for (int r = 0; r < rounds + 1; ++r)
    for (size_t bi = 0; bi < bits.size(); ++bi)
      for (size_t v = 0; v < vs.size(); ++v) {
        const int b = bits[bi];
        if (vs[v].lowbit_only && b >= 6) continue;
        if (vs[v].any_bit && bi > 0) continue;
        CK(hipEventRecord(e0, st_));
        vs[v].launch(s, N, b, g, st_);
        CK(hipEventRecord(e1, st_));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0) t[v][bi].push_back(ms);
      }
  for (size_t v = 0; v < vs.size(); ++v) {
    printf("%-24s", vs[v].name);
    for (size_t bi = 0; bi < bits.size(); ++bi) {
      if (t[v][bi].empty()) { printf("%7s", "-"); continue; }
      std::sort(t[v][bi].begin(), t[v][bi].end());
      printf("%7.0f", 32.0 * N / (t[v][bi][t[v][bi].size() / 2] * 1e-3) / 1e9);
    }
    printf("\n");
  }
  // out-of-place copy reference
  if (hipMalloc(&s2, N * sizeof(d2)) == hipSuccess) {
    std::vector<float> tc;
    for (int r = 0; r < rounds + 1; ++r) {
      CK(hipEventRecord(e0, st_));
      hipLaunchKernelGGL((k_copy<256, 4>), dim3(N / 1024), dim3(256), 0, st_, s2, s);
      CK(hipEventRecord(e1, st_)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r) tc.push_back(ms);
    }
    std::sort(tc.begin(), tc.end());
    printf("%-24s%7.0f   (out-of-place copy, read N + write N)\n", "copy b256 u4", 32.0 * N / (tc[tc.size() / 2] * 1e-3) / 1e9);
  }
  return 0;
}
