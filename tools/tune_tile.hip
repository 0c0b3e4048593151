// tune_tile.hip — standalone A/B harness for the SKELETON of the LDS-resident tile sweep (not part of the product):
// what bounds a sweep that carries G gates, and which block structure keeps HBM busy while tiles compute.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize tools/tune_tile.hip -o tools/tune_tile
//   tools/tune_tile [n = 30] [reps = 5]
// Every variant reads and writes the 2^n vector once (32 * 2^n bytes) through 2^11-amplitude tiles = index bits 0..5 plus
// five higher positions, exactly the product's tile; the "gates" are G element-wise complex factors per pass (the cost
// shape of a diagonal gate: 6 f64 operations per amplitude), P passes, each pass one LDS round trip as in k_tile_passes.
//   V0  one block per tile, 5 blocks per CU (the product's form)
//   V1  V0 without any LDS traffic or passes (registers only), still 32 KiB of LDS per block -> what 5 blocks per CU can stream
//   V2  V1 without the LDS allocation (occupancy by registers only)
//   V3  persistent blocks, next tile DMA-ed (global_load_lds) into a second LDS slot during the passes: 2 blocks per CU
//   V4  persistent blocks, next tile's loads held in registers during the passes: 4 blocks per CU
//   V5  persistent blocks without prefetch, 5 per CU (control for V3 / V4)
#include <hip/hip_runtime.h>
#include "../rustqip_amd/csrc/qip_kernels.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace qipk;
typedef amp_t<double> A;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Hp { uint32_t h[5]; uint32_t sorted[5]; };

__device__ __forceinline__ uint64_t tile_base(uint64_t t, const Hp& hp) {
  uint64_t w = t << 6;
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const uint32_t p = hp.sorted[j];
    w = ((w >> p) << (p + 1)) | (w & ((1ull << p) - 1ull));
  }
  return w;
}
__device__ __forceinline__ uint64_t row_off(int u, uint32_t wave, const Hp& hp) {  // tile row (u, wave) -> amplitude offset
  return ((uint64_t)(wave & 1u) << hp.h[0]) | ((uint64_t)((wave >> 1) & 1u) << hp.h[1]) | ((uint64_t)(u & 1) << hp.h[2]) |
         ((uint64_t)((u >> 1) & 1) << hp.h[3]) | ((uint64_t)((u >> 2) & 1) << hp.h[4]);
}

// __syncthreads() is fence + s_barrier + fence, and hipcc's workgroup-scope release waits for EVERY outstanding memory
// operation (s_waitcnt vmcnt(0)): inside a persistent loop that drains the prefetch at the first barrier of the passes.
// The passes only exchange data through LDS, so their barrier needs the LDS counter alone.
template <bool LDSONLY> __device__ __forceinline__ void tile_barrier() {
  if constexpr (LDSONLY) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  else __syncthreads();
}

// the passes: P round trips through the swizzled tile, G element-wise factors in each (pass bits 8, 9, 10 = the u bits)
template <int P, int G, bool LDSONLY = true>
__device__ __forceinline__ void passes(A* tile, uint32_t tid, A f) {
#pragma unroll
  for (int p = 0; p < P; ++p) {
    A e[8];
    const uint32_t slot_tb = tile_slot<A>(tid);
#pragma unroll
    for (int i = 0; i < 8; ++i) e[i] = tile[slot_tb ^ tile_slot<A>((uint32_t)i << 8)];
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
      for (int i = 0; i < 8; ++i) e[i] = cmul(f, e[i]);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) tile[slot_tb ^ tile_slot<A>((uint32_t)i << 8)] = e[i];
    tile_barrier<LDSONLY>();
  }
}

// REMAP 1: the four blocks an XCD receives in a row (b, b + 8, b + 16, b + 24) take four ADJACENT tiles (rows 1 KiB apart in
// every one of the tile's 32 row streams) instead of tiles 8 apart
template <int REMAP> __device__ __forceinline__ uint64_t block_tile(uint64_t b) {
  if (REMAP == 1) return (b & ~31ull) | ((b & 7ull) << 2) | ((b >> 3) & 3ull);
  if (REMAP == 2) return (b & ~63ull) | ((b & 7ull) << 3) | ((b >> 3) & 7ull);
  return b;
}
template <int P, int G, int MODE, int REMAP = 0, bool NT = true>  // MODE 0 = V0, 1 = V1 (no LDS traffic), 2 = V2 (no LDS at all)
__global__ __launch_bounds__(256, MODE == 2 ? 8 : 5) void k_v0(A* __restrict__ st, Hp hp, A f) {
  extern __shared__ __attribute__((aligned(16))) unsigned char raw[];
  A* tile = reinterpret_cast<A*>(raw);
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint64_t base = tile_base(block_tile<REMAP>(blockIdx.x), hp);
  A x[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) x[u] = ldg<NT>(st + (base | row_off(u, wave, hp)) + lane);
  if (MODE == 0 || MODE == 3) {
    const uint32_t slot_tid = tile_slot<A>(tid);
#pragma unroll
    for (int u = 0; u < 8; ++u) tile[slot_tid ^ tile_slot<A>((uint32_t)u << 8)] = x[u];
    tile_barrier<MODE == 3>();
    passes<P, G, MODE == 3>(tile, tid, f);
#pragma unroll
    for (int u = 0; u < 8; ++u) x[u] = tile[slot_tid ^ tile_slot<A>((uint32_t)u << 8)];
  } else {
#pragma unroll
    for (int g = 0; g < G * P; ++g) {
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = cmul(f, x[i]);
    }
  }
#pragma unroll
  for (int u = 0; u < 8; ++u) stg<NT>(st + (base | row_off(u, wave, hp)) + lane, x[u]);
}

// "order" study: block -> tile through a bit permutation given at run time (tile-index bit src[i] is driven by block-index bit i):
// which tiles are in flight together, and which share an XCD (blocks go round-robin over the 8 XCDs: block bits 0..2).
struct Ord { uint8_t src[24]; uint32_t nb; };
__global__ __launch_bounds__(256, 5) void k_ord(A* __restrict__ st, Hp hp, Ord o, A f) {
  extern __shared__ __attribute__((aligned(16))) unsigned char raw[];
  A* tile = reinterpret_cast<A*>(raw);
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  uint64_t t = 0;
  for (uint32_t i = 0; i < o.nb; ++i) t |= (uint64_t)((blockIdx.x >> i) & 1u) << o.src[i];
  const uint64_t base = tile_base(t, hp);
  A x[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) x[u] = ldg<true>(st + (base | row_off(u, wave, hp)) + lane);
  const uint32_t slot_tid = tile_slot<A>(tid);
#pragma unroll
  for (int u = 0; u < 8; ++u) tile[slot_tid ^ tile_slot<A>((uint32_t)u << 8)] = x[u];
  tile_barrier<false>();
  passes<1, 2, false>(tile, tid, f);
#pragma unroll
  for (int u = 0; u < 8; ++u) x[u] = tile[slot_tid ^ tile_slot<A>((uint32_t)u << 8)];
#pragma unroll
  for (int u = 0; u < 8; ++u) stg<true>(st + (base | row_off(u, wave, hp)) + lane, x[u]);
}

// persistent, MODE 4 = next tile's loads in registers across the passes, 5 = no prefetch.
// Loop shape: the wait for the prefetched tile sits in the SAME iteration that issued it (issue -> passes -> stores ->
// wait -> LDS write), so hipcc's counter bookkeeping sees "8 loads, then 8 stores" every time and waits with vmcnt(8..15):
// for the loads only.  With the wait at the top of the next iteration it merges the first entry (no stores yet) with the
// back edge and waits with vmcnt(0..7), i.e. for the previous tile's stores as well.
template <int P, int G, int MODE>
__global__ __launch_bounds__(256, MODE == 4 ? 4 : 5) void k_persist(A* __restrict__ st, uint64_t ntiles, Hp hp, A f) {
  extern __shared__ __attribute__((aligned(16))) unsigned char raw[];
  A* tile = reinterpret_cast<A*>(raw);
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t slot_tid = tile_slot<A>(tid);
  uint64_t t = blockIdx.x;
  uint64_t base = tile_base(t, hp);
  {
    A x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) x[u] = __builtin_nontemporal_load(st + (base | row_off(u, wave, hp)) + lane);
#pragma unroll
    for (int u = 0; u < 8; ++u) tile[slot_tid ^ tile_slot<A>((uint32_t)u << 8)] = x[u];
  }
  tile_barrier<true>();
  for (;;) {
    const uint64_t tn = t + gridDim.x;
    const bool more = tn < ntiles;
    const uint64_t base_n = tile_base(tn, hp);
    A x[8];
    if (MODE == 4 && more) {
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = __builtin_nontemporal_load(st + (base_n | row_off(u, wave, hp)) + lane);
    }
    passes<P, G>(tile, tid, f);
    A y[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) y[u] = tile[slot_tid ^ tile_slot<A>((uint32_t)u << 8)];
#pragma unroll
    for (int u = 0; u < 8; ++u) __builtin_nontemporal_store(y[u], st + (base | row_off(u, wave, hp)) + lane);
    if (!more) break;
    if (MODE == 5) {
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = __builtin_nontemporal_load(st + (base_n | row_off(u, wave, hp)) + lane);
    }
    tile_barrier<true>();  // every lane has read its share of this tile
#pragma unroll
    for (int u = 0; u < 8; ++u) tile[slot_tid ^ tile_slot<A>((uint32_t)u << 8)] = x[u];
    tile_barrier<true>();
    t = tn;
    base = base_n;
  }
}

// persistent, the next tile DMA-ed straight into the other LDS slot (no registers): global_load_lds writes lane l's 16 bytes
// at row base + 16 l, so the tile's swizzle goes on the SOURCE side: lane l fetches the element whose slot is l
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
template <int P, int G, int BPC>
__global__ __launch_bounds__(256, BPC) void k_dma(A* __restrict__ st, uint64_t ntiles, Hp hp, A f) {
  extern __shared__ __attribute__((aligned(16))) unsigned char raw[];
  A* slot0 = reinterpret_cast<A*>(raw);
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t slot_tid = tile_slot<A>(tid);
  auto fetch = [&](uint64_t b, A* dst) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const uint32_t R = ((uint32_t)u << 2) | wave;                      // tile row = tile index >> 6
      const uint32_t src_lane = lane ^ ((R * 4u + (lane >> 4)) & 15u);     // element of the row whose swizzled slot is `lane`
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(st + (b | row_off(u, wave, hp)) + src_lane), (lds_ptr_t)(dst + R * 64u), 16, 0, 0);
    }
  };
  uint64_t t = blockIdx.x;
  uint64_t base = tile_base(t, hp);
  uint32_t cur = 0;
  fetch(base, slot0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  tile_barrier<true>();
  for (;;) {
    A* tile = slot0 + (cur ? 2048 : 0);
    const uint64_t tn = t + gridDim.x;
    const bool more = tn < ntiles;
    const uint64_t base_n = tile_base(tn, hp);
    if (more) fetch(base_n, slot0 + (cur ? 0 : 2048));
    passes<P, G>(tile, tid, f);
    A y[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) y[u] = tile[slot_tid ^ tile_slot<A>((uint32_t)u << 8)];
#pragma unroll
    for (int u = 0; u < 8; ++u) __builtin_nontemporal_store(y[u], st + (base | row_off(u, wave, hp)) + lane);
    if (!more) break;
    // the next tile's DMA is older than the eight stores just issued: wait for it (own loads by the counter, the other
    // waves' by the barrier), not for the stores; the barrier also frees this slot for the DMA after next
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    tile_barrier<true>();
    t = tn;
    base = base_n;
    cur ^= 1u;
  }
}

// r4 probe: a tile whose ELEVEN index positions are all free parameters — T[0..5] are driven by the lane id (the product: 0..5 = one
// contiguous 1-KiB row per wave-level access), T[6..7] by the wave id, T[8..10] by a lane's eight back-to-back accesses; the block
// index drives every other position, ascending.  No LDS (the V2 skeleton): read, scale, write back.
struct Tp { uint32_t t[11]; uint32_t sorted[11]; };
__global__ __launch_bounds__(256, 8) void k_probe(A* __restrict__ st, Tp tp, A f) {
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  uint64_t w = blockIdx.x;
#pragma unroll
  for (int j = 0; j < 11; ++j) {
    const uint32_t p = tp.sorted[j];
    w = ((w >> p) << (p + 1)) | (w & ((1ull << p) - 1ull));
  }
  uint64_t off = 0;
#pragma unroll
  for (int k = 0; k < 6; ++k) off |= (uint64_t)((lane >> k) & 1u) << tp.t[k];
  off |= ((uint64_t)(wave & 1u) << tp.t[6]) | ((uint64_t)(wave >> 1) << tp.t[7]);
  A x[8];
  uint64_t a[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    a[u] = w | off | ((uint64_t)(u & 1) << tp.t[8]) | ((uint64_t)((u >> 1) & 1) << tp.t[9]) | ((uint64_t)((u >> 2) & 1) << tp.t[10]);
    x[u] = ldg<true>(st + a[u]);
  }
#pragma unroll
  for (int u = 0; u < 8; ++u) stg<true>(st + a[u], cmul(f, x[u]));
}

// r6 probe for the out-of-place bit-permutation sweep: TB tile positions on the SOURCE side (S[0..5] lane, S[6..7] wave, S[8..] a lane's
// back-to-back loads) and TB on the DESTINATION side, two buffers; the block index drives the other positions ascending on both sides.
// No LDS and no data consistency (what is timed is the DRAM address pattern of a tile's reads and writes, which is what set the r4 finding).
template <int TB> struct Tp2 { uint32_t s[TB], d[TB], ss[TB], ds[TB]; };
template <int TB>
__global__ __launch_bounds__(256, TB >= 12 ? 2 : (TB == 11 ? 4 : 8)) void k_probe2(const A* __restrict__ in, A* __restrict__ out, Tp2<TB> tp) {
  constexpr int E = (1 << TB) / 256;
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  uint64_t ws = blockIdx.x, wd = blockIdx.x;
#pragma unroll
  for (int j = 0; j < TB; ++j) {
    const uint32_t p = tp.ss[j], q = tp.ds[j];
    ws = ((ws >> p) << (p + 1)) | (ws & ((1ull << p) - 1ull));
    wd = ((wd >> q) << (q + 1)) | (wd & ((1ull << q) - 1ull));
  }
  uint64_t os = 0, od = 0;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    os |= (uint64_t)((lane >> k) & 1u) << tp.s[k];
    od |= (uint64_t)((lane >> k) & 1u) << tp.d[k];
  }
  os |= ((uint64_t)(wave & 1u) << tp.s[6]) | ((uint64_t)(wave >> 1) << tp.s[7]);
  od |= ((uint64_t)(wave & 1u) << tp.d[6]) | ((uint64_t)(wave >> 1) << tp.d[7]);
  A x[E];
#pragma unroll
  for (int u = 0; u < E; ++u) {
    uint64_t a = ws | os;
#pragma unroll
    for (int b = 0; b < TB - 8; ++b) a |= (uint64_t)((u >> b) & 1) << tp.s[8 + b];
    x[u] = ldg<true>(in + a);
  }
#pragma unroll
  for (int u = 0; u < E; ++u) {
    uint64_t a = wd | od;
#pragma unroll
    for (int b = 0; b < TB - 8; ++b) a |= (uint64_t)((u >> b) & 1) << tp.d[8 + b];
    stg<true>(out + a, x[u]);
  }
}

// r4 experiment (with a kill criterion): a 13-bit tile held in REGISTERS — 32 amplitudes per lane (five
// register bits), 256 lanes per block, seven free positions per sweep instead of five — with LDS only as a 32-KiB transposition
// buffer: a pass moves the tile through it in four quarters.  Keep only if a light sweep stays <= 6.5 ms at >= 2 blocks per CU.
struct Hp7 { uint32_t h[7]; uint32_t sorted[7]; uint32_t p5; };
template <int P, int G, int BPC>
__global__ __launch_bounds__(256, BPC) void k_v13(A* __restrict__ st, Hp7 hp, A f) {
  extern __shared__ __attribute__((aligned(16))) unsigned char raw[];
  A* buf = reinterpret_cast<A*>(raw);
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  uint64_t w = (uint64_t)blockIdx.x << 6;
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    const uint32_t p = hp.sorted[j];
    w = ((w >> p) << (p + 1)) | (w & ((1ull << p) - 1ull));
  }
  if (hp.p5 != 5u) {  // split rows: positions in the space where p5 and 5 have traded places
    const uint64_t b = (w >> hp.p5) & 1ull;
    w = (w & ~(1ull << hp.p5)) | (b << 5);
  }
  w |= ((uint64_t)(wave & 1u) << hp.h[0]) | ((uint64_t)(wave >> 1) << hp.h[1]);
  const uint32_t lane_off = (lane & 31u) | ((lane >> 5) << hp.p5);
  A x[32];
  auto off = [&](int u) {  // (wave-uniform: scalar registers)
    uint64_t o = 0;
#pragma unroll
    for (int b = 0; b < 5; ++b) o |= (uint64_t)((u >> b) & 1) << hp.h[2 + b];
    return o;
  };
#pragma unroll
  for (int u = 0; u < 32; ++u) x[u] = ldg<true>(st + (w | off(u)) + lane_off);
#pragma unroll
  for (int p = 0; p < P; ++p) {
    // one pass: the tile through LDS in four quarters of 8 elements per lane, written in one arrangement and read in another
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      const uint32_t slot_w = tile_slot<A>(tid);
#pragma unroll
      for (int i = 0; i < 8; ++i) buf[slot_w ^ tile_slot<A>((uint32_t)i << 8)] = x[8 * qd + i];
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      const uint32_t tr = ((tid & 7u) << 5) | (tid >> 3);  // another lane -> tile-bit assignment
      const uint32_t slot_r = tile_slot<A>(tr);
#pragma unroll
      for (int i = 0; i < 8; ++i) x[8 * qd + i] = buf[slot_r ^ tile_slot<A>((uint32_t)i << 8)];
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
      for (int i = 0; i < 32; ++i) x[i] = cmul(f, x[i]);
    }
  }
#pragma unroll
  for (int u = 0; u < 32; ++u) stg<true>(st + (w | off(u)) + lane_off, x[u]);
}

// r6 experiment (kill criterion first): the same 13-bit tile with 16 amplitudes per lane (FOUR register bits) and 512
// lanes per block — half the registers per lane, twice the waves: 2 blocks per CU are 4 waves per SIMD (<= 128 VGPRs), 3 blocks per CU 6 waves
// per SIMD (<= 84 VGPRs, 64 of them amplitudes).  LDS again only a 32-KiB transposition buffer: a pass moves the tile through it in four rounds
// of 4 elements per lane.  Keep only if a light sweep is <= 5.2 ms at >= 3 blocks per CU AND the loaded shapes (P = 3, 4) beat k_v13.
template <int P, int G, int BPC>
__global__ __launch_bounds__(512, BPC) void k_v13w(A* __restrict__ st, Hp7 hp, A f) {
  extern __shared__ __attribute__((aligned(16))) unsigned char raw[];
  A* buf = reinterpret_cast<A*>(raw);
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0..7: three tile bits
  uint64_t w = (uint64_t)blockIdx.x << 6;
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    const uint32_t p = hp.sorted[j];
    w = ((w >> p) << (p + 1)) | (w & ((1ull << p) - 1ull));
  }
  if (hp.p5 != 5u) {
    const uint64_t b = (w >> hp.p5) & 1ull;
    w = (w & ~(1ull << hp.p5)) | (b << 5);
  }
  w |= ((uint64_t)(wave & 1u) << hp.h[0]) | ((uint64_t)((wave >> 1) & 1u) << hp.h[1]) | ((uint64_t)(wave >> 2) << hp.h[2]);
  const uint32_t lane_off = (lane & 31u) | ((lane >> 5) << hp.p5);
  A x[16];
  auto off = [&](int u) {
    uint64_t o = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) o |= (uint64_t)((u >> b) & 1) << hp.h[3 + b];
    return o;
  };
#pragma unroll
  for (int u = 0; u < 16; ++u) x[u] = ldg<true>(st + (w | off(u)) + lane_off);
#pragma unroll
  for (int p = 0; p < P; ++p) {
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {  // 4 elements per lane x 512 lanes = 2^11 amplitudes = the 32-KiB buffer
      const uint32_t slot_w = tile_slot<A>(tid);
#pragma unroll
      for (int i = 0; i < 4; ++i) buf[slot_w ^ tile_slot<A>((uint32_t)i << 9)] = x[4 * qd + i];
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      const uint32_t tr = ((tid & 15u) << 5) | (tid >> 4);  // another lane -> tile-bit assignment
      const uint32_t slot_r = tile_slot<A>(tr);
#pragma unroll
      for (int i = 0; i < 4; ++i) x[4 * qd + i] = buf[slot_r ^ tile_slot<A>((uint32_t)i << 9)];
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
      for (int i = 0; i < 16; ++i) x[i] = cmul(f, x[i]);
    }
  }
#pragma unroll
  for (int u = 0; u < 16; ++u) stg<true>(st + (w | off(u)) + lane_off, x[u]);
}

__global__ void k_init(A* st, uint64_t n) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    A v;
    v.x = 1e-5 * (double)((i * 2654435761ull) & 0xffff) + 0.25;
    v.y = 1e-5 * (double)((i * 40503ull) & 0xffff) - 0.125;
    st[i] = v;
  }
}
__global__ void k_sum(const A* st, uint64_t n, double* out) {
  double s = 0;
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    s += st[i].x * (double)((i % 7) + 1) + st[i].y * (double)((i % 5) + 1);
  atomicAdd(out, s);
}

static A* g_st;
static uint64_t g_n;
static int g_reps;
static int g_cus;

template <typename F> static void run(const char* name, int P, int G, const char* hname, F&& launch) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  k_init<<<4096, 256>>>(g_st, g_n);
  launch();  // warm
  CK(hipDeviceSynchronize());
  double* dsum;
  CK(hipMalloc(&dsum, 8));
  CK(hipMemset(dsum, 0, 8));
  k_sum<<<4096, 256>>>(g_st, 1ull << 24, dsum);
  double hs = 0;
  CK(hipMemcpy(&hs, dsum, 8, hipMemcpyDeviceToHost));
  CK(hipFree(dsum));
  float best = 1e9f;
  for (int r = 0; r < g_reps; ++r) {
    CK(hipEventRecord(e0));
    launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = std::min(best, ms);
  }
  CK(hipGetLastError());
  printf("%-26s P=%d G=%-3d %-14s %7.3f ms  %6.0f GB/s  chk=%.9e\n", name, P, G, hname, best, 32.0 * (double)g_n / best / 1e6, hs);
  fflush(stdout);
  CK(hipEventDestroy(e0));
  CK(hipEventDestroy(e1));
}

template <int P, int G> static void variants(const Hp& hp, const char* hname) {
  const uint64_t ntiles = g_n >> 11;
  A f;
  f.x = 0.6;
  f.y = 0.8;
  const size_t lds = 32768;
  run("V0 block/tile 5pCU", P, G, hname, [&] { hipLaunchKernelGGL((k_v0<P, G, 0>), dim3((unsigned)ntiles), dim3(256), lds, 0, g_st, hp, f); });
  run("V0b V0, LDS-only barrier", P, G, hname, [&] { hipLaunchKernelGGL((k_v0<P, G, 3>), dim3((unsigned)ntiles), dim3(256), lds, 0, g_st, hp, f); });
  run("V1 regs only, 32K LDS", P, G, hname, [&] { hipLaunchKernelGGL((k_v0<P, G, 1>), dim3((unsigned)ntiles), dim3(256), lds, 0, g_st, hp, f); });
  run("V2 regs only, no LDS", P, G, hname, [&] { hipLaunchKernelGGL((k_v0<P, G, 2>), dim3((unsigned)ntiles), dim3(256), 0, 0, g_st, hp, f); });
  run("V3 persist DMA 2pCU", P, G, hname, [&] { hipLaunchKernelGGL((k_dma<P, G, 2>), dim3((unsigned)std::min<uint64_t>(ntiles, g_cus * 2)), dim3(256), 2 * lds, 0, g_st, ntiles, hp, f); });
  run("V4 persist regs 4pCU", P, G, hname, [&] { hipLaunchKernelGGL((k_persist<P, G, 4>), dim3((unsigned)std::min<uint64_t>(ntiles, g_cus * 4)), dim3(256), lds, 0, g_st, ntiles, hp, f); });
  run("V5 persist nopre 5pCU", P, G, hname, [&] { hipLaunchKernelGGL((k_persist<P, G, 5>), dim3((unsigned)std::min<uint64_t>(ntiles, g_cus * 5)), dim3(256), lds, 0, g_st, ntiles, hp, f); });
}

static void light(const Hp& hp, const char* hname) {
  const uint64_t ntiles = g_n >> 11;
  A f;
  f.x = 0.6;
  f.y = 0.8;
  const size_t lds = 32768;
  run("V0", 1, 2, hname, [&] { hipLaunchKernelGGL((k_v0<1, 2, 0>), dim3((unsigned)ntiles), dim3(256), lds, 0, g_st, hp, f); });
  run("V0 remap4", 1, 2, hname, [&] { hipLaunchKernelGGL((k_v0<1, 2, 0, 1>), dim3((unsigned)ntiles), dim3(256), lds, 0, g_st, hp, f); });
  run("V0 remap8", 1, 2, hname, [&] { hipLaunchKernelGGL((k_v0<1, 2, 0, 2>), dim3((unsigned)ntiles), dim3(256), lds, 0, g_st, hp, f); });
  run("V0 plain ld/st", 1, 2, hname, [&] { hipLaunchKernelGGL((k_v0<1, 2, 0, 0, false>), dim3((unsigned)ntiles), dim3(256), lds, 0, g_st, hp, f); });
  run("V2 no LDS", 1, 2, hname, [&] { hipLaunchKernelGGL((k_v0<1, 2, 2>), dim3((unsigned)ntiles), dim3(256), 0, 0, g_st, hp, f); });
  run("V2 no LDS remap4", 1, 2, hname, [&] { hipLaunchKernelGGL((k_v0<1, 2, 2, 1>), dim3((unsigned)ntiles), dim3(256), 0, 0, g_st, hp, f); });
}

// one tile shape under a family of block -> tile orders.  An order is the list of FREE positions (those outside the tile,
// >= 6) in the sequence the block-index bits drive them, lowest block bit first.
static void orders(std::vector<uint32_t> h, const char* hname, int n);

static Hp make_hp(std::vector<uint32_t> h) {
  Hp hp;
  for (int j = 0; j < 5; ++j) hp.h[j] = h[j];
  std::sort(h.begin(), h.end());
  for (int j = 0; j < 5; ++j) hp.sorted[j] = h[j];
  return hp;
}

static void orders(std::vector<uint32_t> h, const char* hname, int n) {
  const Hp hp = make_hp(h);
  std::vector<uint32_t> free_pos;
  for (uint32_t p = 6; p < (uint32_t)n; ++p)
    if (std::find(h.begin(), h.end(), p) == h.end()) free_pos.push_back(p);
  const uint32_t nb = (uint32_t)free_pos.size();
  const uint64_t ntiles = g_n >> 11;
  A f;
  f.x = 0.6;
  f.y = 0.8;
  auto run_order = [&](const char* oname, const std::vector<uint32_t>& seq) {  // seq: free positions, lowest block bit first
    Ord o;
    memset(&o, 0, sizeof o);
    o.nb = nb;
    for (uint32_t i = 0; i < nb; ++i) o.src[i] = (uint8_t)(std::find(free_pos.begin(), free_pos.end(), seq[i]) - free_pos.begin());
    run(oname, 1, 2, hname, [&] { hipLaunchKernelGGL(k_ord, dim3((unsigned)ntiles), dim3(256), 32768, 0, g_st, hp, o, f); });
  };
  auto from = [&](uint32_t P) {  // positions >= P ascending first, then the ones below
    std::vector<uint32_t> s;
    for (uint32_t p : free_pos) if (p >= P) s.push_back(p);
    for (uint32_t p : free_pos) if (p < P) s.push_back(p);
    return s;
  };
  auto xcd = [&](uint32_t P, bool rest_from11) {  // the three XCD bits <- the first three free positions >= P; the rest ascending (or from 11)
    std::vector<uint32_t> first, rest;
    for (uint32_t p : free_pos) (p >= P && first.size() < 3 ? first : rest).push_back(p);
    if (rest_from11) {
      std::vector<uint32_t> r2;
      for (uint32_t p : rest) if (p >= 11) r2.push_back(p);
      for (uint32_t p : rest) if (p < 11) r2.push_back(p);
      rest = r2;
    }
    first.insert(first.end(), rest.begin(), rest.end());
    return first;
  };
  char name[64];
  run_order("asc", free_pos);
  { std::vector<uint32_t> d(free_pos.rbegin(), free_pos.rend()); run_order("desc", d); }
  for (uint32_t P : {8u, 9u, 11u, 13u, 16u, 19u}) { snprintf(name, sizeof name, "from %u", P); run_order(name, from(P)); }
  for (uint32_t P : {8u, 9u, 11u, 13u, 16u, 20u, 24u}) { snprintf(name, sizeof name, "xcd<-%u.. rest asc", P); run_order(name, xcd(P, false)); }
  for (uint32_t P : {6u, 8u, 11u, 16u, 20u, 24u}) { snprintf(name, sizeof name, "xcd<-%u.. rest from11", P); run_order(name, xcd(P, true)); }
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 30;
  g_reps = argc > 2 ? atoi(argv[2]) : 5;
  g_n = 1ull << n;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  g_cus = prop.multiProcessorCount;
  printf("device %s, %d CUs, n = %d\n", prop.name, g_cus, n);
  CK(hipMalloc(&g_st, g_n * sizeof(A)));
  CK(hipFuncSetAttribute((const void*)k_dma<1, 0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  const Hp top = make_hp({(uint32_t)n - 5, (uint32_t)n - 4, (uint32_t)n - 3, (uint32_t)n - 2, (uint32_t)n - 1});
  const Hp low = make_hp({6, 7, 8, 9, 10});
  const Hp mix = make_hp({12, 15, 18, 21, 24});
#define DMA_ATTR(P, G) CK(hipFuncSetAttribute((const void*)k_dma<P, G, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536))
  DMA_ATTR(1, 0); DMA_ATTR(1, 2); DMA_ATTR(2, 8); DMA_ATTR(2, 32); DMA_ATTR(2, 64);
  if (argc > 3 && !strcmp(argv[3], "light")) {
    const uint32_t N = (uint32_t)n;
    struct { std::vector<uint32_t> h; const char* name; } sets[] = {
        {{6, 7, 8, 9, 10}, "w6,7 u8,9,10"},        {{6, 7, 11, 12, 13}, "w6,7 u11,12,13"},   {{6, 7, 12, 13, 14}, "w6,7 u12,13,14"},
        {{8, 9, 11, 12, 13}, "w8,9 u11,12,13"},    {{11, 12, 13, 14, 15}, "w11,12 u13-15"},  {{6, 7, N - 3, N - 2, N - 1}, "w6,7 u top3"},
        {{N - 2, N - 1, 6, 7, 8}, "w top2 u6,7,8"}, {{N - 5, N - 4, N - 3, N - 2, N - 1}, "top5"}, {{N - 3, N - 2, N - 1, N - 5, N - 4}, "top5 (u = lower two + ...)"},
        {{12, 15, 18, 21, 24}, "w12,15 u18,21,24"}, {{6, 15, 18, 21, 24}, "w6,15 u18,21,24"}, {{6, 7, 18, 21, 24}, "w6,7 u18,21,24"},
        {{18, 21, 24, 6, 7}, "w18,21 u24,6,7"}};
    for (auto& s : sets) light(make_hp(s.h), s.name);
    CK(hipFree(g_st));
    return 0;
  }
  if (argc > 3 && !strcmp(argv[3], "scan")) {
    // r4: which index positions are cheap in which ROLE of the tile.  h[0], h[1] = the positions the wave id fills (four waves of a
    // block), h[2..4] = the positions a lane's eight back-to-back loads / stores differ in.  One pass, two factors (a light sweep), the
    // product's block -> tile order (ascending).  Output: one line per set, `scan <tag> <h0> <h1> <h2> <h3> <h4> <ms>`.
    const uint32_t N = (uint32_t)n;
    const uint64_t ntiles = g_n >> 11;
    A f;
    f.x = 0.6;
    f.y = 0.8;
    auto probe = [&](const char* tag, std::vector<uint32_t> h) {
      std::vector<uint32_t> chk = h;
      std::sort(chk.begin(), chk.end());
      for (int j = 0; j < 5; ++j)
        if (chk[j] < 6 || chk[j] >= N || (j && chk[j] == chk[j - 1])) return;
      const Hp hp = make_hp(h);
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0));
      CK(hipEventCreate(&e1));
      hipLaunchKernelGGL((k_v0<1, 2, 0>), dim3((unsigned)ntiles), dim3(256), 32768, 0, g_st, hp, f);
      float best = 1e9f;
      for (int r = 0; r < g_reps; ++r) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_v0<1, 2, 0>), dim3((unsigned)ntiles), dim3(256), 32768, 0, g_st, hp, f);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
      }
      CK(hipGetLastError());
      printf("scan %s %u %u %u %u %u %.3f\n", tag, h[0], h[1], h[2], h[3], h[4], best);
      fflush(stdout);
      CK(hipEventDestroy(e0));
      CK(hipEventDestroy(e1));
    };
    k_init<<<4096, 256>>>(g_st, g_n);
    CK(hipDeviceSynchronize());
    for (uint32_t a = 8; a + 2 < N; ++a) probe("u3", {6, 7, a, a + 1, a + 2});
    for (uint32_t a = 6; a + 1 < N; ++a) probe("w2", {a, a + 1, 11, 12, 13});
    for (uint32_t a = 6; a + 4 < N; ++a) probe("run5", {a, a + 1, a + 2, a + 3, a + 4});
    for (int slot = 0; slot < 5; ++slot)
      for (uint32_t p = 6; p < N; ++p) {
        std::vector<uint32_t> h = {11, 12, 13, 14, 15};
        h[slot] = p;
        char tag[16];
        snprintf(tag, sizeof tag, "sub%d", slot);
        probe(tag, h);
      }
    {  // the top five in every role assignment (which two are the wave bits)
      const uint32_t t[5] = {N - 5, N - 4, N - 3, N - 2, N - 1};
      for (int a = 0; a < 5; ++a)
        for (int b = a + 1; b < 5; ++b) {
          std::vector<uint32_t> h = {t[a], t[b]};
          for (int c = 0; c < 5; ++c)
            if (c != a && c != b) h.push_back(t[c]);
          probe("top5", h);
        }
    }
    {  // a dense k = 4 gate on positions {0, 14, 24, 29}: three forced positions + two free ones, in both roles
      for (uint32_t x = 6; x + 1 < 24; ++x) {
        if (x == 14 || x + 1 == 14) continue;
        probe("k4wfree", {x, x + 1, 14, 24, 29});
        probe("k4ufree", {14, 24, x, x + 1, 29});
        probe("k4ufre2", {24, 29, x, x + 1, 14});
      }
    }
    {  // strided sets
      for (uint32_t st = 2; st <= 5; ++st)
        for (uint32_t a = 6; a + 4 * st < N; a += 2) probe("strd", {a, a + st, a + 2 * st, a + 3 * st, a + 4 * st});
    }
    // the same bytes through a plain streaming kernel for reference: every position "free"
    CK(hipFree(g_st));
    return 0;
  }
  if (argc > 3 && !strcmp(argv[3], "v13")) {
    const uint32_t N = (uint32_t)n;
    const uint64_t ntiles = g_n >> 13;
    A f;
    f.x = 0.6;
    f.y = 0.8;
    auto mk7 = [&](std::vector<uint32_t> h, uint32_t p5) {
      Hp7 hp;
      for (int j = 0; j < 7; ++j) hp.h[j] = h[j];
      std::vector<uint32_t> sp = h;
      for (uint32_t& v : sp) if (v == 5u) v = p5;
      std::sort(sp.begin(), sp.end());
      for (int j = 0; j < 7; ++j) hp.sorted[j] = sp[j];
      hp.p5 = p5;
      return hp;
    };
    struct { std::vector<uint32_t> h; uint32_t p5; const char* name; } sets[] = {
        {{11, 12, 13, 14, 15, 16, 17}, 5, "contig 11..17"},          {{12, 13, 14, 15, 16, 17, 18}, 11, "split 12..18"},
        {{N - 7, N - 6, N - 5, N - 4, N - 3, N - 2, N - 1}, 5, "contig top7"}, {{N - 7, N - 6, N - 5, N - 4, N - 3, N - 2, N - 1}, 11, "split top7"},
        {{12, 14, 17, 20, 22, 24, N - 1}, 11, "split scattered"},     {{6, 9, 14, 18, 21, 24, N - 1}, 11, "split scattered2"},
        {{5, 6, 7, 8, 9, 10, 12}, 11, "split low"}};
#define V13(P, G, BPC) run("V13 " #BPC " blocks/CU", P, G, s.name, [&] { hipLaunchKernelGGL((k_v13<P, G, BPC>), dim3((unsigned)ntiles), dim3(256), 32768, 0, g_st, hp, f); })
    for (auto& s : sets) {
      const Hp7 hp = mk7(s.h, s.p5);
      V13(1, 2, 2);
      V13(1, 2, 3);
      V13(2, 8, 2);
      V13(2, 8, 3);
      V13(3, 8, 3);
    }
#undef V13
    CK(hipFree(g_st));
    return 0;
  }
  if (argc > 3 && !strcmp(argv[3], "v13w")) {
    // r6: 512 lanes x 16 amplitudes against 256 lanes x 32, same tile, same positions, light and loaded shapes
    const uint32_t N = (uint32_t)n;
    const uint64_t ntiles = g_n >> 13;
    A f;
    f.x = 0.6;
    f.y = 0.8;
    auto mk7 = [&](std::vector<uint32_t> h, uint32_t p5) {
      Hp7 hp;
      for (int j = 0; j < 7; ++j) hp.h[j] = h[j];
      std::vector<uint32_t> sp = h;
      for (uint32_t& v : sp) if (v == 5u) v = p5;
      std::sort(sp.begin(), sp.end());
      for (int j = 0; j < 7; ++j) hp.sorted[j] = sp[j];
      hp.p5 = p5;
      return hp;
    };
    struct { std::vector<uint32_t> h; uint32_t p5; const char* name; } sets[] = {
        {{12, 13, 14, 15, 16, 17, 18}, 11, "split 12..18"},
        {{N - 7, N - 6, N - 5, N - 4, N - 3, N - 2, N - 1}, 11, "split top7"},
        {{12, 14, 17, 20, 22, 24, N - 1}, 11, "split scattered"},
        {{5, 6, 7, 8, 9, 10, 12}, 11, "split low"}};
#define V13(P, G, BPC) run("V13  256x32 " #BPC "/CU", P, G, s.name, [&] { hipLaunchKernelGGL((k_v13<P, G, BPC>), dim3((unsigned)ntiles), dim3(256), 32768, 0, g_st, hp, f); })
#define V13W(P, G, BPC) run("V13W 512x16 " #BPC "/CU", P, G, s.name, [&] { hipLaunchKernelGGL((k_v13w<P, G, BPC>), dim3((unsigned)ntiles), dim3(512), 32768, 0, g_st, hp, f); })
    for (auto& s : sets) {
      const Hp7 hp = mk7(s.h, s.p5);
      V13(1, 2, 2);  V13W(1, 2, 2);  V13W(1, 2, 3);
      V13(3, 8, 2);  V13W(3, 8, 2);  V13W(3, 8, 3);
      V13(4, 16, 2); V13W(4, 16, 2); V13W(4, 16, 3);
    }
#undef V13
#undef V13W
    CK(hipFree(g_st));
    return 0;
  }
  if (argc > 3 && !strcmp(argv[3], "probe")) {
    // `probe <l0> ... <l5> <w0> <w1> <u0> <u1> <u2>` lines on stdin -> one timing each
    const uint64_t ntiles = g_n >> 11;
    A f;
    f.x = 0.6;
    f.y = 0.8;
    k_init<<<4096, 256>>>(g_st, g_n);
    CK(hipDeviceSynchronize());
    char line[256];
    while (fgets(line, sizeof line, stdin)) {
      Tp tp;
      char tag[64];
      if (sscanf(line, "%63s %u %u %u %u %u %u %u %u %u %u %u", tag, &tp.t[0], &tp.t[1], &tp.t[2], &tp.t[3], &tp.t[4], &tp.t[5], &tp.t[6], &tp.t[7],
                 &tp.t[8], &tp.t[9], &tp.t[10]) != 12)
        continue;
      bool ok = true;
      for (int j = 0; j < 11; ++j) tp.sorted[j] = tp.t[j];
      std::sort(tp.sorted, tp.sorted + 11);
      for (int j = 0; j < 11; ++j) ok = ok && tp.sorted[j] < (uint32_t)n && (j == 0 || tp.sorted[j] != tp.sorted[j - 1]);
      if (!ok) { printf("probe %s bad\n", tag); continue; }
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0));
      CK(hipEventCreate(&e1));
      hipLaunchKernelGGL(k_probe, dim3((unsigned)ntiles), dim3(256), 0, 0, g_st, tp, f);
      float best = 1e9f;
      for (int r = 0; r < g_reps; ++r) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_probe, dim3((unsigned)ntiles), dim3(256), 0, 0, g_st, tp, f);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
      }
      CK(hipGetLastError());
      printf("probe %s", tag);
      for (int j = 0; j < 11; ++j) printf(" %u", tp.t[j]);
      printf(" %.3f\n", best);
      fflush(stdout);
      CK(hipEventDestroy(e0));
      CK(hipEventDestroy(e1));
    }
    CK(hipFree(g_st));
    return 0;
  }
  if (argc > 3 && !strcmp(argv[3], "probe2")) {
    // `<tag> <TB> <s0> ... <s(TB-1)> <d0> ... <d(TB-1)>` lines on stdin -> one timing each (out of place: a second 2^n buffer)
    A* out2 = nullptr;
    CK(hipMalloc(&out2, g_n * sizeof(A)));
    k_init<<<4096, 256>>>(g_st, g_n);
    CK(hipDeviceSynchronize());
    char line[512];
    while (fgets(line, sizeof line, stdin)) {
      char tag[64];
      int TB = 0, used = 0;
      if (sscanf(line, "%63s %d%n", tag, &TB, &used) != 2 || (TB != 10 && TB != 11 && TB != 12)) continue;
      uint32_t v[24];
      const char* p = line + used;
      bool ok = true;
      for (int j = 0; j < 2 * TB && ok; ++j) {
        int adv = 0;
        ok = sscanf(p, "%u%n", &v[j], &adv) == 1;
        p += adv;
      }
      if (!ok) { printf("probe2 %s bad\n", tag); continue; }
      auto timeit = [&](auto tp, int tb) {
        for (int j = 0; j < tb; ++j) { tp.s[j] = v[j]; tp.d[j] = v[tb + j]; tp.ss[j] = v[j]; tp.ds[j] = v[tb + j]; }
        std::sort(tp.ss, tp.ss + tb);
        std::sort(tp.ds, tp.ds + tb);
        for (int j = 0; j < tb; ++j)
          if (tp.ss[j] >= (uint32_t)n || tp.ds[j] >= (uint32_t)n || (j && (tp.ss[j] == tp.ss[j - 1] || tp.ds[j] == tp.ds[j - 1]))) return -1.0f;
        const unsigned blocks = (unsigned)(g_n >> tb);
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        float best = 1e9f;
        for (int r = 0; r <= g_reps; ++r) {
          CK(hipEventRecord(e0));
          if (tb == 10) hipLaunchKernelGGL(k_probe2<10>, dim3(blocks), dim3(256), 0, 0, g_st, out2, *(Tp2<10>*)&tp);
          else if (tb == 11) hipLaunchKernelGGL(k_probe2<11>, dim3(blocks), dim3(256), 0, 0, g_st, out2, *(Tp2<11>*)&tp);
          else hipLaunchKernelGGL(k_probe2<12>, dim3(blocks), dim3(256), 0, 0, g_st, out2, *(Tp2<12>*)&tp);
          CK(hipEventRecord(e1));
          CK(hipEventSynchronize(e1));
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          if (r) best = std::min(best, ms);
        }
        CK(hipGetLastError());
        CK(hipEventDestroy(e0));
        CK(hipEventDestroy(e1));
        return best;
      };
      float ms = -1;
      if (TB == 10) { Tp2<10> tp; ms = timeit(tp, 10); }
      else if (TB == 11) { Tp2<11> tp; ms = timeit(tp, 11); }
      else { Tp2<12> tp; ms = timeit(tp, 12); }
      printf("probe2 %-40s TB=%d %.3f ms %6.0f GB/s |", tag, TB, ms, ms > 0 ? 32.0 * (double)g_n / ms / 1e6 : 0.0);
      for (int j = 0; j < 2 * TB; ++j) printf("%s%u", j == TB ? " -> " : " ", v[j]);
      printf("\n");
      fflush(stdout);
    }
    CK(hipFree(out2));
    CK(hipFree(g_st));
    return 0;
  }
  if (argc > 3 && !strcmp(argv[3], "order")) {
    const uint32_t N = (uint32_t)n;
    struct { std::vector<uint32_t> h; const char* name; } sets[] = {
        {{11, 12, 13, 14, 15}, "11..15"}, {{6, 7, 8, 9, 10}, "6..10"}, {{N - 5, N - 4, N - 3, N - 2, N - 1}, "top5"},
        {{6, 15, 18, 21, 24}, "6,15,18,21,24"}, {{12, 15, 18, 21, 24}, "12,15,18,21,24"}, {{20, 21, 22, 23, 24}, "20..24"},
        {{6, 7, N - 3, N - 2, N - 1}, "6,7,top3"}, {{8, 12, 17, 22, 27}, "8,12,17,22,27"}, {{16, 17, 18, 19, 20}, "16..20"},
        {{11, 12, N - 3, N - 2, N - 1}, "11,12,top3"}};
    for (auto& s : sets) orders(s.h, s.name, n);
    CK(hipFree(g_st));
    return 0;
  }
  variants<1, 0>(top, "bits n-5..n-1");
  variants<1, 0>(low, "bits 6..10");
  variants<1, 0>(mix, "bits 12..24");
  variants<1, 2>(top, "bits n-5..n-1");
  variants<1, 2>(mix, "bits 12..24");
  variants<2, 8>(top, "bits n-5..n-1");
  variants<2, 8>(mix, "bits 12..24");
  variants<2, 32>(top, "bits n-5..n-1");
  variants<2, 64>(top, "bits n-5..n-1");
  CK(hipFree(g_st));
  return 0;
}
