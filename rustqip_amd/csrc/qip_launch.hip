// qip_launch.hip — one launcher per kernel class (qip_kernels.h) and apply_op: the op descriptor -> plan -> launch path.
#include "qip_tile.h"

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
Ins make_ins(std::vector<uint32_t> positions, uint64_t ormask) {
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


// Stream-ordered copy of an op payload into the device arena.  Eagerly the (pageable) source is staged by
// the runtime before the call returns.  While a program records, the bytes go into the host image of the program's
// device pool instead (ProgPool): one upload when the recording is done, nothing from the host at replay.
int arena_upload(qip_hip_state* s, const void* src, size_t bytes, size_t arena_off) {
  if (bytes == 0) return QIP_OK;
  QCHK(ensure_arena(s, arena_off + bytes));
  if (ProgPool* pp = s->capture_pool) {
    if (pp->overflow) return QIP_OK;
    const size_t at = (size_t)((char*)s->arena - (char*)pp->base) + arena_off;
    if (pp->image.size() < at + bytes) pp->image.resize(at + bytes);
    memcpy(pp->image.data() + at, src, bytes);
    return QIP_OK;
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
// `split` (0 = none): an index position that is made thread bit 5 — the upper half of a wave — on BOTH sides (11 for 16-byte
// elements: byte-address bit 15, "split rows"); the tile then holds the R row bits, the destination bits fed by the source's row
// bits, `split` and the destination bit fed by source position `split`: 11 or 12 bits (*tile_bits_out).  Without it: 2 R bits.
static int make_perm_desc(uint32_t n, const uint32_t* pi, uint32_t R, uint32_t fold_bits, uint32_t split, PermDesc* out, uint32_t* tile_bits_out) {
  if (split && (split < R || split >= n || R != 5u)) return fail(QIP_ERR_INVALID, "internal: split rows need R = 5 and a position >= 5 below n");
  PermDesc& d = *out;
  memset(&d, 0, sizeof d);
  std::vector<char> in_tile(n, 0);
  for (uint32_t b = 0; b < R; ++b) in_tile[b] = 1;             // the destination's row bits
  for (uint32_t b = 0; b < n; ++b)
    if (pi[b] < R) in_tile[b] = 1;                              // destination bits fed by the source's row bits
  if (split) {
    in_tile[split] = 1;
    for (uint32_t b = 0; b < n; ++b)
      if (pi[b] == split) in_tile[b] = 1;
  }
  uint32_t cnt = 0;
  for (uint32_t b = 0; b < n; ++b) cnt += in_tile[b];
  const uint32_t TB = split ? (cnt <= 11u ? 11u : 12u) : 2 * R;
  if (n < TB || TB > (uint32_t)kPermMaxTile || cnt > TB) return fail(QIP_ERR_INVALID, "internal: permutation tile does not fit n = %u", n);
  for (uint32_t b = 0; b < n && cnt < TB; ++b)                  // pad with the lowest positions left
    if (!in_tile[b]) {
      in_tile[b] = 1;
      ++cnt;
    }
  // coordinate order on each side: the rows, then the split position, then the rest ascending
  std::vector<uint32_t> tb, sb, srcs;
  for (uint32_t b = 0; b < R; ++b) tb.push_back(b), sb.push_back(b);
  if (split) tb.push_back(split), sb.push_back(split);
  for (uint32_t b = 0; b < n; ++b)
    if (in_tile[b]) {
      srcs.push_back(pi[b]);
      if (b >= R && b != split) tb.push_back(b);
    }
  std::sort(srcs.begin(), srcs.end());
  for (uint32_t sp : srcs)
    if (sp >= R && sp != split) sb.push_back(sp);
  if (tb.size() != TB || sb.size() != TB) return fail(QIP_ERR_INVALID, "internal: permutation tile is not closed");
  std::vector<uint32_t> ts = tb;
  std::sort(ts.begin(), ts.end());
  for (uint32_t i = 0; i < TB; ++i) {
    d.tbits[i] = tb[i];
    d.sbits[i] = sb[i];
    d.tsorted[i] = ts[i];
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
  if (tile_bits_out) *tile_bits_out = TB;
  return QIP_OK;
}

// The pair form for 8-byte elements whose index bit 0 moves (k_permute_pairs): tile = positions {0..5, 12} on both sides, 12 or 13
// bits; *fits = false when the permutation needs 14.  Coordinate order on each side: position 0 (the pair bit), 1..5, 12, the rest
// ascending.  Folds: the source-side lane bits (coordinates 1..5) that land on tile bits above 5 are XOR-ed into free slot bits
// 1..5 (never bit 0: a stored pair is one aligned 16-byte LDS read), so that a wave's 8-byte writes spread over the banks.
static int make_perm_pairs_desc(uint32_t n, const uint32_t* pi, PermDesc* out, uint32_t* tile_bits_out, bool* fits) {
  *fits = false;
  if (n < 14u) return QIP_OK;
  PermDesc& d = *out;
  memset(&d, 0, sizeof d);
  auto special = [](uint32_t p) { return p <= 5u || p == 12u; };
  std::vector<char> in_tile(n, 0);
  for (uint32_t b = 0; b < n; ++b)
    if (special(b) || special(pi[b])) in_tile[b] = 1;
  uint32_t cnt = 0;
  for (uint32_t b = 0; b < n; ++b) cnt += in_tile[b];
  if (cnt > 13u) return QIP_OK;
  const uint32_t TB = cnt <= 12u ? 12u : 13u;
  for (uint32_t b = 0; b < n && cnt < TB; ++b)  // pad with the lowest positions left
    if (!in_tile[b]) {
      in_tile[b] = 1;
      ++cnt;
    }
  std::vector<uint32_t> tb = {0, 1, 2, 3, 4, 5, 12}, sb = tb, srcs;
  for (uint32_t b = 0; b < n; ++b)
    if (in_tile[b]) {
      srcs.push_back(pi[b]);
      if (!special(b)) tb.push_back(b);
    }
  std::sort(srcs.begin(), srcs.end());
  for (uint32_t sp : srcs)
    if (!special(sp)) sb.push_back(sp);
  if (tb.size() != TB || sb.size() != TB) return fail(QIP_ERR_INVALID, "internal: permutation tile is not closed");
  std::vector<uint32_t> ts = tb;
  std::sort(ts.begin(), ts.end());
  for (uint32_t i = 0; i < TB; ++i) {
    d.tbits[i] = tb[i];
    d.sbits[i] = sb[i];
    d.tsorted[i] = ts[i];
  }
  for (uint32_t i = 0; i < TB; ++i) {
    uint32_t k = 0;
    while (k < TB && pi[tb[k]] != sb[i]) ++k;
    if (k == TB) return fail(QIP_ERR_INVALID, "internal: permutation tile is not closed");
    d.u2c[i] = k;
  }
  char taken[6] = {1, 0, 0, 0, 0, 0};  // slot bits 0..5; bit 0 is never a target
  for (uint32_t i = 1; i <= 5; ++i)
    if (d.u2c[i] >= 1 && d.u2c[i] <= 5) taken[d.u2c[i]] = 1;
  uint32_t slot = 1;
  for (uint32_t i = 1; i <= 5; ++i) {
    if (d.u2c[i] <= 5) continue;  // (0: the lanes already differ in the slot's bit 0; 1..5: in place)
    while (slot <= 5 && taken[slot]) ++slot;
    if (slot > 5) break;
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
  *tile_bits_out = TB;
  *fits = true;
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
int launch_permute(qip_hip_state* s, const uint32_t* pi_in) {
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
  // 16-byte elements: 512-byte rows with SPLIT wave accesses — thread bit 5 is index position 11 on both sides, so every
  // wave-level read and write is two 512-byte halves 32 KiB apart (r6; measured on every permutation of tools/bench_permute.py:
  // profiles/r06_permute.md) — in tiles of 2^11 or 2^12 elements (32 / 64 KiB of LDS).  A state below 2^12 elements keeps
  // contiguous rows in 2^10-element tiles.  8-byte elements (Complex<f32> whose index bit 0 moves): 16-byte PAIRS on both global
  // sides with the same split and the 8-byte transposition inside LDS (k_permute_pairs) when positions {0..5, 12} of both sides fit
  // 13 tile bits, else 512-byte rows of 64 elements (the same split for 8-byte accesses — two 256-byte pieces — measured slower).
  const uint32_t R = wide ? 5u : 6u;
  const uint32_t split = (wide && n >= 12u) ? 11u : 0u;
  ProfRec rec;
  if (s->profile) QCHK(prof_begin(s, KC_PERMUTE, 2.0 * (double)s->amp_bytes * (double)s->namps, &rec));
  const bool nt = use_nt(s);
  PermDesc dp;
  bool pairs_fit = false;
  uint32_t pairs_tb = 0;
  if (!wide) QCHK(make_perm_pairs_desc(n, pi.data(), &dp, &pairs_tb, &pairs_fit));  // (n >= 14 and at most 13 tile bits)
  if (n < (split ? 12u : 2 * R)) {
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
  } else if (pairs_fit) {  // 8-byte elements, index bit 0 moves, the pair tile fits: 16-byte pairs on both global sides
    const dim3 block(kBlock);
    const f32x4* src = (const f32x4*)s->cur;
    f32x4* dst = (f32x4*)s->alt;
    if (pairs_tb == 12) {
      const dim3 grid = grid2d(1ull << (n - 12), 1);
      if (nt) hipLaunchKernelGGL((k_permute_pairs<12, true, 1>), grid, block, 0, s->stream, src, dst, dp);
      else hipLaunchKernelGGL((k_permute_pairs<12, false, 1>), grid, block, 0, s->stream, src, dst, dp);
    } else {  // 64 KiB of LDS: two tiles per block, the second in flight (as k_permute_bits does for its 2^12 tiles)
      const dim3 grid = grid2d((1ull << (n - 13)) / 2, 1);
      if (nt) hipLaunchKernelGGL((k_permute_pairs<13, true, 2>), grid, block, 0, s->stream, src, dst, dp);
      else hipLaunchKernelGGL((k_permute_pairs<13, false, 2>), grid, block, 0, s->stream, src, dst, dp);
    }
  } else {
    PermDesc d;
    uint32_t TB = 0;
    QCHK(make_perm_desc(n, pi.data(), R, wide ? 3u : 4u, split, &d, &TB));  // (fold width follows the element size, not R)
    // a 2^12-element tile of 16-byte elements is 64 KiB of LDS = two blocks per CU: such a block moves TWO tiles, the second one's
    // loads in flight while the first is stored (three transpositions 5.99 -> 5.74 ms, random 5.75 -> 5.55; no gain for 2^11 tiles)
    const uint32_t pipe = (wide && TB == 12 && n >= TB + 1) ? 2u : 1u;
    const dim3 grid = grid2d((1ull << (n - TB)) / pipe, 1), block(kBlock);
#define PB2(A, RR, TT, PP)                                                                                                         \
  do {                                                                                                                             \
    if (nt) hipLaunchKernelGGL((k_permute_bits<A, RR, TT, true, PP>), grid, block, 0, s->stream, (const A*)s->cur, (A*)s->alt, d);  \
    else hipLaunchKernelGGL((k_permute_bits<A, RR, TT, false, PP>), grid, block, 0, s->stream, (const A*)s->cur, (A*)s->alt, d);    \
  } while (0)
    if (s->dtype == QIP_C64 && TB == 10) PB2(amp_t<double>, 5, 10, 1);
    else if (s->dtype == QIP_C64 && TB == 11) PB2(amp_t<double>, 5, 11, 1);
    else if (s->dtype == QIP_C64 && pipe == 2) PB2(amp_t<double>, 5, 12, 2);
    else if (s->dtype == QIP_C64) PB2(amp_t<double>, 5, 12, 1);
    else if (packed && TB == 10) PB2(f32x4, 5, 10, 1);
    else if (packed && TB == 11) PB2(f32x4, 5, 11, 1);
    else if (packed && pipe == 2) PB2(f32x4, 5, 12, 2);
    else if (packed) PB2(f32x4, 5, 12, 1);
    else PB2(amp_t<float>, 6, 12, 1);
#undef PB2
  }
  HIPCHK(hipGetLastError());
  if (s->profile) QCHK(prof_end(s, &rec));
  std::swap(s->cur, s->alt);
  std::swap(s->owns_cur, s->owns_alt);
  return QIP_OK;
}

int state_settle(qip_hip_state* s) {
  if (s->layout.empty()) return QIP_OK;
  const std::vector<uint32_t> phys = s->layout;  // final index bit d takes the bit that lives on phys[d] now
  s->layout.clear();
  const int rc = launch_permute(s, phys.data());
  if (rc != QIP_OK) s->layout = phys;
  return rc;
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
    // an exported entry point: the two shape arguments index fixed-size tables of the descriptor, so they are checked here
    // (the library itself only ever passes 5 / 6 and 3 / 4)
    if (row_bits == 100) {  // the pair form of 8-byte elements (k_permute_pairs); "fits": false when the tile would need 14 bits
      PermDesc d;
      uint32_t TB = 0;
      bool fits = false;
      if (make_perm_pairs_desc(n, pi, &d, &TB, &fits) != QIP_OK) return nullptr;
      if (!fits) return (json = "{\"fits\":false}").c_str();
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
      json = "{\"fits\":true,\"n\":" + std::to_string(n) + ",\"tile_bits\":" + std::to_string(TB) + "," + arr("tsorted", d.tsorted, TB) + "," +
             arr("tbits", d.tbits, TB) + "," + arr("sbits", d.sbits, TB) + "," + arr("u2c", d.u2c, TB) + "," + arr("fold_from", d.fold_from, d.nfold) + "," +
             arr("fold_to", d.fold_to, d.nfold) + "," + arr("outer_dst", od, d.n_outer) + "," + arr("outer_src", os, d.n_outer) + "}";
      return json.c_str();
    }
    // row_bits = 0: the shape the library picks for 16-byte elements (512-byte rows, split at position 11 when n >= 12)
    const uint32_t split = (row_bits == 0 && n >= 12) ? 11u : 0u;
    if (row_bits == 0) row_bits = 5;
    if (row_bits < 1 || 2 * row_bits > (uint32_t)kPermMaxTile) return fail(QIP_ERR_INVALID, "row_bits must be 0 (the library's choice) or 1..%d", kPermMaxTile / 2), nullptr;
    if (fold_bits > 4 || fold_bits > 2 * row_bits) return fail(QIP_ERR_INVALID, "fold_bits must be <= min(4, 2 * row_bits)"), nullptr;
    PermDesc d;
    uint32_t TB = 0;
    if (make_perm_desc(n, pi, row_bits, fold_bits, split, &d, &TB) != QIP_OK) return nullptr;
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
    json = "{\"n\":" + std::to_string(n) + ",\"row_bits\":" + std::to_string(row_bits) + ",\"tile_bits\":" + std::to_string(TB) + ",\"split\":" + std::to_string(split) + "," +
           arr("tsorted", d.tsorted, TB) + "," + arr("tbits", d.tbits, TB) + "," +
           arr("sbits", d.sbits, TB) + "," + arr("u2c", d.u2c, TB) + "," + arr("fold_from", d.fold_from, d.nfold) + "," +
           arr("fold_to", d.fold_to, d.nfold) + "," + arr("outer_dst", od, d.n_outer) + "," + arr("outer_src", os, d.n_outer) + "}";
    return json.c_str();
  } catch (const std::exception& e) {
    fail(QIP_ERR_INVALID, "internal error: %s", e.what());
    return nullptr;
  }
}

template <typename T>
int launch_gather(qip_hip_state* s, const FlatOp& f, const amp_t<T>* in, uint64_t in_len,
                         amp_t<T>* out, uint64_t out_len, uint64_t in_off, uint64_t out_off,
                         int accumulate);

// A operand of k_gate_kq_mfma, one double per (tile row block, K-step, lane); see the kernel header.
// f32_layout: the C/D rows of v_mfma_f32_16x16x4_f32 are 4 * (lane >> 4) + reg, those of the f64 form (lane >> 4) + 4 * reg.
static void build_afrag(const Plan& p, const std::vector<uint32_t>& tau, std::vector<double>* out, bool f32_layout = false) {
  const uint32_t k = (uint32_t)p.opos.size();
  const uint32_t S = 1u << k, TT = S / 8, KS = S / 2;
  uint32_t perm_bit[12];  // c~ bit b (b-th lowest target position) -> sub-index bit of the reference
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

// A operands of mfma3_item (qip_kernels.h; k >= 5): 16 COMPLEX rows per block, parts P = G_re, R = -(G_re + G_im), Q = G_im - G_re
// (computed in double, rounded once to the state's precision by the caller); a matrix without an imaginary part gets P only
// (*real_only).  Index ((rb * NP + part) * (S/4) + s) * 64 + lane; lane l = (row i = l & 15 of the block, column kk = l >> 4 of the
// K-step); C row i belongs to lane q = i & 3 as its reg = i >> 2 (f64 layout) or to lane q = i >> 2 as reg = i & 3 (f32 layout),
// i.e. to the amplitude c~' = 16 rb + 4 reg + q; column c~ = 4 s + kk.
static void build_afrag3(const Plan& p, const std::vector<uint32_t>& tau, std::vector<double>* out, bool f32_layout, bool* real_only) {
  const uint32_t k = (uint32_t)p.opos.size();
  const uint32_t S = 1u << k, RB = S / 16, KS3 = S / 4;
  uint32_t perm_bit[12];  // c~ bit b (b-th lowest target position) -> sub-index bit of the reference
  for (uint32_t b = 0; b < k; ++b)
    for (uint32_t j = 0; j < k; ++j)
      if (p.opos[j] == tau[b]) perm_bit[b] = k - 1 - j;
  auto c_of = [&](uint32_t ct) {
    uint32_t c = 0;
    for (uint32_t b = 0; b < k; ++b) c |= ((ct >> b) & 1u) << perm_bit[b];
    return c;
  };
  bool real = true;
  for (size_t e = 0; e < (size_t)S * S && real; ++e) real = p.table[2 * e + 1] == 0.0;
  *real_only = real;
  const uint32_t NP = real ? 1u : 3u;
  out->assign((size_t)RB * NP * KS3 * 64, 0.0);
  for (uint32_t rb = 0; rb < RB; ++rb)
    for (uint32_t s = 0; s < KS3; ++s)
      for (uint32_t l = 0; l < 64; ++l) {
        const uint32_t i = l & 15, kk = l >> 4;
        const uint32_t qp = f32_layout ? i >> 2 : i & 3, reg = f32_layout ? i & 3 : i >> 2;
        const uint32_t ctp = 16 * rb + 4 * reg + qp, ct = 4 * s + kk;
        const size_t e = (size_t)c_of(ctp) * S + c_of(ct);
        const double re = p.table[2 * e], im = p.table[2 * e + 1];
        (*out)[(((size_t)rb * NP + 0) * KS3 + s) * 64 + l] = re;
        if (!real) {
          (*out)[(((size_t)rb * NP + 1) * KS3 + s) * 64 + l] = -(re + im);
          (*out)[(((size_t)rb * NP + 2) * KS3 + s) * 64 + l] = im - re;
        }
      }
}

// dense k = 4, 5 through the LDS-staged matrix-core kernel (k_gate_tile_mfma).  *done = false when the op does not qualify:
// controls inside a row, a state below one tile.
template <typename T>
static int launch_tile_mfma(qip_hip_state* s, const Plan& p, amp_t<T>* st, bool* done) {
  *done = false;
  const uint32_t k = (uint32_t)p.opos.size();
  if ((k != 4 && k != 5) || s->n < (uint32_t)kTileBits + 6) return QIP_OK;
  // k = 5: Complex<f64> only (Complex<f32> measured even with the direct kernel: 3.41 vs 3.47 ms at n = 30), and only where a block
  // gets at least 8 tiles with two blocks on every CU: 2^12 tiles (n >= 23 + controls); below that the direct kernel is faster
  if (k == 5 && (!std::is_same<T, double>::value || s->n < (uint32_t)kTileBits + 12 + (uint32_t)p.cpos.size())) return QIP_OK;
  const uint32_t p5 = tile_p5_of<T>(s->n);  // the rows' sixth bit (qip_tile.h): 11 = split rows
  for (uint32_t c : p.cpos)
    if (tile_is_low(c, p5)) return QIP_OK;
  std::vector<uint32_t> tau = p.opos;
  std::sort(tau.begin(), tau.end());
  std::vector<uint32_t> high;  // the tile's five positions above the rows: the targets there, then free ones from 11 upwards
  for (uint32_t t : tau)
    if (!tile_is_low(t, p5)) high.push_back(t);
  auto taken = [&](uint32_t pp) {
    return tile_is_low(pp, p5) || std::find(high.begin(), high.end(), pp) != high.end() || std::find(p.cpos.begin(), p.cpos.end(), pp) != p.cpos.end();
  };
  if (p5 == 5u && std::find(high.begin(), high.end(), 6u) != high.end() && !taken(7u) && high.size() < (size_t)kTileHigh) high.push_back(7u);
  for (uint32_t pp = 11; high.size() < (size_t)kTileHigh && pp < s->n; ++pp)
    if (!taken(pp)) high.push_back(pp);
  for (uint32_t pp = 5; high.size() < (size_t)kTileHigh && pp < s->n; ++pp)
    if (!taken(pp)) high.push_back(pp);
  if (high.size() != (size_t)kTileHigh) return QIP_OK;
  std::sort(high.begin(), high.end());
  TileMfmaDesc d;
  memset(&d, 0, sizeof d);
  for (int jx = 0; jx < kTileHigh; ++jx) d.hpos[jx] = high[jx];
  d.p5 = p5;
  auto tile_bit = [&](uint32_t pos) { return tile_is_low(pos, p5) ? tile_low_bit(pos) : (uint32_t)kTileLow + (uint32_t)(std::find(high.begin(), high.end(), pos) - high.begin()); };
  uint32_t is_target = 0;
  for (uint32_t b = 0; b < k; ++b) {
    d.tb[b] = tile_bit(tau[b]);
    is_target |= 1u << d.tb[b];
  }
  int nn = 0;
  for (uint32_t b = 0; b < (uint32_t)kTileBits; ++b)
    if (!((is_target >> b) & 1u)) d.nb[nn++] = b;
  std::vector<double> afrag;
  // k = 5: a matrix without an imaginary part takes the two-product form (mfma3_item, NP = 1: half the matrix instructions, a 0/1
  // matrix stays exact); a complex one keeps the four-product real form — the three-product form measured no faster here
  // (5.62 vs 5.63 ms at n = 30: the pipelined tile kernel is no longer bound by the matrix pipe) and rounds worse
  bool real_only = false;
  if (k == 5) build_afrag3(p, tau, &afrag, std::is_same<T, float>::value, &real_only);
  if (!real_only) build_afrag(p, tau, &afrag, std::is_same<T, float>::value);
  std::vector<T> af_t(afrag.begin(), afrag.end());
  QCHK(arena_upload(s, af_t.data(), af_t.size() * sizeof(T), 0));
  // (tile_block_base: positions in the space where p5 and 5 have traded places; the kernel exchanges the two bits back)
  std::vector<uint32_t> opened = high, ctl = p.cpos;
  for (uint32_t& c : ctl)
    if (c == 5u) c = p5;
  for (uint32_t c : ctl) opened.push_back(c);
  for (uint32_t& o : opened)
    if (o == 5u) o = p5;
  Ins ins = make_ins(opened, mask_of(ctl));
  const uint64_t ntiles = 1ull << (s->n - (uint32_t)kTileBits - (uint32_t)p.cpos.size());
  const size_t lds = sizeof(amp_t<T>) << kTileBits;
  const T* af = (const T*)s->arena;
  const bool nt = use_nt(s);
  // k = 5: a block walks `pipe` consecutive tiles (the next tile's rows in flight during the matrix instructions): at least 8
  // tiles each, 512 - 2048 blocks (two per CU, one to four rounds); n = 30: 256 tiles per block
  const uint32_t pipe = k == 5 ? (uint32_t)std::min<uint64_t>(256, std::max<uint64_t>(8, ntiles / 1024)) : 1u;  // (powers of two)
#define TM(KK, LL, PP)                                                                                                                                       \
  do {                                                                                                                                                       \
    if (nt) hipLaunchKernelGGL((k_gate_tile_mfma<T, KK, true, LL, PP>), grid2d(ntiles / pipe, 1), dim3(kTileBlock), lds, s->stream, st, ins, d, af, pipe);    \
    else hipLaunchKernelGGL((k_gate_tile_mfma<T, KK, false, LL, PP>), grid2d(ntiles / pipe, 1), dim3(kTileBlock), lds, s->stream, st, ins, d, af, pipe);      \
  } while (0)
  if (k == 4) TM(4, false, 0);
  else if constexpr (std::is_same<T, double>::value) {
    if (real_only) TM(5, true, 1);
    else TM(5, true, 0);
  } else return fail(QIP_ERR_UNSUPPORTED, "internal: the tile form of dense k = 5 is a Complex<f64> kernel");
#undef TM
  HIPCHK(hipGetLastError());
  *done = true;
  return QIP_OK;
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
  bool real_only = false;
  build_afrag3(p, tau, &afrag, std::is_same<T, float>::value, &real_only);
  std::vector<T> af_t(afrag.begin(), afrag.end());
  QCHK(ensure_arena(s, af_t.size() * sizeof(T)));
  QCHK(arena_upload(s, af_t.data(), af_t.size() * sizeof(T), 0));
  std::vector<uint32_t> pos = p.cpos;
  for (uint32_t t : p.opos) pos.push_back(t);
  Ins ins = make_ins(pos, mask_of(p.cpos));
  MfmaDesc d;
  memset(&d, 0, sizeof d);
  for (uint32_t b = 0; b < k; ++b) d.tau[b] = tau[b];
  const uint64_t nitems = 1ull << (s->n - (uint32_t)pos.size() - 4);  // waves' worth of 16 groups
  // resident blocks per CU = the kernel's launch bound: two waves per SIMD, except Complex<f64> at k = 8 (X alone is 256 registers per lane)
  const unsigned per_cu = (k <= 7 || std::is_same<T, float>::value) ? 2u : 1u;
  const unsigned blocks = (unsigned)std::min<uint64_t>((nitems + 3) / 4, 256ull * per_cu);
  const dim3 grid(blocks), block(kBlock);
  const T* af = (const T*)s->arena;
  const bool nt = use_nt(s);
#define BM(K)                                                                                                                    \
  do {                                                                                                                           \
    if (real_only) {                                                                                                             \
      if (nt) hipLaunchKernelGGL((k_gate_big_mfma<T, K, true, 1>), grid, block, 0, s->stream, st, nitems, ins, d, af);           \
      else hipLaunchKernelGGL((k_gate_big_mfma<T, K, false, 1>), grid, block, 0, s->stream, st, nitems, ins, d, af);             \
    } else {                                                                                                                     \
      if (nt) hipLaunchKernelGGL((k_gate_big_mfma<T, K, true, 3>), grid, block, 0, s->stream, st, nitems, ins, d, af);           \
      else hipLaunchKernelGGL((k_gate_big_mfma<T, K, false, 3>), grid, block, 0, s->stream, st, nitems, ins, d, af);             \
    }                                                                                                                            \
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

// dense k = 9, 10 on the matrix cores: X in LDS, the A operand streamed from L2 in pairs of K-steps (k_gate_huge_mfma)
template <typename T>
static int launch_huge_mfma(qip_hip_state* s, const Plan& p, amp_t<T>* st) {
  const uint32_t k = (uint32_t)p.opos.size();
  std::vector<uint32_t> tau = p.opos;
  std::sort(tau.begin(), tau.end());
  std::vector<double> afrag;
  bool real_only = false;
  build_afrag3(p, tau, &afrag, std::is_same<T, float>::value, &real_only);
  // pairs of consecutive K-steps side by side, so a lane fetches both with one 16-byte (f32: 8-byte) load; the parts of one pair adjacent
  const size_t S = (size_t)1 << k, RB = S / 16, KS3 = S / 4, KP = S / 8, NP = real_only ? 1 : 3;
  std::vector<T> a2(afrag.size());
  for (size_t rb = 0; rb < RB; ++rb)
    for (size_t sp = 0; sp < KP; ++sp)
      for (size_t pt = 0; pt < NP; ++pt)
        for (size_t l = 0; l < 64; ++l)
          for (size_t e = 0; e < 2; ++e) a2[((((rb * KP + sp) * NP + pt) * 64 + l) * 2) + e] = (T)afrag[((rb * NP + pt) * KS3 + 2 * sp + e) * 64 + l];
  QCHK(ensure_arena(s, a2.size() * sizeof(T)));
  QCHK(arena_upload(s, a2.data(), a2.size() * sizeof(T), 0));
  if (!s->capture_pool) HIPCHK(hipStreamSynchronize(s->stream));  // (the staging vector dies with this frame; 8 - 32 MiB once per gate)
  std::vector<uint32_t> pos = p.cpos;
  for (uint32_t t : p.opos) pos.push_back(t);
  Ins ins = make_ins(pos, mask_of(p.cpos));
  HugeDesc d;
  memset(&d, 0, sizeof d);
  for (uint32_t b = 0; b < k; ++b) d.tau[b] = tau[b];
  const uint64_t groups = 1ull << (s->n - (uint32_t)pos.size());
  const uint64_t nitems = groups / 16;
  const unsigned blocks = (unsigned)std::min<uint64_t>(nitems, (uint64_t)s->num_cus);  // one 128-KiB block per CU
  const T* af = (const T*)s->arena;
  const bool nt = use_nt(s);
#define HM(K, NPH)                                                                                                                       \
  do {                                                                                                                                   \
    if (real_only) {                                                                                                                     \
      if (nt) hipLaunchKernelGGL((k_gate_huge_mfma<T, K, NPH, true, 1>), dim3(blocks), dim3(512), 0, s->stream, st, nitems, ins, d, af);  \
      else hipLaunchKernelGGL((k_gate_huge_mfma<T, K, NPH, false, 1>), dim3(blocks), dim3(512), 0, s->stream, st, nitems, ins, d, af);    \
    } else {                                                                                                                             \
      if (nt) hipLaunchKernelGGL((k_gate_huge_mfma<T, K, NPH, true, 3>), dim3(blocks), dim3(512), 0, s->stream, st, nitems, ins, d, af);  \
      else hipLaunchKernelGGL((k_gate_huge_mfma<T, K, NPH, false, 3>), dim3(blocks), dim3(512), 0, s->stream, st, nitems, ins, d, af);    \
    }                                                                                                                                    \
  } while (0)
  // (16 groups x 2^10 Complex<f64> amplitudes are 256 KiB: two phases of 128 KiB)
  if (k == 9) HM(9, 1);
  else if (k == 10) {
    if constexpr (std::is_same<T, double>::value) HM(10, 2);
    else HM(10, 1);
  } else return fail(QIP_ERR_UNSUPPORTED, "L2-streamed matrix-core kernel for k = %u", k);
#undef HM
  HIPCHK(hipGetLastError());
  return QIP_OK;
}

// dense k = 5..10 on a state with fewer than 16 groups (n < k + controls + 4): one block per 16 rows of a group (k_dense_small),
// cur -> alt.  *done = false when the op does not qualify.
template <typename T>
static int launch_dense_small(qip_hip_state* s, const Plan& p, bool* done) {
  *done = false;
  const uint32_t k = (uint32_t)p.opos.size();
  if (k < 5 || k > 10 || p.table.size() != ((size_t)2 << (2 * k))) return QIP_OK;
  QCHK(ensure_alt(s));
  const size_t S = (size_t)1 << k;
  std::vector<amp_t<T>> mt(S * S);  // transposed: mt[c * S + row]
  for (size_t row = 0; row < S; ++row)
    for (size_t c = 0; c < S; ++c) mt[c * S + row] = mk<T>(p.table[2 * (row * S + c)], p.table[2 * (row * S + c) + 1]);
  QCHK(arena_upload(s, mt.data(), mt.size() * sizeof(amp_t<T>), 0));
  if (!s->capture_pool && mt.size() * sizeof(amp_t<T>) > (1u << 20)) HIPCHK(hipStreamSynchronize(s->stream));  // (pageable staging of a large table dies with this frame)
  DenseSmallDesc d;
  memset(&d, 0, sizeof d);
  d.k = k;
  for (uint32_t b = 0; b < k; ++b) d.tpos[b] = p.opos[k - 1 - b];
  std::vector<uint32_t> pos = p.cpos;
  for (uint32_t t : p.opos) pos.push_back(t);
  const Ins ins = make_ins(pos, mask_of(p.cpos));
  const uint32_t groups = 1u << (s->n - (uint32_t)pos.size());
  // rows the controls leave alone are the identity: they travel to the second buffer as they are
  if (!p.cpos.empty()) HIPCHK(hipMemcpyAsync(s->alt, s->cur, s->namps * s->amp_bytes, hipMemcpyDeviceToDevice, s->stream));
  const size_t lds = (S + 256) * sizeof(amp_t<T>);
  hipLaunchKernelGGL((k_dense_small<T>), dim3((unsigned)(S / 16), groups), dim3(kBlock), lds, s->stream, (const amp_t<T>*)s->cur, (amp_t<T>*)s->alt, ins, d,
                     (const amp_t<T>*)s->arena);
  HIPCHK(hipGetLastError());
  std::swap(s->cur, s->alt);  // builder.rs:514
  std::swap(s->owns_cur, s->owns_alt);
  *done = true;
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
    // r4: k = 4 always where the LDS-staged form can run (a state of >= 17 qubits): the bit-exact VALU fold of a dense 16 x 16
    // gate is 128 unfused f64 operations per amplitude = 3.5 ms of vector issue at n = 30 on top of the HBM time (65.8 % whatever
    // the targets), the matrix-core form through the tile streams whole (split) rows: 75 - 81 % (profiles/r04_ops_table.md).
    // Option mfma = 0 keeps the bit-exact VALU form.
    const bool want_mfma = k == 5 || (k >= 3 && low_targets >= 2) || s->mfma == 2 ||  // 2 = force (tuning aid)
                           (k == 4 && s->n >= (uint32_t)kTileBits + 6 && !g_force_k4_direct);
    if (s->mfma && want_mfma && k >= 3 && k <= kMaxMfmaK && s->n >= used + 4) {
      *actual_cls = KC_GATE_KQ_MFMA;
      if ((k == 4 || k == 5) && s->unroll == 0 && !g_force_k4_direct) {  // operands through an LDS-resident tile: whole rows on both global sides
        bool done = false;
        QCHK(launch_tile_mfma<T>(s, p, st, &done));
        if (done) return QIP_OK;
      }
      return launch_kq_mfma<T>(s, p, st);
    }
  }
  if (s->mfma && k > kMaxMfmaK && k <= kMaxBigK && s->n >= used + 4) {
    *actual_cls = KC_GATE_KQ_BIG;
    return launch_big_mfma<T>(s, p, st);
  }
  if (s->mfma && k > kMaxBigK && k <= kMaxHugeK && s->n >= used + 4) {
    *actual_cls = KC_GATE_KQ_BIG;
    return launch_huge_mfma<T>(s, p, st);
  }
  if (k > kMaxRegK && s->mfma && f.distinct) {  // a state too small for the matrix-core kernels: the small dense kernel (1e-12 bar, like them)
    bool done = false;
    QCHK(launch_dense_small<T>(s, p, &done));
    if (done) {
      *actual_cls = KC_DENSE_SMALL;
      return QIP_OK;
    }
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
    case 3: KQ(3, 1); break;  // (r4: the two-groups-per-lane form spilled 344 SGPRs and was a tuning aid only — dropped)
    case 4: KQ(4, 1); break;
    default: return fail(QIP_ERR_UNSUPPORTED, "register kernel for k = %u", k);
  }
#undef KQ
  HIPCHK(hipGetLastError());
  return QIP_OK;
}

// Ship the inner op's payload to the arena and run the literal gather kernel in -> out.
template <typename T>
int launch_gather(qip_hip_state* s, const FlatOp& f, const amp_t<T>* in, uint64_t in_len,
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

// SparseMatrix on k >= 6 distinct qubits with at most four entries per row (optionally controlled): cur -> alt through
// k_sparse_ell.  *done = false when the op does not qualify (more entries per row, too large a table): the literal kernel runs.
constexpr uint32_t kMaxEllK = 18;
template <typename T>
static int launch_sparse_ell(qip_hip_state* s, const Plan& p, const FlatOp& f, bool* done) {
  *done = false;
  const uint32_t k = f.n_op;
  if (k > kMaxEllK || s->namps < ((uint64_t)4 << kStrideShift)) return QIP_OK;
  const uint64_t rows = 1ull << k;
  const uint64_t* rp = f.inner->sparse_rowptr;
  uint64_t widest = 0;
  for (uint64_t r = 0; r < rows; ++r) widest = std::max<uint64_t>(widest, rp[r + 1] - rp[r]);
  if (widest > 4) return QIP_OK;
  const uint32_t E = widest <= 1 ? 1u : (widest <= 2 ? 2u : 4u);
  // the table is indexed by the sub-index in POSITION order (bit b of m' = the op's b-th lowest index position): which
  // stored row that is, and where a stored column's bits go, are host arithmetic
  std::vector<uint32_t> order(k);  // order[b] = j: op index j sits on the b-th lowest position
  for (uint32_t j = 0; j < k; ++j) order[j] = j;
  std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return p.opos[a] < p.opos[b]; });
  EllDesc d;
  memset(&d, 0, sizeof d);
  for (uint32_t c : p.cpos) d.cmask |= 1ull << c;
  for (uint32_t t : p.opos) d.opmask |= 1ull << t;
  for (uint32_t b = 0; b < k;) {
    uint32_t e = b + 1;
    while (e < k && p.opos[order[e]] == p.opos[order[e - 1]] + 1) ++e;
    d.lo[d.nruns] = p.opos[order[b]];
    d.len[d.nruns] = e - b;
    d.shift[d.nruns] = b;
    d.nruns += 1;
    b = e;
  }
  std::vector<uint32_t> nnz(rows);
  std::vector<uint64_t> off(rows * E, 0);
  std::vector<amp_t<T>> val(rows * E, mk<T>(0, 0));
  const amp_t<T>* sv = (const amp_t<T>*)f.inner->sparse_vals;
  for (uint64_t mp = 0; mp < rows; ++mp) {
    uint64_t m = 0;  // stored row: op index j is sub-index bit k-1-j (matrix_ops.rs:12-21)
    for (uint32_t b = 0; b < k; ++b) m |= ((mp >> b) & 1ull) << (k - 1 - order[b]);
    nnz[mp] = (uint32_t)(rp[m + 1] - rp[m]);
    for (uint64_t q = rp[m]; q < rp[m + 1]; ++q) {
      const uint64_t c = f.inner->sparse_cols[q];
      uint64_t o = 0;
      for (uint32_t j = 0; j < k; ++j) o |= ((c >> (k - 1 - j)) & 1ull) << p.opos[j];
      off[mp * E + (q - rp[m])] = o;
      val[mp * E + (q - rp[m])] = sv[q];
    }
  }
  const size_t b_nnz = rows * 4, b_off = rows * E * 8, b_val = rows * E * sizeof(amp_t<T>);
  const size_t o_off = (b_nnz + 15) & ~(size_t)15, o_val = (o_off + b_off + 15) & ~(size_t)15;
  QCHK(ensure_alt(s));
  QCHK(ensure_arena(s, o_val + b_val + 16));
  QCHK(arena_upload(s, nnz.data(), b_nnz, 0));
  QCHK(arena_upload(s, off.data(), b_off, o_off));
  QCHK(arena_upload(s, val.data(), b_val, o_val));
  const uint32_t* dn = (const uint32_t*)s->arena;
  const uint64_t* doff = (const uint64_t*)((char*)s->arena + o_off);
  const amp_t<T>* dval = (const amp_t<T>*)((char*)s->arena + o_val);
  constexpr int U = 4;
  const dim3 grid = grid2d(s->namps, kBlock * U);
  const amp_t<T>* in = (const amp_t<T>*)s->cur;
  amp_t<T>* out = (amp_t<T>*)s->alt;
  bool full = true;
  for (uint64_t r = 0; r < rows; ++r) full = full && rp[r + 1] - rp[r] == E;
  const bool ctl = d.cmask != 0;
#define ELL2(EE, NTV, FULLV, CTLV) \
  hipLaunchKernelGGL((k_sparse_ell<T, EE, U, NTV, FULLV, CTLV>), grid, dim3(kBlock), 0, s->stream, in, out, d, dn, doff, dval)
#define ELL(EE)                                                 \
  do {                                                          \
    const bool nt = use_nt(s);                                  \
    if (nt && full && ctl) ELL2(EE, true, true, true);          \
    else if (nt && full) ELL2(EE, true, true, false);           \
    else if (nt && ctl) ELL2(EE, true, false, true);            \
    else if (nt) ELL2(EE, true, false, false);                  \
    else if (full && ctl) ELL2(EE, false, true, true);          \
    else if (full) ELL2(EE, false, true, false);                \
    else if (ctl) ELL2(EE, false, false, true);                 \
    else ELL2(EE, false, false, false);                         \
  } while (0)
  if (E == 1) ELL(1);
  else if (E == 2) ELL(2);
  else ELL(4);
#undef ELL2
#undef ELL
  HIPCHK(hipGetLastError());
  std::swap(s->cur, s->alt);  // builder.rs:514
  std::swap(s->owns_cur, s->owns_alt);
  *done = true;
  return QIP_OK;
}

// SparseMatrix on k >= 6 distinct qubits (optionally controlled), at most four entries per row, whose positions outside the
// wave row number 3..7: in place through k_sparse_tile (the group staged in LDS beside the row).  *done = false: not this shape.
int64_t g_sparse_tile = 1;  // global option "sparse_tile": 0 = always the out-of-place gather (k_sparse_ell), for A/B runs
// the shape test alone (program capture asks it too: an op that takes this form stays in place, so the program can be a graph)
static bool sparse_tile_shape(uint32_t n, int dtype, const Plan& p, const FlatOp& f, SparseTileDesc* dout, std::vector<uint32_t>* hp_out,
                              std::vector<uint32_t>* ctl_out_out, uint32_t* e_out) {
  if (!g_sparse_tile || f.inner->kind != QIP_OP_SPARSE || !f.distinct) return false;
  const uint32_t k = f.n_op;
  if (k < (dtype == QIP_C64 ? 4u : 6u)) return false;  // (Complex<f32>, measured at n = 30: k = 5 one group per lane 67.6 % against 63.2 %; k = 4 even)
  const uint32_t p5 = tile_p5(dtype, n);
  SparseTileDesc d;
  memset(&d, 0, sizeof d);
  d.p5 = p5;
  std::vector<uint32_t> hp, ctl_out;
  for (uint32_t t : p.opos) {
    if (tile_is_low(t, p5)) d.low_op |= 1u << tile_low_bit(t);
    else hp.push_back(t);
  }
  for (uint32_t c : p.cpos) {
    if (tile_is_low(c, p5)) d.low_ctl |= 1u << tile_low_bit(c);
    else ctl_out.push_back(c);
  }
  std::sort(hp.begin(), hp.end());
  d.kh = (uint32_t)hp.size();
  d.nlow = (uint32_t)__builtin_popcount(d.low_op);
  // 2^(6+kh) amplitudes per tile: up to 64 KiB (two blocks per CU) — Complex<f64> kh = 7 is one 128-KiB block of 1024 lanes per CU
  if (d.kh < 3 || d.kh > 7 || n < 6 + d.kh + (uint32_t)ctl_out.size() + 2) return false;
  const uint64_t rows = 1ull << k;
  const uint64_t* rp = f.inner->sparse_rowptr;
  uint64_t widest = 0;
  for (uint64_t r = 0; r < rows; ++r) widest = std::max<uint64_t>(widest, rp[r + 1] - rp[r]);
  if (widest > 4) return false;
  for (uint32_t j = 0; j < d.kh; ++j) d.hpos[j] = hp[j];
  if (dout) *dout = d;
  if (hp_out) *hp_out = hp;
  if (ctl_out_out) *ctl_out_out = ctl_out;
  if (e_out) *e_out = widest <= 1 ? 1u : (widest <= 2 ? 2u : 4u);
  return true;
}
bool sparse_tile_applies(const qip_hip_state* s, const Plan& p, const FlatOp& f) {
  return sparse_tile_shape(s->n, s->dtype, p, f, nullptr, nullptr, nullptr, nullptr);
}

// Everything the host decides about one k_sparse_tile launch — pure host code (qip_hip_debug_sparse_tile serialises it: the CPU
// tests replay it with a numpy model of the kernel against the oracle, tests/test_tile_plan_cpu.py)
template <typename T> struct SparseTilePlan {
  SparseTileDesc d;
  Ins ins;
  uint64_t ntiles = 0;
  uint32_t E = 1, threads = 0;
  std::vector<uint32_t> nnz, slot;
  std::vector<amp_t<T>> val;
};
template <typename T>
static bool build_sparse_tile_plan(uint32_t n, int dtype, const Plan& p, const FlatOp& f, SparseTilePlan<T>* out) {
  const uint32_t k = f.n_op;
  SparseTileDesc d;
  std::vector<uint32_t> hp, ctl_out;
  uint32_t E = 1;
  if (!sparse_tile_shape(n, dtype, p, f, &d, &hp, &ctl_out, &E)) return false;
  const uint32_t p5 = d.p5;
  const uint64_t rows = 1ull << k;
  const uint64_t* rp = f.inner->sparse_rowptr;
  // m' bit b <- op index order[b]: the op's lane bits ascending, then its tile rows ascending
  std::vector<uint32_t> order;
  for (uint32_t b = 0; b < 6; ++b)
    if ((d.low_op >> b) & 1u)
      for (uint32_t j = 0; j < k; ++j)
        if (tile_is_low(p.opos[j], p5) && tile_low_bit(p.opos[j]) == b) order.push_back(j);
  for (uint32_t h : hp)
    for (uint32_t j = 0; j < k; ++j)
      if (p.opos[j] == h) order.push_back(j);
  std::vector<uint32_t> tbit(k);  // where op index j's bit lives in a tile index
  for (uint32_t j = 0; j < k; ++j) {
    if (tile_is_low(p.opos[j], p5)) tbit[j] = tile_low_bit(p.opos[j]);
    else tbit[j] = 6u + (uint32_t)(std::find(hp.begin(), hp.end(), p.opos[j]) - hp.begin());
  }
  std::vector<uint32_t>&nnz = out->nnz, &slot = out->slot;
  std::vector<amp_t<T>>& val = out->val;
  nnz.assign(rows, 0);
  slot.assign(rows * E, 0);
  val.assign(rows * E, mk<T>(0, 0));
  const amp_t<T>* sv = (const amp_t<T>*)f.inner->sparse_vals;
  for (uint64_t mp = 0; mp < rows; ++mp) {
    uint64_t m = 0;  // stored row: op index j is sub-index bit k-1-j (matrix_ops.rs:12-21)
    for (uint32_t b = 0; b < k; ++b) m |= ((mp >> b) & 1ull) << (k - 1 - order[b]);
    nnz[mp] = (uint32_t)(rp[m + 1] - rp[m]);
    for (uint64_t q = rp[m]; q < rp[m + 1]; ++q) {
      const uint64_t c = f.inner->sparse_cols[q];
      uint32_t o = 0;
      for (uint32_t j = 0; j < k; ++j) o |= (uint32_t)((c >> (k - 1 - j)) & 1ull) << tbit[j];
      slot[mp * E + (q - rp[m])] = o;
      val[mp * E + (q - rp[m])] = sv[q];
    }
  }
  // the block base: the tile's high positions and the outside controls opened, the controls reading 1 (tile_block_base works in
  // the space where p5 and 5 have traded places)
  std::vector<uint32_t> opened = hp;
  for (uint32_t c : ctl_out) opened.push_back(c);
  uint64_t ones = 0;
  for (uint32_t& o : opened)
    if (o == 5u) o = p5;
  for (uint32_t c : ctl_out) ones |= 1ull << (c == 5u ? p5 : c);
  out->d = d;
  out->ins = make_ins(opened, ones);
  out->ntiles = 1ull << (n - 6 - d.kh - (uint32_t)ctl_out.size());
  out->threads = 1u << (d.kh + 3);
  out->E = E;
  return true;
}

template <typename T>
static int launch_sparse_tile(qip_hip_state* s, const Plan& p, const FlatOp& f, bool* done) {
  *done = false;
  SparseTilePlan<T> sp;
  if (!build_sparse_tile_plan<T>(s->n, s->dtype, p, f, &sp)) return QIP_OK;
  const SparseTileDesc& d = sp.d;
  const uint32_t E = sp.E;
  const uint64_t rows = 1ull << f.n_op;
  const std::vector<uint32_t>&nnz = sp.nnz, &slot = sp.slot;
  const std::vector<amp_t<T>>& val = sp.val;
  const size_t b_nnz = rows * 4, b_slot = rows * E * 4, b_val = rows * E * sizeof(amp_t<T>);
  const size_t o_slot = (b_nnz + 15) & ~(size_t)15, o_val = (o_slot + b_slot + 15) & ~(size_t)15;
  QCHK(ensure_arena(s, o_val + b_val + 16));
  QCHK(arena_upload(s, nnz.data(), b_nnz, 0));
  QCHK(arena_upload(s, slot.data(), b_slot, o_slot));
  QCHK(arena_upload(s, val.data(), b_val, o_val));
  const uint32_t* dn = (const uint32_t*)s->arena;
  const uint32_t* dslot = (const uint32_t*)((char*)s->arena + o_slot);
  const amp_t<T>* dval = (const amp_t<T>*)((char*)s->arena + o_val);
  const Ins ins = sp.ins;
  const uint64_t ntiles = sp.ntiles;
  const unsigned threads = sp.threads;
  // r6: Complex<f32> rows of four entries read their table from LDS (copied behind the tile by every block) when it is small: a
  // lane's 8 x (1 + 2 E) = 72 table loads per tile through the vector memory path bound that sweep, not HBM (k = 7: 5.98 -> 3.23 ms =
  // 36 -> 67 % of 8 TB/s).  Measured and left alone: Complex<f64> E = 4 (6.81 -> 6.99 ms: its HBM time is twice as long and hides the
  // loads), E = 2 in both precisions (f64 k = 6: 5.69 -> 6.22 ms, f32: 3.74 -> 3.79); profiles/r06_sparse_tile.md
  const size_t tile_bytes = sizeof(amp_t<T>) << (6 + d.kh);
  const size_t table_bytes = (size_t)rows * E * (sizeof(amp_t<T>) + 4) + (size_t)rows * 4;
  const bool tl = E == 4 && std::is_same<T, float>::value && table_bytes <= 16 * 1024 && tile_bytes + table_bytes <= 160 * 1024;
  const size_t lds = tile_bytes + (tl ? table_bytes : 0);
  const dim3 grid = grid2d(ntiles, 1);
  amp_t<T>* st = (amp_t<T>*)s->cur;
  const bool nt = use_nt(s);
#define SPT3(EE, NTV, TLV)                                                                                                                 \
  do {                                                                                                                                      \
    if (lds > 64 * 1024)                                                                                                                    \
      HIPCHK(hipFuncSetAttribute((const void*)k_sparse_tile<T, EE, NTV, TLV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));       \
    hipLaunchKernelGGL((k_sparse_tile<T, EE, NTV, TLV>), grid, dim3(threads), lds, s->stream, st, ntiles, ins, d, dn, dslot, dval, (uint32_t)rows); \
  } while (0)
#define SPT(EE)                        \
  do {                                 \
    if (nt) SPT3(EE, true, false);     \
    else SPT3(EE, false, false);       \
  } while (0)
  if (E == 1) SPT(1);
  else if (E == 2) SPT(2);
  else if (tl) {
    if constexpr (std::is_same<T, float>::value) {
      if (nt) SPT3(4, true, true);
      else SPT3(4, false, true);
    }
  } else SPT(4);
#undef SPT3
#undef SPT
  HIPCHK(hipGetLastError());
  *done = true;
  return QIP_OK;
}

// Host-only test hook (include/qip_hip.h): the k_sparse_tile plan of one op as JSON, "{\"applies\":0}" when the op takes another kernel
template <typename T>
static void sparse_tile_json(uint32_t n, int dtype, const Plan& p, const FlatOp& f, std::string* js) {
  SparseTilePlan<T> sp;
  if (!build_sparse_tile_plan<T>(n, dtype, p, f, &sp)) {
    *js = "{\"applies\":0}";
    return;
  }
  auto list = [&](const char* key, const std::vector<uint64_t>& v) {
    *js += std::string(",\"") + key + "\":[";
    for (size_t i = 0; i < v.size(); ++i) *js += (i ? "," : "") + std::to_string(v[i]);
    *js += "]";
  };
  *js = "{\"applies\":1,\"n\":" + std::to_string(n) + ",\"kh\":" + std::to_string(sp.d.kh) + ",\"p5\":" + std::to_string(sp.d.p5) +
        ",\"low_op\":" + std::to_string(sp.d.low_op) + ",\"nlow\":" + std::to_string(sp.d.nlow) + ",\"low_ctl\":" + std::to_string(sp.d.low_ctl) +
        ",\"E\":" + std::to_string(sp.E) + ",\"threads\":" + std::to_string(sp.threads) + ",\"ntiles\":" + std::to_string(sp.ntiles) +
        ",\"ins_ormask\":" + std::to_string(sp.ins.ormask);
  list("hpos", std::vector<uint64_t>(sp.d.hpos, sp.d.hpos + sp.d.kh));
  list("ins_pos", std::vector<uint64_t>(sp.ins.pos, sp.ins.pos + sp.ins.npos));
  list("nnz", std::vector<uint64_t>(sp.nnz.begin(), sp.nnz.end()));
  list("slot", std::vector<uint64_t>(sp.slot.begin(), sp.slot.end()));
  *js += ",\"val\":[";
  char buf[80];
  for (size_t i = 0; i < sp.val.size(); ++i) {
    snprintf(buf, sizeof buf, "%s[%.17g,%.17g]", i ? "," : "", (double)sp.val[i].x, (double)sp.val[i].y);
    *js += buf;
  }
  *js += "]}";
}
extern "C" const char* qip_hip_debug_sparse_tile(int dtype, uint32_t n, const qip_op* op) {
  static thread_local std::string json;
  try {
    if (!op || (dtype != QIP_C64 && dtype != QIP_C32)) return fail(QIP_ERR_INVALID, "null op or bad dtype"), nullptr;
    FlatOp f;
    if (flatten_op(n, op, false, &f) != QIP_OK) return nullptr;
    Plan p;
    if (make_plan(dtype, n, f, false, &p) != QIP_OK) return nullptr;
    if (dtype == QIP_C64) sparse_tile_json<double>(n, dtype, p, f, &json);
    else sparse_tile_json<float>(n, dtype, p, f, &json);
    return json.c_str();
  } catch (const std::exception& e) {
    fail(QIP_ERR_INVALID, "internal error: %s", e.what());
    return nullptr;
  }
}

template <typename T>
int apply_op_t(qip_hip_state* s, const qip_op* op) {
  if (s->jit_prepare) return QIP_OK;  // compiling a program's segment kernels: single ops have nothing to prepare
  (void)hipGetLastError();  // a stale error of an unrelated earlier call must not be blamed on this launch
  arena_begin_group(s);
  FlatOp f;
  QCHK(flatten_op(s->n, op, false, &f));
  Plan p;
  QCHK(make_plan(s->dtype, s->n, f, s->force_generic || g_force_generic, &p));
  if (p.cls == KC_NOOP) {
    if (s->profile) s->prof_launches[KC_NOOP] += 1;
    return QIP_OK;
  }
  // Uncontrolled dense 2- / 3-qubit gates, and Swap ops with a bit inside a 1-KiB row, run as a one-op TILE SWEEP: it streams
  // whole rows on both sides whatever the target bits are (a lane of k_gate_kq with a target below bit 6 reads 16-byte pieces
  // 64 bytes apart: 59 % of peak for k = 2 on bits 0, 1), its free positions are padded from 11 upwards (the fastest tile
  // shapes measured), and the arithmetic is the unfused register fold of k_gate_kq — IEEE-equal to the dedicated VALU kernel
  // and to the oracle, where the matrix-core form of k = 3 is an fma chain.  Measured at n = 30 (profiles/r03_ops_table.md):
  // k = 2 73 -> 77 % (bits 0, 1: 59 -> 77), k = 3 70 -> 79 % (bits 0-2: 58 -> 78), Swap(1) n-1 <-> 0 75 -> 82 %.
  // Uncontrolled dense single-qubit gates on a position above the rows take the same route (H / X over all targets at n = 30:
  // median 6.53 TB/s against 6.30 for k_gate1q_pair; positions 0..5 keep the cross-lane kernel, 6.5 TB/s).
  // Global option "single_via_tile": 0 = dedicated kernels only, 1 = only when a target lies inside a row, 2 = every dense
  // k = 2, 3 and low-bit swap, 3 (default) = single-qubit gates as well.
  if (g_single_via_tile && !s->force_generic && !g_force_generic && s->mfma != 0 && s->unroll == 0 && !s->swap_single &&
      (p.cls == KC_GATE_KQ || p.cls == KC_GATE_KQ_MFMA || p.cls == KC_SWAP_BITS || (p.cls == KC_GATE1Q_PAIR && g_single_via_tile >= 3)) &&
      // r4: a CONTROLLED dense k = 2, 3 gate takes the sweep as well (controls above the rows come off the grid, controls inside
      // them are lane predicates): k_gate_kq on low targets ran at 42 - 59 %.  Controlled single-qubit gates and swaps keep
      // their dedicated half / quarter sweeps.
      (p.cpos.empty() || ((p.cls == KC_GATE_KQ || p.cls == KC_GATE_KQ_MFMA) && p.opos.size() <= 3)) && s->n >= 17 + (uint32_t)p.cpos.size()) {
    bool low = false;
    for (uint32_t t : p.opos) low = low || t < 6;
    // Complex<f32> (a tile row is 512 B there) has its own switch, "single_via_tile_f32"; measured at n = 30 the sweep is level
    // with or ahead of the packed dedicated kernels as well (H on the top bit 75 -> 81 %, dense k = 2 77 -> 80 %): same default
    const int64_t mode = std::is_same<T, double>::value ? g_single_via_tile : std::min<int64_t>(g_single_via_tile, g_single_via_tile_f32);
    const bool dense23 = p.cls != KC_SWAP_BITS && p.cls != KC_GATE1Q_PAIR && (p.opos.size() == 2 || p.opos.size() == 3);
    const bool dense1 = p.cls == KC_GATE1Q_PAIR && !low && mode >= 3;
    if ((dense23 && (low || mode >= 2)) || (p.cls == KC_SWAP_BITS && low) || dense1) {
      bool done = false;
      QCHK(tile_apply_single<T>(s, op, &done, p.alg_bytes));
      if (done) return QIP_OK;
    }
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
    case KC_SPARSE_KQ: {
      // k = 4, 5 with few entries per row and >= 3 positions outside the wave row: the LDS-staged tile form (whole wave rows, the
      // table through scalar loads) runs ahead of one-group-per-lane; the very same fold, bit-equal
      bool done = false;
      rc = launch_sparse_tile<T>(s, p, f, &done);
      if (rc == QIP_OK && done) rec.cls = KC_SPARSE_TILE;
      else if (rc == QIP_OK) rc = launch_sparse_kq<T>(s, p, f, st);
      break;
    }
    default: {
      if (f.inner->kind == QIP_OP_SPARSE && f.distinct && f.n_op >= 6 && !s->force_generic && !g_force_generic) {
        bool done = false;
        rc = launch_sparse_tile<T>(s, p, f, &done);  // in place: also inside a recorded program (nothing swaps buffers)
        if (rc != QIP_OK || done) {
          rec.cls = KC_SPARSE_TILE;
          break;
        }
        rc = launch_sparse_ell<T>(s, p, f, &done);  // out of place (a recorded program follows the buffers: one graph per parity)
        if (rc != QIP_OK || done) {
          rec.cls = KC_SPARSE_ELL;
          break;
        }
      }
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


// the other translation units call these (qip_internal.h)
template int apply_op_t<double>(qip_hip_state*, const qip_op*);
template int apply_op_t<float>(qip_hip_state*, const qip_op*);
template int launch_gather<double>(qip_hip_state*, const FlatOp&, const amp_t<double>*, uint64_t, amp_t<double>*, uint64_t, uint64_t, uint64_t, int);
template int launch_gather<float>(qip_hip_state*, const FlatOp&, const amp_t<float>*, uint64_t, amp_t<float>*, uint64_t, uint64_t, uint64_t, int);
