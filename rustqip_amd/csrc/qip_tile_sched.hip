// qip_tile_sched.hip — the host-only half of the LDS-resident tile sweeps: which gates form a segment, the passes of a
// segment, the qubit relabelling, and the plan export the CPU tests replay.  No kernel is launched from this file.
#include "qip_tile.h"
#include <array>

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

int classify_tile_item(int dtype, uint32_t n, const qip_op* op, TileItem* it) {
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
uint64_t tile_lane_assignment(const uint32_t pb[3], uint32_t S) {
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


Ins tile_ins(const std::vector<uint32_t>& high, uint32_t p5) {
  std::vector<uint32_t> v = high;
  for (uint32_t& h : v)
    if (h == 5u) h = p5;  // (p5 itself is a low position and never in `high`)
  return make_ins(v, 0);
}

template <typename T>
int build_tile_segment(uint32_t n, bool passes, const std::vector<const TileItem*>& seg_in,
                              std::vector<uint32_t> high, TileSegmentPlan<T>* out, int order_rule, uint32_t p5_override) {
  // The order of the gates INSIDE the segment (order_rule 1 / 2 = the "tile" option; 0 keeps the schedule's order).  A pass is
  // one round trip of the tile through LDS (~0.5 ms per sweep at n = 30: 64 KiB per tile at 128 B per clock), and a pass holds
  // the gates whose exchange bits fit its three register bits: taken in circuit order, a random circuit opens a new pass every
  // 3 - 4 gates.  List scheduling over the commutation relation the scheduler itself uses (two gates commute when on every
  // shared bit both only TEST it; tile = 1 additionally needs one of the two rounding-free, which keeps every amplitude's
  // sequence of rounded operations — the result stays IEEE-equal to circuit order) fills a pass with every ready gate that
  // fits before it opens the next one.
  std::vector<const TileItem*> seg = seg_in;
  const uint32_t p5 = p5_override ? p5_override : tile_p5_of<T>(n);
  out->p5 = p5;
  // pad the free bits with unused positions outside the low set so the tile always has kTileHigh of them
  const uint32_t pad_from = std::min<uint32_t>((uint32_t)g_tile_pad_from, n > (uint32_t)kTileBits ? n - 5 : (uint32_t)kTileLow);
  // (contiguous rows only) position 6 in the tile without position 7 makes the waves' rows alternate 1-KiB pieces (a one-op
  // sweep on position 6: 5.7 TB/s against 6.4 with 7 beside it): 7 is the first pad then
  if (p5 == 5u && g_tile_pad_from > kTileLow && high.size() < (size_t)kTileHigh && n > 7 && std::find(high.begin(), high.end(), 6u) != high.end() &&
      std::find(high.begin(), high.end(), 7u) == high.end())
    high.push_back(7u);
  for (uint32_t p = std::max<uint32_t>(pad_from, kTileLow); high.size() < (size_t)kTileHigh && p < n; ++p)
    if (!tile_is_low(p, p5) && std::find(high.begin(), high.end(), p) == high.end()) high.push_back(p);
  for (uint32_t p = 5; high.size() < (size_t)kTileHigh && p < n; ++p)
    if (!tile_is_low(p, p5) && std::find(high.begin(), high.end(), p) == high.end()) high.push_back(p);
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
  if (g_tile_wave_rule == 1) std::sort(high.begin(), high.end());                                  // lowest positions = wave bits
  else if (g_tile_wave_rule == 2) std::sort(high.begin(), high.end(), std::greater<uint32_t>());  // highest positions = wave bits
  else std::stable_sort(high.begin(), high.end(), [&](uint32_t a, uint32_t b) { return uses[a] < uses[b]; });
  auto tile_bit = [&](uint32_t pos) -> uint32_t {  // kTileOutside when the position is not part of the tile
    if (tile_is_low(pos, p5)) return tile_low_bit(pos);
    const auto f = std::find(high.begin(), high.end(), pos);
    return f == high.end() ? kTileOutside : kTileLow + (uint32_t)(f - high.begin());
  };
  if (passes && order_rule >= 1 && g_tile_sched != 0 && seg.size() >= 3 && seg.size() <= 256) {
    const size_t N = seg.size();
    typedef std::array<uint64_t, 4> Set;  // a set of the segment's gates
    auto has = [](const Set& m, size_t j) { return (m[j >> 6] >> (j & 63)) & 1ull; };
    auto put = [](Set& m, size_t j) { m[j >> 6] |= 1ull << (j & 63); };
    std::vector<uint32_t> ex(N, 0), full(N, 0);  // masks over the tile's eleven bits: exchange bits; those plus the in-tile controls when three slots allow
    for (size_t i = 0; i < N; ++i) {
      const TileItem& it = *seg[i];
      uint32_t e = 0, c = 0;
      if (it.kind == 0) e = 1u << tile_bit(it.t0);
      if (it.kind == 2 || it.kind == 3) e = (1u << tile_bit(it.t0)) | (1u << tile_bit(it.t1));
      if (it.kind == 4) e = (1u << tile_bit(it.t0)) | (1u << tile_bit(it.t1)) | (1u << tile_bit(it.t2));
      for (uint32_t cp : it.cpos)
        if (tile_bit(cp) != kTileOutside) c |= 1u << tile_bit(cp);
      ex[i] = e;
      full[i] = e && __builtin_popcount(e | c) <= 3 ? (e | c) : e;
    }
    std::vector<Set> preds(N, Set{0, 0, 0, 0});
    for (size_t j = 0; j < N; ++j)
      for (size_t i = 0; i < j; ++i) {
        const TileItem &a = *seg[i], &b = *seg[j];
        const bool commute = !(b.nd_mask & (a.nd_mask | a.d_mask)) && !(b.d_mask & a.nd_mask);
        if (!commute || (order_rule < 2 && !a.exact && !b.exact)) put(preds[j], i);
      }
    auto ready = [&](const Set& placed, size_t j) {
      if (has(placed, j)) return false;
      for (int w = 0; w < 4; ++w)
        if (preds[j][w] & ~placed[w]) return false;
      return true;
    };
    // the grouping rule of the pass table below, as a count
    auto count_passes = [&](const std::vector<size_t>& sq) {
      uint32_t open = 0;
      size_t np = 1;
      for (size_t j : sq) {
        if (__builtin_popcount(open | full[j]) <= 3) open |= full[j];
        else if (__builtin_popcount(open | ex[j]) <= 3) open |= ex[j];
        else {
          ++np;
          open = full[j];
        }
      }
      return np;
    };
    std::vector<size_t> original(N);
    for (size_t j = 0; j < N; ++j) original[j] = j;
    // A: grow the open pass — every ready gate that fits it as it stands, then the ready gate that adds the fewest bits; when
    // nothing fits the earliest ready gate opens the next pass
    std::vector<size_t> order_a;
    {
      Set placed{0, 0, 0, 0};
      uint32_t bits = 0;
      while (order_a.size() < N) {
        bool progressed = false;
        for (size_t j = 0; j < N; ++j)
          if (ready(placed, j) && (full[j] & ~bits) == 0) {
            put(placed, j);
            order_a.push_back(j);
            progressed = true;
          }
        if (progressed) continue;
        size_t pick = N;
        int pick_sz = 4;
        for (size_t j = 0; j < N; ++j)
          if (ready(placed, j)) {
            const int sz = __builtin_popcount(bits | full[j]);
            if (sz <= 3 && sz < pick_sz) {
              pick = j;
              pick_sz = sz;
            }
          }
        if (pick == N) {
          bits = 0;
          for (size_t j = 0; j < N; ++j)
            if (ready(placed, j)) {
              pick = j;
              break;
            }
        }
        bits |= full[pick];
        put(placed, pick);
        order_a.push_back(pick);
      }
    }
    // B: pass by pass, the three tile bits whose closure — every gate that becomes ready and exchanges only across them —
    // is largest (gates that compute count 1, diagonal ones, which fit any pass, a little).  165 triples x a closure scan per
    // pass: ~5 us of host time per gate (r6: measured 320 us per 55-gate segment), which a saved pass — one LDS round trip of
    // the tile, 0.5 ms per sweep at n = 30, 8 us at n = 24, 0.5 us at n = 20 — repays only on large states: below 2^24
    // amplitudes the run loop was host-bound by this search (2000 gates at n = 12: 12.4 ms, of which 11.6 here), so it is skipped.
    std::vector<size_t> order_b;
    if (n >= 24u) {
      Set placed{0, 0, 0, 0};
      auto closure = [&](uint32_t tri, Set pl, std::vector<size_t>* emit) {
        double w = 0;
        for (bool again = true; again;) {
          again = false;
          for (size_t j = 0; j < N; ++j)
            if ((ex[j] & ~tri) == 0 && ready(pl, j)) {
              put(pl, j);
              w += ex[j] ? 1.0 : 0.01;
              if (emit) emit->push_back(j);
              again = true;
            }
        }
        return w;
      };
      while (order_b.size() < N) {
        uint32_t best_T = 0;
        double best_w = -1;
        for (uint32_t a = 0; a < (uint32_t)kTileBits; ++a)
          for (uint32_t b = a + 1; b < (uint32_t)kTileBits; ++b)
            for (uint32_t c = b + 1; c < (uint32_t)kTileBits; ++c) {
              const uint32_t tri = (1u << a) | (1u << b) | (1u << c);
              const double w = closure(tri, placed, nullptr);
              if (w > best_w) {
                best_w = w;
                best_T = tri;
              }
            }
        std::vector<size_t> emit;
        closure(best_T, placed, &emit);
        if (emit.empty()) break;  // (cannot happen: some ready gate exchanges across at most three bits)
        for (size_t j : emit) {
          put(placed, j);
          order_b.push_back(j);
        }
      }
    }
    // adopt a new order only when it really saves passes: where the circuit's own order is already as good (QFT: two passes
    // per segment either way) it is kept — its runs of diagonal gates merge better
    const std::vector<size_t>* best = &original;
    size_t best_np = count_passes(original);
    for (const std::vector<size_t>* cand : {&order_a, &order_b})
      if (cand->size() == N && count_passes(*cand) < best_np) {
        best_np = count_passes(*cand);
        best = cand;
      }
    if (best != &original) {
      std::vector<const TileItem*> sq(N);
      for (size_t k = 0; k < N; ++k) sq[k] = seg[(*best)[k]];
      seg = sq;
    }
  }
  out->order.resize(seg.size());
  for (size_t i = 0; i < seg.size(); ++i) out->order[i] = (uint32_t)(std::find(seg_in.begin(), seg_in.end(), seg[i]) - seg_in.begin());
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
    pd.p5 = p5;
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
      if (merged.size() > 3 && !add.empty()) {  // the controls were optional: the exchange bits alone may still fit the open pass
        std::vector<uint32_t> bare;
        if (gates[i].kind == 0) bare = {gates[i].b0};
        if (gates[i].kind == 2 || gates[i].kind == 3) bare = {gates[i].b0, gates[i].b1};
        if (gates[i].kind == 4) bare = {gates[i].b0, gates[i].b1, gates[i].tpos_out};
        if (merge(bare).size() <= 3) merged = merge(bare);
      }
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
// r5: runs of diagonal gates for the interpreter kernel (qip_tile.h TileInterpPlan, qip_kernels.h TileDiagItem).  Each gate of a
// run becomes the one or two steps that perform EXACTLY the products its own code path (TOP_DIAG_UNIFORM / _LANE / _LANE_CTL /
// _REGJ in k_tile_passes) performs, in the same order:
//   UNIFORM  (target outside, no lane-bit control): f = m[target bit of the base]; a unit f is skipped; the elements whose pass-bit
//            controls are 1 are multiplied by f                       -> per non-unit half h: {f1 = m[h], outside: omask + target = h}
//   LANE     (target on a lane bit):  F = lane's target bit ? m[1] : m[0], every element whose pass-bit controls are 1 is multiplied
//            (also where F is the unit)                                -> {f0 = m[0], f1 = m[1], sel = target bit}
//   LANE_CTL (lane-bit controls; target on a lane bit or outside): the same with F = (1, 0) on lanes whose controls are 0; with the
//            target outside both halves are kept, units too (the op multiplies by whatever m[bit] is)
//   REG J    (target = pass bit J): per non-unit half h, F = m[h] (or (1, 0) where a lane-bit control is 0) on the elements with
//            bit J = h and the pass-bit controls 1                     -> {f1 = m[h], reg_mask += bit J, reg_val += h << J}
// ---------------------------------------------------------------------------------------
int64_t g_tile_diag_runs = 1;
template <typename T>
void tile_merge_diag_runs(const TileSegmentPlan<T>& plan, TileInterpPlan<T>* out, uint32_t min_run) {
  out->gates.clear();
  out->items.clear();
  out->pd = plan.pd;
  out->runs = out->gates_in_runs = 0;
  auto unit = [](const amp_t<T>& a) { return a.x == (T)1 && a.y == (T)0; };
  auto is_diag = [](const TileGate<T>& g) { return g.kind == 1 && g.op <= (uint32_t)TOP_DIAG_REG2; };
  for (uint32_t pi = 0; pi < plan.pd.npasses; ++pi) {
    const TilePass& ps = plan.pd.pass[pi];
    TilePass& po = out->pd.pass[pi];
    po.first = (uint32_t)out->gates.size();
    uint32_t gi = ps.first;
    const uint32_t gend = ps.first + ps.count;
    while (gi < gend) {
      uint32_t ge = gi;
      while (ge < gend && is_diag(plan.gates[ge])) ++ge;
      if (ge - gi < min_run) {  // not a run: the gate as it is (and the non-diagonal gate that ended the scan, if any)
        const uint32_t upto = std::max(ge, gi + 1);
        for (; gi < upto && gi < gend; ++gi) out->gates.push_back(plan.gates[gi]);
        continue;
      }
      TileGate<T> run;
      memset(&run, 0, sizeof run);
      run.kind = 1;
      run.op = TOP_DIAG_RUN;
      run.b0 = kTileOutside;
      run.nz = (uint32_t)out->items.size();
      // element i of a lane's eight holds the pass-bit combination c[i] (bit j of i on pass bit pb[j]): the pass-bit condition of a
      // step is resolved here into one bit per element
      uint32_t cbits[8];
      for (int i = 0; i < 8; ++i) cbits[i] = ((uint32_t)(i & 1) << ps.pb[0]) | ((uint32_t)((i >> 1) & 1) << ps.pb[1]) | ((uint32_t)((i >> 2) & 1) << ps.pb[2]);
      auto push = [&](amp_t<T> f0, amp_t<T> f1, uint32_t lane_mask, uint32_t lane_val, uint32_t reg_mask, uint32_t reg_val, uint64_t omask, uint64_t oval,
                      uint32_t sel_bit_plus_1) {
        TileDiagItem<T> t;
        memset(&t, 0, sizeof t);
        t.f0 = f0;
        t.f1 = f1;
        t.omask = omask;
        t.oval = oval;
        t.lane_mask = lane_mask;
        t.lane_val = lane_val;
        uint32_t emask = 0;
        for (int i = 0; i < 8; ++i)
          if ((cbits[i] & reg_mask) == reg_val) emask |= 1u << i;
        t.emask_sel = emask | (sel_bit_plus_1 << 24);
        t.reg_pack = reg_mask | (reg_val << 16);  // (tile-index bits: < 2^11 each)
        out->items.push_back(t);
      };
      for (; gi < ge; ++gi) {
        const TileGate<T>& g = plan.gates[gi];
        const bool outside = g.b0 == kTileOutside;
        if (g.op == TOP_DIAG_UNIFORM || (outside && (g.op == TOP_DIAG_LANE || g.op == TOP_DIAG_LANE_CTL))) {
          for (int h = 0; h < 2; ++h) {
            if (g.op == TOP_DIAG_UNIFORM && unit(g.m[h])) continue;
            push(g.m[h], g.m[h], g.cm_lane, g.cm_lane, g.cm_reg, g.cm_reg, g.omask | (1ull << g.tpos_out), g.omask | ((uint64_t)h << g.tpos_out), 0u);
          }
        } else if (g.op == TOP_DIAG_LANE || g.op == TOP_DIAG_LANE_CTL) {
          // F = (target bit ? m[1] : m[0]), then (1, 0) where a lane-bit control is 0.  When one entry IS the unit (phase gates) the
          // target bit is just one more lane condition with the SAME F on every lane: (bit = h and controls) ? m[h] : (1, 0)
          if (unit(g.m[0]) || unit(g.m[1])) {
            const int h = unit(g.m[0]) ? 1 : 0;
            push(g.m[h], g.m[h], g.cm_lane | (1u << g.b0), g.cm_lane | ((uint32_t)h << g.b0), g.cm_reg, g.cm_reg, g.omask, g.omask, 0u);
          } else {
            push(g.m[0], g.m[1], g.cm_lane, g.cm_lane, g.cm_reg, g.cm_reg, g.omask, g.omask, g.b0 + 1u);
          }
        } else {  // TOP_DIAG_REG0..2: the target is a pass bit
          for (int h = 0; h < 2; ++h) {
            if (unit(g.m[h])) continue;
            push(g.m[h], g.m[h], g.cm_lane, g.cm_lane, g.cm_reg | (1u << g.b0), g.cm_reg | ((uint32_t)h << g.b0), g.omask, g.omask, 0u);
          }
        }
        out->gates_in_runs += 1;
      }
      run.b1 = (uint32_t)out->items.size() - run.nz;
      out->runs += 1;
      if (run.b1) out->gates.push_back(run);  // (a run of identities has no step)
    }
    po.count = (uint32_t)out->gates.size() - po.first;
  }
}
template void tile_merge_diag_runs<double>(const TileSegmentPlan<double>&, TileInterpPlan<double>*, uint32_t);
template void tile_merge_diag_runs<float>(const TileSegmentPlan<float>&, TileInterpPlan<float>*, uint32_t);

// ---------------------------------------------------------------------------------------
// Wide tiles (r4): the plan of one segment for the register-resident 13-bit tile (qip_tile.h WidePlan).  Tile bits 0..5 = the
// rows (positions 0..4 and p5), 6..12 = the seven high positions.  At load / store time the thread id fills tile bits 0..7
// (rows, then the two wave positions) and a lane's 32 accesses walk tile bits 8..12: that is arrangement 0, and gates on
// those five bits need no LDS traffic at all.  Any other gate needs its exchange bits among the five register bits: the tile
// is TRANSPOSED through the 32-KiB LDS buffer in four quarters — the quarter is selected by two tile bits that are register
// bits before and after, so a transposition brings in up to three new register bits and keeps at least two.  The last
// transposition(s) lead back to arrangement 0.  Gates keep the order they are given in.
// ---------------------------------------------------------------------------------------
template <typename T>
int build_wide_segment(uint32_t n, const std::vector<const TileItem*>& seg_in, std::vector<uint32_t> high, WidePlan<T>* out, int order_rule) {
  std::vector<const TileItem*> seg = seg_in;
  const uint32_t p5 = tile_p5_of<T>(n);
  out->p5 = p5;
  if (n < (uint32_t)kWideBits) return fail(QIP_ERR_UNSUPPORTED, "wide tiles need n >= %d", kWideBits);
  const uint32_t pad_from = std::min<uint32_t>((uint32_t)g_tile_pad_from, n > (uint32_t)kWideBits ? n - 7 : 5u);
  for (uint32_t p = std::max<uint32_t>(pad_from, 5u); high.size() < (size_t)kWideHigh && p < n; ++p)
    if (!tile_is_low(p, p5) && std::find(high.begin(), high.end(), p) == high.end()) high.push_back(p);
  for (uint32_t p = 5; high.size() < (size_t)kWideHigh && p < n; ++p)
    if (!tile_is_low(p, p5) && std::find(high.begin(), high.end(), p) == high.end()) high.push_back(p);
  if (high.size() != (size_t)kWideHigh) return fail(QIP_ERR_UNSUPPORTED, "a wide segment with %zu high positions", high.size());
  // the two positions that are exchange targets least often become the wave bits (tile bits 6, 7): the five others are
  // register bits from the start
  std::vector<uint32_t> uses(64, 0);
  for (const TileItem* it : seg) {
    if (it->kind == 0) uses[it->t0] += 1;
    if (it->kind >= 2) {
      uses[it->t0] += 1;
      uses[it->t1] += 1;
    }
    if (it->kind == 4) uses[it->t2] += 1;
  }
  std::sort(high.begin(), high.end());
  std::stable_sort(high.begin(), high.end(), [&](uint32_t a, uint32_t b) { return uses[a] < uses[b]; });
  std::sort(high.begin(), high.begin() + 2);
  std::sort(high.begin() + 2, high.end());
  out->high = high;
  auto tile_bit = [&](uint32_t pos) -> uint32_t {
    if (tile_is_low(pos, p5)) return tile_low_bit(pos);
    const auto f = std::find(high.begin(), high.end(), pos);
    return f == high.end() ? kTileOutside : kTileLow + (uint32_t)(f - high.begin());
  };
  typedef std::vector<uint32_t> Bits;
  auto has = [](const Bits& v, uint32_t b) { return std::find(v.begin(), v.end(), b) != v.end(); };
  const Bits load_bits = {8, 9, 10, 11, 12};
  auto item_exch = [&](const TileItem& it) {
    Bits e;
    if (it.kind == 0) e = {tile_bit(it.t0)};
    if (it.kind == 2 || it.kind == 3) e = {tile_bit(it.t0), tile_bit(it.t1)};
    if (it.kind == 4) e = {tile_bit(it.t0), tile_bit(it.t1), tile_bit(it.t2)};
    return e;
  };
  // The register sets a given gate order leads to.  At a gate whose exchange bits are not all register bits the tile is
  // transposed: the new bits of that gate and of the gates behind it join while three new bits and five in all suffice; the
  // slots left keep the old register bits whose next use (in this order) comes soonest (Belady).  Returns the sets, one per
  // pass, the first gate of each, and (by size) the number of transpositions including the way back to the load set.
  struct Arr {
    Bits R;
    size_t first;
  };
  auto arrangements = [&](const std::vector<const TileItem*>& sq) {
    std::vector<Bits> ex(sq.size());
    for (size_t i = 0; i < sq.size(); ++i) ex[i] = item_exch(*sq[i]);
    std::vector<Arr> arr;
    arr.push_back({load_bits, 0});
    for (size_t gi = 0; gi < sq.size(); ++gi) {
      const Bits& R = arr.back().R;
      bool fits = true;
      for (uint32_t t : ex[gi]) fits = fits && has(R, t);
      if (fits) continue;
      Bits N, K;
      for (uint32_t t : ex[gi]) (has(R, t) ? K : N).push_back(t);
      for (size_t gj = gi + 1; gj < sq.size(); ++gj) {
        Bits n2 = N, k2 = K;
        for (uint32_t t : ex[gj]) {
          if (has(n2, t) || has(k2, t)) continue;
          (has(R, t) ? k2 : n2).push_back(t);
        }
        if (n2.size() > 3 || n2.size() + k2.size() > (size_t)kWideRegBits) break;
        N = n2;
        K = k2;
      }
      Bits newR = N;
      for (uint32_t t : K) newR.push_back(t);
      std::vector<std::pair<size_t, uint32_t>> rest;  // (next use, bit) of the old register bits not yet kept
      for (uint32_t t : R) {
        if (has(newR, t)) continue;
        size_t nu = sq.size() + (has(load_bits, t) ? 0 : 1);  // (never used again: prefer the load set's bits, the way back is shorter)
        for (size_t gj = gi + 1; gj < sq.size(); ++gj)
          if (has(ex[gj], t)) {
            nu = gj;
            break;
          }
        rest.push_back({nu, t});
      }
      std::sort(rest.begin(), rest.end());
      for (const auto& r : rest)
        if (newR.size() < (size_t)kWideRegBits) newR.push_back(r.second);
      arr.push_back({newR, gi});
    }
    if (arr.size() > 1) {
      for (int hop = 0; hop < 3; ++hop) {
        const Bits& R = arr.back().R;
        size_t common = 0;
        for (uint32_t t : load_bits) common += has(R, t);
        if (common >= 2) {
          arr.push_back({load_bits, sq.size()});
          break;
        }
        Bits mid;
        for (uint32_t t : R)
          if (mid.size() < 2) mid.push_back(t);
        for (uint32_t t : load_bits)
          if (mid.size() < (size_t)kWideRegBits && !has(mid, t)) mid.push_back(t);
        arr.push_back({mid, sq.size()});
      }
    }
    return arr;
  };
  // The order of the gates inside the segment (order_rule 1 / 2 = the "tile" option, as in build_tile_segment): list scheduling
  // over the scheduler's own commutation relation — two gates commute when on every shared bit both only test it; tile = 1
  // additionally needs one of the two rounding-free, which keeps every amplitude's sequence of rounded operations — emits
  // every ready gate that fits the register set before it transposes, and transposes for the ready gate that needs the
  // fewest new bits.  Adopted only where it saves transpositions.
  if (order_rule >= 1 && g_tile_sched != 0 && seg.size() >= 3 && seg.size() <= 256) {
    const size_t NG = seg.size();
    typedef std::array<uint64_t, 4> Set;
    auto sget = [](const Set& m, size_t j) { return (m[j >> 6] >> (j & 63)) & 1ull; };
    auto sput = [](Set& m, size_t j) { m[j >> 6] |= 1ull << (j & 63); };
    std::vector<Set> preds(NG, Set{0, 0, 0, 0});
    for (size_t j = 0; j < NG; ++j)
      for (size_t i = 0; i < j; ++i) {
        const TileItem &a = *seg[i], &b = *seg[j];
        const bool commute = !(b.nd_mask & (a.nd_mask | a.d_mask)) && !(b.d_mask & a.nd_mask);
        if (!commute || (order_rule < 2 && !a.exact && !b.exact)) sput(preds[j], i);
      }
    std::vector<Bits> ex(NG);
    for (size_t i = 0; i < NG; ++i) ex[i] = item_exch(*seg[i]);
    Set placed{0, 0, 0, 0};
    auto ready = [&](size_t j) {
      if (sget(placed, j)) return false;
      for (int w = 0; w < 4; ++w)
        if (preds[j][w] & ~placed[w]) return false;
      return true;
    };
    std::vector<size_t> ord;
    Bits R = load_bits;
    while (ord.size() < NG) {
      bool progressed = false;
      for (size_t j = 0; j < NG; ++j) {
        if (!ready(j)) continue;
        bool fits = true;
        for (uint32_t t : ex[j]) fits = fits && has(R, t);
        if (fits) {
          sput(placed, j);
          ord.push_back(j);
          progressed = true;
        }
      }
      if (progressed) continue;
      size_t pick = NG, pick_new = 9;
      for (size_t j = 0; j < NG; ++j) {
        if (!ready(j)) continue;
        size_t nb = 0;
        for (uint32_t t : ex[j]) nb += !has(R, t);
        if (nb < pick_new) {
          pick = j;
          pick_new = nb;
        }
      }
      if (pick == NG) break;  // (cannot happen: the earliest unplaced gate is always ready)
      Bits N, K;
      for (uint32_t t : ex[pick]) (has(R, t) ? K : N).push_back(t);
      for (size_t j = 0; j < NG; ++j) {  // other ready gates whose bits still fit join the new set
        if (!ready(j) || j == pick) continue;
        Bits n2 = N, k2 = K;
        for (uint32_t t : ex[j]) {
          if (has(n2, t) || has(k2, t)) continue;
          (has(R, t) ? k2 : n2).push_back(t);
        }
        if (n2.size() > 3 || n2.size() + k2.size() > (size_t)kWideRegBits) continue;
        N = n2;
        K = k2;
      }
      Bits newR = N;
      for (uint32_t t : K) newR.push_back(t);
      std::vector<std::pair<size_t, uint32_t>> rest;
      for (uint32_t t : R) {
        if (has(newR, t)) continue;
        size_t nu = NG + (has(load_bits, t) ? 0 : 1);
        for (size_t j = 0; j < NG; ++j)
          if (!sget(placed, j) && has(ex[j], t)) {
            nu = j;
            break;
          }
        rest.push_back({nu, t});
      }
      std::sort(rest.begin(), rest.end());
      for (const auto& r : rest)
        if (newR.size() < (size_t)kWideRegBits) newR.push_back(r.second);
      R = newR;
    }
    if (ord.size() == NG) {
      std::vector<const TileItem*> sq(NG);
      for (size_t k = 0; k < NG; ++k) sq[k] = seg[ord[k]];
      if (arrangements(sq).size() < arrangements(seg).size()) seg = sq;
    }
  }
  const std::vector<Arr> arr = arrangements(seg);
  std::vector<TileGate<T>>& gates = out->gates;
  std::vector<amp_t<T>>& mats = out->mats;
  gates.assign(seg.size(), TileGate<T>());
  mats.clear();
  out->order.resize(seg.size());
  for (size_t i = 0; i < seg.size(); ++i) {
    out->order[i] = (uint32_t)(std::find(seg_in.begin(), seg_in.end(), seg[i]) - seg_in.begin());
    const TileItem& it = *seg[i];
    TileGate<T>& g = gates[i];
    memset(&g, 0, sizeof g);
    g.kind = (uint32_t)it.kind;
    g.b0 = tile_bit(it.t0);
    g.b1 = it.kind >= 2 ? tile_bit(it.t1) : 0;
    if (it.kind == 3 || it.kind == 4) {
      g.nz = (uint32_t)(mats.size() / 16);
      const int cnt = it.kind == 3 ? 16 : 64;
      for (int e = 0; e < cnt; ++e) mats.push_back(mk<T>(it.mat[2 * e], it.mat[2 * e + 1]));
    }
    if (it.kind == 4) g.tpos_out = tile_bit(it.t2);
    if (it.kind == 1 && g.b0 == kTileOutside) g.tpos_out = it.t0;
    for (uint32_t c : it.cpos) {
      const uint32_t tb = tile_bit(c);
      if (tb == kTileOutside) g.omask |= 1ull << c;
      else g.cmask |= 1u << tb;
    }
    if (it.kind != 3 && it.kind != 4) g.nz = it.nz;
    if (it.kind == 0) {
      for (int e = 0; e < 4; ++e) g.m[e] = mk<T>(it.m[2 * e], it.m[2 * e + 1]);
      const bool real = it.m[1] == 0 && it.m[3] == 0 && it.m[5] == 0 && it.m[7] == 0;
      const bool is_x = it.nz == 6u && it.m[2] == 1 && it.m[3] == 0 && it.m[4] == 1 && it.m[5] == 0;
      g.b1 = (real ? 1u : 0u) | (is_x ? 2u : 0u);
    } else if (it.kind == 1) {
      g.m[0] = mk<T>(it.m[0], it.m[1]);
      g.m[1] = mk<T>(it.m[2], it.m[3]);
    }
    if ((it.kind != 1 && g.b0 == kTileOutside) || (it.kind >= 2 && g.b1 == kTileOutside) || (it.kind == 4 && g.tpos_out == kTileOutside))
      return fail(QIP_ERR_INVALID, "internal: an exchange target outside the wide tile");
  }
  // LDS layout of one transposition (see WidePass::bufpos): thread-id bits 0..3 of the writing arrangement land on buffer bits
  // 0..3, those of the reading arrangement on buffer bits of pairwise distinct class (bit mod 4, below 8) — with the slot
  // swizzle (tile_slot: bits 4..7 folded onto 0..3) the 16 lanes of a ds_read_b128 group then hit 16 different 16-byte slots
  auto layout = [&](const WidePass& a, WidePass* b) {
    int bp[kWideBits];
    bool used[11] = {false};
    for (int t = 0; t < kWideBits; ++t) bp[t] = -1;
    for (int k = 0; k < 4; ++k) {
      bp[a.L[k]] = k;
      used[k] = true;
    }
    for (int k = 0; k < 4; ++k) {
      const uint32_t t = b->L[k];
      if (bp[t] >= 0) continue;
      bool cls_taken[4] = {false, false, false, false};
      for (int k2 = 0; k2 < 4; ++k2)
        if (bp[b->L[k2]] >= 0 && bp[b->L[k2]] < 8) cls_taken[bp[b->L[k2]] & 3] = true;
      for (int c = 0; c < 4 && bp[t] < 0; ++c) {
        if (cls_taken[c]) continue;
        for (int cand : {c, c + 4})
          if (!used[cand]) {
            bp[t] = cand;
            used[cand] = true;
            break;
          }
      }
    }
    int next = 0;
    for (int t = 0; t < kWideBits; ++t) {
      if ((uint32_t)t == b->q[0] || (uint32_t)t == b->q[1] || bp[t] >= 0) continue;
      while (used[next]) ++next;
      bp[t] = next;
      used[next] = true;
    }
    for (int t = 0; t < kWideBits; ++t) b->bufpos[t] = bp[t] < 0 ? 0u : (uint32_t)bp[t];
  };
  std::vector<WidePass>& passes = out->passes;
  passes.clear();
  WidePass load;
  for (int j = 0; j < kWideRegBits; ++j) load.R[j] = 8 + (uint32_t)j;
  for (int k = 0; k < 8; ++k) load.L[k] = (uint32_t)k;
  for (size_t ai = 0; ai < arr.size(); ++ai) {
    const size_t next_first = ai + 1 < arr.size() ? arr[ai + 1].first : seg.size();
    WidePass b;
    if (ai == 0 || (ai + 1 == arr.size() && arr[ai].R == load_bits && arr[ai].first == seg.size())) {
      b = load;  // the load arrangement: at the start, and as the way back (same register order, same lane map)
    } else {
      for (int j = 0; j < kWideRegBits; ++j) b.R[j] = arr[ai].R[j];
      int k = 0;
      for (uint32_t t = 0; t < (uint32_t)kWideBits; ++t)
        if (!has(arr[ai].R, t)) b.L[k++] = t;
    }
    b.first = (uint32_t)std::min(arr[ai].first, seg.size());
    b.count = (uint32_t)(next_first - std::min(arr[ai].first, seg.size()));
    if (ai == 0) {
      b.first = 0;
      b.count = (uint32_t)next_first;
    } else {
      const Bits prevR(passes.back().R, passes.back().R + kWideRegBits);
      Bits common;
      for (int j = 0; j < kWideRegBits; ++j)
        if (has(prevR, b.R[j])) common.push_back(b.R[j]);
      if (common.size() < 2) return fail(QIP_ERR_INVALID, "internal: a wide transposition needs two common register bits");
      b.transposed = true;
      b.q[0] = common[0];
      b.q[1] = common[1];
      layout(passes.back(), &b);
    }
    passes.push_back(b);
  }
  if (passes.size() > 1) {
    const WidePass& last = passes.back();
    for (int j = 0; j < kWideRegBits; ++j)
      if (last.R[j] != load.R[j]) return fail(QIP_ERR_INVALID, "internal: the wide plan does not return to the load arrangement");
    for (int k = 0; k < 8; ++k)
      if (last.L[k] != load.L[k]) return fail(QIP_ERR_INVALID, "internal: the wide plan does not return to the load lane map");
  }
  if (getenv("QIP_WIDE_DEBUG")) {
    size_t exg = 0;
    for (const auto& g : gates) exg += g.kind != 1;
    fprintf(stderr, "[wide] %zu gates (%zu exchanging), %zu transpositions\n", gates.size(), exg, passes.size() - 1);
  }
  return QIP_OK;
}
template int build_wide_segment<double>(uint32_t, const std::vector<const TileItem*>&, std::vector<uint32_t>, WidePlan<double>*, int);
template int build_wide_segment<float>(uint32_t, const std::vector<const TileItem*>&, std::vector<uint32_t>, WidePlan<float>*, int);

// ---------------------------------------------------------------------------------------
// Which five positions should a segment claim?  First come, first served (the scan below claims positions in the order the
// circuit asks for them) spends them on whatever the next few gates touch.  The gain rule looks at what each position
// would BUY: starting from the positions the head gate needs, it repeatedly claims the free
// position that lets the most further gates join this segment (a dry run of the same scan with the claimed set frozen;
// gates that exchange amplitudes count 1, diagonal ones — which never need a position — a little), until five are claimed
// or no single position adds anything; what is left is claimed first-come as before.  Host arithmetic only, and the
// schedule's invariants do not depend on it: every gate still joins under the commutation / exactness rules of the scan.
// ---------------------------------------------------------------------------------------
thread_local int t_tile_high = kTileHigh;  // make_tile_schedule sets it from the mode (bit 4: wide tiles, kWideHigh)
static thread_local int t_seg_rule = 0;  // the rule in force for the plan being made (make_tile_schedule tries several)
struct SegScan {
  const std::vector<TileItem>& L;
  const std::vector<char>& done;
  uint64_t count, window;
  bool reorder, relabel;
  size_t max_ops, max_exch;
  uint32_t p5;  // position of tile bit 5: which positions are free of charge (tile_is_low)
};

// weight of the gates that join a segment grown from `head` when exactly the positions `H` may be used above the rows
static double seg_dry_run(const SegScan& c, uint64_t head, std::vector<uint32_t> phys, const std::vector<uint32_t>& H) {
  uint64_t blocked_nd = 0, blocked_d = 0;
  bool skipped_inexact = false, any_skipped = false;
  size_t joined = 0, exch_gates = 0;
  double w = 0;
  for (uint64_t i = head; i < c.count && i <= head + (any_skipped ? c.window : c.count) && joined < c.max_ops; ++i) {
    if (c.done[i]) continue;
    const TileItem& it = c.L[i];
    if (c.relabel && !any_skipped && !it.swap_pairs.empty()) {
      for (const auto& pr : it.swap_pairs) std::swap(phys[pr.first], phys[pr.second]);
      continue;
    }
    const bool commutes = !(it.nd_mask & (blocked_nd | blocked_d)) && !(it.d_mask & blocked_nd);
    bool fits = it.tileable && commutes && (c.reorder || it.exact || !skipped_inexact) && (it.kind == 1 || exch_gates < c.max_exch);
    if (fits) {
      uint32_t exch[3];
      int ne = 0;
      if (it.kind == 0) exch[ne++] = it.t0;
      if (it.kind == 2 || it.kind == 3) { exch[ne++] = it.t0; exch[ne++] = it.t1; }
      if (it.kind == 4) { exch[ne++] = it.t0; exch[ne++] = it.t1; exch[ne++] = it.t2; }
      for (int e = 0; e < ne && fits; ++e) {
        const uint32_t pp = phys[exch[e]];
        fits = tile_is_low(pp, c.p5) || std::find(H.begin(), H.end(), pp) != H.end();
      }
    }
    if (fits) {
      joined += 1;
      exch_gates += it.kind != 1;
      w += it.kind != 1 ? 1.0 : (t_seg_rule == 3 ? 0.5 : 0.125);
    } else {
      blocked_nd |= it.nd_mask;
      blocked_d |= it.d_mask;
      skipped_inexact = skipped_inexact || !it.exact;
      any_skipped = true;
    }
  }
  return w;
}

static std::vector<uint32_t> seg_choose_high(const SegScan& c, uint64_t head, const std::vector<uint32_t>& phys, uint32_t n) {
  std::vector<uint32_t> H;
  if (t_seg_rule == 0) return H;
  {
    const TileItem& it = c.L[head];
    uint32_t exch[3];
    int ne = 0;
    if (it.kind == 0) exch[ne++] = it.t0;
    if (it.kind == 2 || it.kind == 3) { exch[ne++] = it.t0; exch[ne++] = it.t1; }
    if (it.kind == 4) { exch[ne++] = it.t0; exch[ne++] = it.t1; exch[ne++] = it.t2; }
    for (int e = 0; e < ne; ++e) {
      const uint32_t pp = phys[exch[e]];
      if (!tile_is_low(pp, c.p5) && std::find(H.begin(), H.end(), pp) == H.end()) H.push_back(pp);
    }
  }
  double base = seg_dry_run(c, head, phys, H);
  while (H.size() < (size_t)t_tile_high) {
    int best = -1;
    double best_w = base;
    for (uint32_t pp = 5; pp < n; ++pp) {
      if (tile_is_low(pp, c.p5) || std::find(H.begin(), H.end(), pp) != H.end()) continue;
      H.push_back(pp);
      const double w = seg_dry_run(c, head, phys, H);
      H.pop_back();
      if (w > best_w + 1e-9) {
        best_w = w;
        best = (int)pp;
      }
    }
    if (best < 0) break;
    H.push_back((uint32_t)best);
    base = best_w;
  }
  return H;
}


// Pure host scheduling (no device, no launches).  Invariants, checked by tests/test_host_ops.py through
// qip_hip_plan_tiles: every op appears in exactly one step; an op only overtakes ops it commutes with (on every
// shared bit both only test it); without `reorder` only when it, or every op it overtakes, is rounding-free.
int schedule_tiles(int dtype, uint32_t n, const qip_op* ops, uint64_t count, bool reorder,
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
  const uint32_t p5 = tile_p5(dtype, n);
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
      if (run.ops.size() >= 2 && __builtin_popcountll(moved & ~tile_low_mask(p5)) > t_tile_high) {
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
    {
      std::vector<uint32_t> ident(n);
      for (uint32_t b = 0; b < n; ++b) ident[b] = b;
      const SegScan sc{items, done, count, window, reorder, false, (size_t)kTileMaxGates, (size_t)kTileMaxExchGates, p5};
      st.high = seg_choose_high(sc, head, ident, n);
    }
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
          if (!tile_is_low(p, p5) && std::find(st.high.begin(), st.high.end(), p) == st.high.end() &&
              std::find(need.begin(), need.end(), p) == need.end())
            need.push_back(p);
        fits = st.high.size() + need.size() <= (size_t)t_tile_high;
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

int schedule_tiles_relabel(int dtype, uint32_t n, const qip_op* ops, uint64_t count, bool reorder, bool allow_2q,
                                  TileSchedule* out) {
  // `out->init_phys` (optional): the layout the state is in when the schedule starts (a previous call left it relabelled);
  // `out->keep_layout`: do not close with the restoring permutation sweep — the final layout is returned in `final_phys`
  // and stays in force (option tile_relabel = 3: the layout persists across apply_ops calls)
  std::vector<TileItem> L(count);  // the caller's ops, logical bit positions
  for (uint64_t i = 0; i < count; ++i) {
    int rc = classify_tile_item(dtype, n, &ops[i], &L[i]);
    if (rc != QIP_OK) {
      std::string msg = g_last_error;
      return fail(rc, "op %llu: %s", (unsigned long long)i, msg.c_str());
    }
    if (L[i].kind >= 3 && !allow_2q) L[i].tileable = false;
  }
  const uint32_t p5 = tile_p5(dtype, n);
  std::vector<uint32_t> phys(n);  // phys[p] = physical position of logical bit position p
  for (uint32_t p = 0; p < n; ++p) phys[p] = p;
  if (out->init_phys.size() == n) phys = out->init_phys;
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
    {
      const SegScan sc{L, done, count, window, reorder, true, max_circuit_ops, (size_t)kTileMaxExchGates - (size_t)kTileLow, p5};
      st.high = seg_choose_high(sc, head, phys, n);
    }
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
          if (!tile_is_low(pp, p5) && std::find(st.high.begin(), st.high.end(), pp) == st.high.end() &&
              std::find(need.begin(), need.end(), pp) == need.end())
            need.push_back(pp);
        }
        fits = st.high.size() + need.size() <= (size_t)t_tile_high;
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
      auto in_tile = [&](uint32_t pp) { return tile_is_low(pp, p5) || std::find(st.high.begin(), st.high.end(), pp) != st.high.end(); };
      std::vector<uint32_t> by_use;  // logical positions with a future use, soonest first
      for (uint32_t p = 0; p < n; ++p)
        if (nxt[p] != ~0ull) by_use.push_back(p);
      std::stable_sort(by_use.begin(), by_use.end(), [&](uint32_t a, uint32_t b) { return nxt[a] < nxt[b]; });
      for (uint32_t p : by_use) {
        if (st.high.size() >= (size_t)t_tile_high) break;
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
        if (wanted && !tile_is_low(phys[p], p5)) bring.push_back(p);
        if (!wanted && tile_is_low(phys[p], p5)) evict.push_back(p);
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
  out->final_phys = phys;
  if (!identity && !out->keep_layout) {
    TileStep back;
    back.perm = phys;
    out->steps.push_back(back);
    for (uint32_t p = 0; p < n; ++p) out->final_phys[p] = p;
  }
  out->circuit = out->owned.data();
  out->count = out->owned.size();
  return QIP_OK;
}

// mode: bits 0-1 = the "tile" option (1 = circuit order, 2 = commuting reorder), bit 2 = relabel the qubits when that
// shortens the plan, bit 3 = relabel unconditionally
static int make_tile_schedule_rule(int dtype, uint32_t n, const qip_op* ops, uint64_t count, int mode, bool allow_2q, TileSchedule* out,
                                   bool allow_permute) {
  const bool reorder = (mode & 3) >= 2;
  bool start_identity = true;
  for (uint32_t p = 0; p < out->init_phys.size(); ++p) start_identity = start_identity && out->init_phys[p] == p;
  if ((mode & 4) && allow_permute) {
    // relabelling pays for random circuits; layered ones (Grover's X / H walls, QFT) gain nothing and would only pay the
    // closing permutation: schedule both ways (host work, microseconds per gate) and keep the shorter plan.  With a
    // persistent layout (keep_layout) the closing sweep is not part of this call; a plan that starts from a relabelled
    // state has no plain alternative short of restoring the order first (one sweep).
    const std::vector<uint32_t> init = out->init_phys;
    const bool keep = out->keep_layout;
    QCHK(schedule_tiles_relabel(dtype, n, ops, count, reorder, allow_2q, out));
    if (!start_identity) return QIP_OK;
    TileSchedule plain;
    QCHK(schedule_tiles(dtype, n, ops, count, reorder, &plain.items, &plain.steps, allow_2q));
    if (out->steps.size() < plain.steps.size() || (mode & 8)) return QIP_OK;  // bit 3: keep it regardless (tests)
    *out = TileSchedule();
    out->init_phys = init;
    out->keep_layout = keep;
  }
  out->final_phys.resize(n);
  for (uint32_t p = 0; p < n; ++p) out->final_phys[p] = p;
  out->circuit = ops;
  out->count = count;
  return schedule_tiles(dtype, n, ops, count, reorder, &out->items, &out->steps, allow_2q, allow_permute);
}

// The plan under each rule for claiming a segment's positions (first come / what a position buys, two weights for the
// diagonal gates), shortest kept: host arithmetic, microseconds per gate, against ~6 ms per sweep saved at n = 30.  Below
// n = 24 a sweep costs less than the search: first come only.  Global option "tile_sched": 0 = first come only (and the
// circuit's own gate order inside every segment), 1 = default, 2 = search at every size and in every mode (tests).
static int make_tile_schedule_inner(int dtype, uint32_t n, const qip_op* ops, uint64_t count, int mode, bool allow_2q, TileSchedule* out,
                                    bool allow_permute);
int make_tile_schedule(int dtype, uint32_t n, const qip_op* ops, uint64_t count, int mode, bool allow_2q, TileSchedule* out,
                       bool allow_permute) {
  // mode bit 4: wide tiles (seven free positions per segment; a state of at least kWideBits + 1 qubits)
  t_tile_high = ((mode & 16) && n > (uint32_t)kWideBits) ? kWideHigh : kTileHigh;
  const int rc = make_tile_schedule_inner(dtype, n, ops, count, mode, allow_2q, out, allow_permute);
  t_tile_high = kTileHigh;
  return rc;
}
static int make_tile_schedule_inner(int dtype, uint32_t n, const qip_op* ops, uint64_t count, int mode, bool allow_2q, TileSchedule* out,
                                    bool allow_permute) {
  // (only where gates may be reordered freely, tile = 2: in the IEEE-equal mode the sweeps are bound by f64 issue, a plan with
  // fewer, heavier sweeps is not faster there — Grover 15 -> 14 sweeps measured 2 % slower — and first come stays)
  const bool search = g_tile_sched != 0 && (n >= 24 || g_tile_sched == 2) && count >= 8 && ((mode & 3) >= 2 || g_tile_sched == 2);  // (2 = always: tests)
  if (!search) {
    t_seg_rule = 0;
    return make_tile_schedule_rule(dtype, n, ops, count, mode, allow_2q, out, allow_permute);
  }
  const std::vector<uint32_t> init = out->init_phys;
  const bool keep = out->keep_layout;
  TileSchedule best;
  bool have = false;
  for (int rule : {0, 1, 3}) {
    TileSchedule cand;
    cand.init_phys = init;
    cand.keep_layout = keep;
    t_seg_rule = rule;
    const int rc = make_tile_schedule_rule(dtype, n, ops, count, mode, allow_2q, &cand, allow_permute);
    t_seg_rule = 0;
    QCHK(rc);
    if (!have || cand.steps.size() < best.steps.size()) {
      best = std::move(cand);
      have = true;
    }
  }
  *out = std::move(best);
  return QIP_OK;
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
    } else if (st.ops.size() > 1 && (mode & 16) && n > (uint32_t)kWideBits) {
      // wide tiles (mode bit 4): the plan of build_wide_segment — arrangements (register bits, lane map, quarter bits, buffer
      // layout of the transposition into them) and the gates in 13-bit tile-index space, in the order they are applied
      std::vector<const TileItem*> seg;
      for (uint64_t i : st.ops) seg.push_back(&items[i]);
      WidePlan<T> plan;
      QCHK(build_wide_segment<T>(n, seg, st.high, &plan, mode & 3));
      js += ",\"wide\":1,\"low\":[0,1,2,3,4," + std::to_string(plan.p5) + "],\"high\":[";
      for (size_t k = 0; k < plan.high.size(); ++k) js += (k ? "," : "") + std::to_string(plan.high[k]);
      js += "],\"order\":[";
      for (size_t k = 0; k < plan.order.size(); ++k) js += (k ? "," : "") + std::to_string(plan.order[k]);
      js += "],\"passes\":[";
      for (size_t pi = 0; pi < plan.passes.size(); ++pi) {
        const WidePass& ps = plan.passes[pi];
        if (pi) js += ",";
        js += "{\"first\":" + std::to_string(ps.first) + ",\"count\":" + std::to_string(ps.count) + ",\"transposed\":" + (ps.transposed ? "1" : "0") +
              ",\"q\":[" + std::to_string(ps.q[0]) + "," + std::to_string(ps.q[1]) + "],\"R\":[";
        for (int j = 0; j < kWideRegBits; ++j) js += (j ? "," : "") + std::to_string(ps.R[j]);
        js += "],\"L\":[";
        for (int k = 0; k < 8; ++k) js += (k ? "," : "") + std::to_string(ps.L[k]);
        js += "],\"bufpos\":[";
        for (int t = 0; t < kWideBits; ++t) js += (t ? "," : "") + std::to_string(ps.bufpos[t]);
        js += "]}";
      }
      js += "],\"gates\":[";
      for (size_t gi = 0; gi < plan.gates.size(); ++gi) {
        const TileGate<T>& g = plan.gates[gi];
        if (gi) js += ",";
        js += "{\"kind\":" + std::to_string(g.kind) + ",\"b0\":" + std::to_string(g.b0) + ",\"b1\":" + std::to_string(g.b1) + ",\"cmask\":" +
              std::to_string(g.cmask) + ",\"omask\":" + std::to_string(g.omask) + ",\"tpos_out\":" + std::to_string(g.tpos_out) + ",\"nz\":" +
              std::to_string(g.nz) + ",\"m\":[";
        for (int e = 0; e < 4; ++e)
          js += std::string(e ? "," : "") + "[" + num((double)g.m[e].x) + "," + num((double)g.m[e].y) + "]";
        js += "]}";
      }
      js += "],\"mats\":[";
      for (size_t e = 0; e < plan.mats.size(); ++e)
        js += std::string(e ? "," : "") + "[" + num((double)plan.mats[e].x) + "," + num((double)plan.mats[e].y) + "]";
      js += "]";
    } else if (st.ops.size() > 1) {
      std::vector<const TileItem*> seg;
      for (uint64_t i : st.ops) seg.push_back(&items[i]);
      TileSegmentPlan<T> plan;
      QCHK(build_tile_segment<T>(n, true, seg, st.high, &plan, mode & 3));
      js += ",\"low\":[0,1,2,3,4," + std::to_string(plan.p5) + "],\"high\":[";
      for (size_t k = 0; k < plan.high.size(); ++k) js += (k ? "," : "") + std::to_string(plan.high[k]);
      js += "],\"order\":[";
      for (size_t k = 0; k < plan.order.size(); ++k) js += (k ? "," : "") + std::to_string(plan.order[k]);
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
      if (mode & 1024) {  // r5: what the interpreter kernel is handed instead — runs of diagonal gates as TileDiagItem steps
        TileInterpPlan<T> ip;
        tile_merge_diag_runs<T>(plan, &ip);
        js += ",\"interp\":{\"runs\":" + std::to_string(ip.runs) + ",\"gates_in_runs\":" + std::to_string(ip.gates_in_runs) + ",\"passes\":[";
        for (uint32_t pi = 0; pi < ip.pd.npasses; ++pi)
          js += std::string(pi ? "," : "") + "[" + std::to_string(ip.pd.pass[pi].first) + "," + std::to_string(ip.pd.pass[pi].count) + "]";
        js += "],\"gates\":[";
        for (size_t gi = 0; gi < ip.gates.size(); ++gi) {
          const TileGate<T>& g = ip.gates[gi];
          if (gi) js += ",";
          if (g.op == (uint32_t)TOP_DIAG_RUN) {
            js += "{\"run\":[" + std::to_string(g.nz) + "," + std::to_string(g.b1) + "]}";
          } else {
            size_t src = 0;  // an unchanged gate: its index in the plan's list
            for (; src < plan.gates.size(); ++src)
              if (!memcmp(&plan.gates[src], &g, sizeof g)) break;
            js += "{\"gate\":" + std::to_string(src) + "}";
          }
        }
        js += "],\"items\":[";
        for (size_t k = 0; k < ip.items.size(); ++k) {
          const TileDiagItem<T>& it = ip.items[k];
          js += std::string(k ? "," : "") + "{\"f0\":[" + num((double)it.f0.x) + "," + num((double)it.f0.y) + "],\"f1\":[" + num((double)it.f1.x) + "," +
                num((double)it.f1.y) + "],\"lane\":[" + std::to_string(it.lane_mask) + "," + std::to_string(it.lane_val) + "],\"reg\":[" +
                std::to_string(it.reg_pack & 0xffffu) + "," + std::to_string(it.reg_pack >> 16) + "],\"out\":[" + std::to_string(it.omask) + "," +
                std::to_string(it.oval) + "],\"sel\":" + std::to_string((it.emask_sel >> 24) ? 1u << ((it.emask_sel >> 24) - 1u) : 0u) +
                ",\"emask\":" + std::to_string(it.emask_sel & 0xffu) + "}";
        }
        js += "]}";
      }
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

template int build_tile_segment<double>(uint32_t, bool, const std::vector<const TileItem*>&, std::vector<uint32_t>, TileSegmentPlan<double>*, int, uint32_t);
template int build_tile_segment<float>(uint32_t, bool, const std::vector<const TileItem*>&, std::vector<uint32_t>, TileSegmentPlan<float>*, int, uint32_t);
