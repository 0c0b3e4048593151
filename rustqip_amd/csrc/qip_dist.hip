// qip_dist.hip — the state sharded over several GPUs (include/qip_hip.h, "qip_hip_dist_*"; SURVEY.md §8 row e).
// Uses the launch helpers of qip_launch.hip through qip_internal.h.
//
// Three layers, so that each can be tested where it can run:
//   DistPlanner   pure host code: logical -> physical qubit map, which qubits go global at a remap (farthest next use; r4: by
//                 modelled cost when that set would gather from inside a wave row),
//                 the op each rank applies to its shard.  Emits a list of steps (local op / pack / exchange).
//                 qip_hip_dist_debug_plan serialises it: the CPU tests replay it with the oracle as the shard and gloo
//                 as the transport (tests/test_distributed_cpu.py).
//   executor      runs the steps on a qip_hip_state shard: qip_hip_state_apply_ops for the runs of local ops, one
//                 out-of-place bit-permutation sweep (k_pack_bits) to gather the outgoing qubits into the top local
//                 positions, the transport's all-to-all for the exchange.
//   transport     RCCL (ncclGroupStart; ncclSend / ncclRecv x (G-1); ncclGroupEnd on the handle's stream; librccl is
//                 dlopen-ed on first use, so the single-GPU library has no link-time dependency on it) or a caller-
//                 supplied pair of callbacks.
//
// Why the exchange always moves ALL g rank bits: xGMI is point-to-point (7 links x ~153 GB/s per GPU).  Exchanging all
// g bits sends 1/G of the shard to each of the G-1 peers over G-1 different links at once (16-GiB shards, G = 8:
// 2 GiB per link, ~14 ms); exchanging one bit sends half the shard to ONE peer over one link (~56 ms).  The full
// exchange is the cheapest one on this topology, and it buys g fresh local qubits instead of one.  What used to cost
// extra — up to g local swap sweeps to bring the outgoing qubits to the top positions — is now one pack sweep.
#include "qip_tile.h"
#include <unistd.h>
#include <dlfcn.h>

#include <map>
#include <memory>
#include <mutex>

int64_t g_dist_plan_cost = 1;  // global option "dist_plan_cost": 0 = the leaving qubits by exchange count alone (rounds 1-3)
int64_t g_dist_fold_pack = 1;  // global option "dist_fold_pack": 0 = the remap always gathers with a sweep of its own

namespace qipd {

// r6, the collective watchdog: a rank whose peers never arrive (one of them died, or issued different calls) must FAIL, not hang
// the launcher.  Every place where the host waits for work that may contain an exchange or an all-reduce waits through this:
// poll the stream, and give up after "collective_timeout_s" seconds (global option, default 120; 0 = plain hipStreamSynchronize).
static int wait_stream(hipStream_t stream, int rank, int world, const char* what) {
  if (world <= 1 || g_collective_timeout_s <= 0) {
    HIPCHK(hipStreamSynchronize(stream));
    return QIP_OK;
  }
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 0;; ++spins) {
    const hipError_t e = hipStreamQuery(stream);
    if (e == hipSuccess) return QIP_OK;
    if (e != hipErrorNotReady) return fail(QIP_ERR_DEVICE, "hipStreamQuery failed: %s (rank %d, %s)", hipGetErrorString(e), rank, what);
    const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (waited > (double)g_collective_timeout_s)
      return fail(QIP_ERR_DEVICE, "rank %d of %d: %s did not complete within %lld s — a peer is missing, hung, or issued different calls "
                  "(global option collective_timeout_s)", rank, world, what, (long long)g_collective_timeout_s);
    if (waited > 2e-4) usleep(waited > 0.05 ? 1000 : 50);  // (spin for the first 200 us: a wait that short must not pay a sleep quantum)
  }
}

// ---- ops owned by the planner (the localized op of a rank) -------------------------------------------------------------
struct LocalOp {
  int32_t kind = QIP_OP_MATRIX;
  std::vector<uint64_t> indices;
  uint32_t n_controls = 0;
  std::vector<double> dense;  // re,im pairs, row-major (converted to the state's dtype when marshalled)
  std::vector<uint64_t> rowptr, cols;
  std::vector<double> vals;   // re,im pairs
  std::unique_ptr<LocalOp> inner;
};

// C descriptors of a batch of LocalOps (stable storage: pointers into the deque / owned buffers)
struct Marshalled {
  std::deque<qip_op> ops;                 // outer descriptors first (contiguous view built by `flat`)
  std::deque<std::vector<float>> f32;     // converted payloads
  std::vector<qip_op> flat;
};
static const qip_op* marshal_one(int dtype, const LocalOp& lo, Marshalled* m) {
  m->ops.emplace_back();
  qip_op* o = &m->ops.back();
  memset(o, 0, sizeof *o);
  o->kind = lo.kind;
  o->n_indices = (uint32_t)lo.indices.size();
  o->indices = lo.indices.data();
  o->n_controls = lo.n_controls;
  auto payload = [&](const std::vector<double>& v) -> const void* {
    if (dtype == QIP_C64) return v.data();
    m->f32.emplace_back(v.begin(), v.end());
    return m->f32.back().data();
  };
  if (lo.kind == QIP_OP_MATRIX) o->dense = payload(lo.dense);
  if (lo.kind == QIP_OP_SPARSE) {
    o->sparse_rowptr = lo.rowptr.data();
    o->sparse_cols = lo.cols.data();
    o->sparse_vals = payload(lo.vals);
  }
  if (lo.kind == QIP_OP_CONTROL) o->inner = marshal_one(dtype, *lo.inner, m);
  return o;
}

// ---- what the planner needs to know about an op -------------------------------------------------------------------------
struct OpInfo {
  FlatOp f;
  std::vector<uint32_t> ctrl, tgt;   // qubit indices
  std::vector<char> diag;            // diag[j]: the op never changes target j's bit (block-diagonal in it)
  std::vector<uint32_t> nondiag_bits;  // logical bit positions (n-1-q) that must be physically local
  // an uncontrolled anti-diagonal 1-qubit gate [[0, a], [b, 0]] (X, Y, ...): on a rank bit it only renames the ranks
  bool antidiag1q = false;
  bool relabel_swap = false;  // an uncontrolled Swap: served by exchanging entries of the logical -> physical map
  double a_re = 0, a_im = 0, b_re = 0, b_im = 0;
};

template <typename T> static inline void entry_of(const void* p, uint64_t e, double* re, double* im) {
  const T* d = static_cast<const T*>(p);
  *re = (double)d[2 * e];
  *im = (double)d[2 * e + 1];
}
static inline void entry(int dtype, const void* p, uint64_t e, double* re, double* im) {
  if (dtype == QIP_C64) entry_of<double>(p, e, re, im);
  else entry_of<float>(p, e, re, im);
}

static int analyse(uint32_t n, int dtype, const qip_op* op, OpInfo* info) {
  QCHK(flatten_op(n, op, false, &info->f));
  const FlatOp& f = info->f;
  if (!f.distinct) return fail(QIP_ERR_INVALID, "a sharded state needs distinct qubit indices in an op");
  for (uint32_t j = 0; j < f.n_control; ++j) info->ctrl.push_back((uint32_t)f.outer->indices[j]);
  for (uint32_t j = f.n_control; j < f.k_all; ++j) info->tgt.push_back((uint32_t)f.outer->indices[j]);
  const uint32_t k = f.n_op;
  info->diag.assign(k, 0);
  uint64_t diff = 0;  // OR of (row ^ col) over the non-zero entries: bits the op can change
  if (f.inner->kind == QIP_OP_SWAP || (f.inner->kind == QIP_OP_MATRIX && k > 10)) {
    diff = ~0ull;
  } else if (f.inner->kind == QIP_OP_MATRIX) {
    const uint64_t side = 1ull << k;
    for (uint64_t r = 0; r < side; ++r)
      for (uint64_t c = 0; c < side; ++c) {
        double re, im;
        entry(dtype, f.inner->dense, r * side + c, &re, &im);
        if (re != 0.0 || im != 0.0) diff |= r ^ c;
      }
  } else {
    const uint64_t rows = 1ull << k;
    for (uint64_t r = 0; r < rows; ++r)
      for (uint64_t p = f.inner->sparse_rowptr[r]; p < f.inner->sparse_rowptr[r + 1]; ++p) {
        double re, im;
        entry(dtype, f.inner->sparse_vals, p, &re, &im);
        if (re != 0.0 || im != 0.0) diff |= r ^ f.inner->sparse_cols[p];
      }
  }
  for (uint32_t j = 0; j < k; ++j) {
    info->diag[j] = !((diff >> (k - 1 - j)) & 1ull);
    if (!info->diag[j]) info->nondiag_bits.push_back(n - 1 - info->tgt[j]);
  }
  info->relabel_swap = f.inner->kind == QIP_OP_SWAP && f.n_control == 0;
  if (f.inner->kind == QIP_OP_MATRIX && k == 1 && f.n_control == 0) {
    double d0r, d0i, d1r, d1i;
    entry(dtype, f.inner->dense, 0, &d0r, &d0i);
    entry(dtype, f.inner->dense, 3, &d1r, &d1i);
    entry(dtype, f.inner->dense, 1, &info->a_re, &info->a_im);
    entry(dtype, f.inner->dense, 2, &info->b_re, &info->b_im);
    info->antidiag1q = d0r == 0.0 && d0i == 0.0 && d1r == 0.0 && d1i == 0.0;
  }
  return QIP_OK;
}

// ---- steps ----------------------------------------------------------------------------------------------------------------
struct Step {
  enum Kind { LOCAL, PACK, EXCHANGE } kind = LOCAL;
  std::unique_ptr<LocalOp> op;   // LOCAL
  std::vector<uint32_t> sel;     // PACK: local bit positions that become local positions L-g .. L-1, in this order
};

struct DistPlanner {
  uint32_t n = 0, g = 0, L = 0;
  int rank = 0, world = 1, dtype = QIP_C64;
  std::vector<uint32_t> phys;       // phys[p] = physical bit position of logical bit position p
  std::vector<uint64_t> last_use;   // clock of the last op that touched logical bit p
  uint64_t clock = 0;
  // Rank bit j (physical position L + j) reads as its physical value XOR bit j of `flip`.  An X (any anti-diagonal 1-qubit
  // gate) on a qubit that lives on a rank bit exchanges the two halves of the ranks wholesale: instead of moving 2^L
  // amplitudes between every pair of ranks, the ranks trade NAMES (and scale their shard by the gate's entry).  The
  // pending flips travel with the next exchange: the bits that come down to local positions are then X-ed locally.
  uint32_t flip = 0;

  int init(uint32_t n_, int dtype_, int rank_, int world_) {
    n = n_;
    dtype = dtype_;
    rank = rank_;
    world = world_;
    g = 0;
    while ((1 << g) < world) ++g;
    if ((1 << g) != world) return fail(QIP_ERR_INVALID, "world size %d is not a power of two", world);
    if (rank < 0 || rank >= world) return fail(QIP_ERR_INVALID, "rank %d out of range for world %d", rank, world);
    if (n < g + std::max(g, 1u)) return fail(QIP_ERR_INVALID, "n = %u is too small to shard over %d ranks", n, world);
    L = n - g;
    phys.resize(n);
    for (uint32_t p = 0; p < n; ++p) phys[p] = p;
    last_use.assign(n, 0);
    clock = 0;
    flip = 0;
    return QIP_OK;
  }
  uint32_t logical_at(uint32_t pp) const {
    for (uint32_t p = 0; p < n; ++p)
      if (phys[p] == pp) return p;
    return n;
  }
  uint32_t rank_bit(uint32_t pp) const { return ((((uint32_t)rank) ^ flip) >> (pp - L)) & 1u; }  // the LOGICAL value
  uint64_t local_qubit(uint32_t pp) const { return L - 1 - pp; }  // local qubit index of local physical bit pp

  // ---- which g logical bits leave at a remap ------------------------------------------------------------------------------------
  // Farthest next use first (never used again = infinitely far: the choice that minimises the NUMBER of exchanges), then least
  // recently used; among equals a qubit that already sits in the top g local positions (no gather at all if all g do), then one
  // on a high bit position (the gather then moves long contiguous runs).
  // r4, by cost: the exchange is the expensive part (shard / world bytes per link), but the gather before it is a full sweep of
  // the shard (k_pack_bits; k_permute_bits when a selected position lies inside a wave row) UNLESS the tile sweep before it can
  // store its tiles packed (TileStorePerm) — which needs every selected position outside the tile's row positions.  When the
  // count-optimal choice would gather from a row position, the same rule restricted to the other positions is tried as well;
  // both are rolled forward over the rest of the circuit (layout only: microseconds) and the cheaper modelled total is kept.
  struct Choice {
    std::vector<uint32_t> leaving;  // logical bit positions, leaving[t] goes to slot slot_of[t]
    std::vector<uint32_t> slot_of;
    std::vector<uint32_t> sel;      // PACK: sel[slot] = local physical position gathered into local position L-g+slot
    bool need_pack = false;
    bool rows = false;              // a gathered position is one of the tile's row positions: the gather cannot ride in a sweep
  };
  uint32_t row_p5() const { return tile_p5(dtype, L); }
  bool choose(const std::vector<uint32_t>& ph, const std::vector<uint64_t>& lu, const std::vector<uint32_t>& must,
              const std::vector<uint64_t>* next_use, bool avoid_rows, Choice* c) const {
    std::vector<uint32_t> cand;
    const uint32_t p5 = row_p5();
    for (uint32_t p = 0; p < n; ++p)
      if (ph[p] < L && std::find(must.begin(), must.end(), p) == must.end() &&
          !(avoid_rows && ph[p] < L - g && tile_is_low(ph[p], p5)))
        cand.push_back(p);
    if (cand.size() < g) return false;
    auto nu = [&](uint32_t p) { return next_use ? (*next_use)[p] : ~0ull; };
    std::stable_sort(cand.begin(), cand.end(), [&](uint32_t a, uint32_t b) {
      if (nu(a) != nu(b)) return nu(a) > nu(b);
      if (lu[a] != lu[b]) return lu[a] < lu[b];
      return ph[a] > ph[b];
    });
    c->leaving.assign(cand.begin(), cand.begin() + g);
    // target slot L-g+t for leaving[t]: keep the ones already on top where they are
    c->slot_of.assign(g, n);
    std::vector<char> slot_taken(g, 0);
    for (uint32_t t = 0; t < g; ++t)
      if (ph[c->leaving[t]] >= L - g) {
        c->slot_of[t] = ph[c->leaving[t]] - (L - g);
        slot_taken[c->slot_of[t]] = 1;
      }
    c->need_pack = false;
    c->rows = false;
    for (uint32_t t = 0; t < g; ++t)
      if (c->slot_of[t] == n) {
        c->need_pack = true;
        c->rows = c->rows || tile_is_low(ph[c->leaving[t]], p5);
        for (uint32_t sl = 0; sl < g; ++sl)
          if (!slot_taken[sl]) {
            c->slot_of[t] = sl;
            slot_taken[sl] = 1;
            break;
          }
      }
    c->sel.assign(g, 0);
    for (uint32_t t = 0; t < g; ++t) c->sel[c->slot_of[t]] = ph[c->leaving[t]];
    return true;
  }
  // the layout after the gather (if any) and the exchange
  void layout_after(std::vector<uint32_t>& ph, const Choice& c) const {
    if (c.need_pack) {
      // the selected positions on top (in slot order), every other local bit keeps its relative order
      std::vector<uint32_t> newpos(L, 0);
      uint32_t next = 0;
      for (uint32_t pp = 0; pp < L; ++pp) {
        const auto it = std::find(c.sel.begin(), c.sel.end(), pp);
        if (it != c.sel.end()) newpos[pp] = L - g + (uint32_t)(it - c.sel.begin());
        else newpos[pp] = next++;
      }
      for (uint32_t p = 0; p < n; ++p)
        if (ph[p] < L) ph[p] = newpos[ph[p]];
    }
    // chunk c of rank r <-> chunk r of rank c: local bit L-g+j and rank bit L+j trade places
    std::vector<uint32_t> at(n, n);
    for (uint32_t p = 0; p < n; ++p) at[ph[p]] = p;
    for (uint32_t j = 0; j < g; ++j) {
      const uint32_t a = at[L - g + j], b = at[L + j];
      ph[a] = L + j;
      ph[b] = L - g + j;
    }
  }
  // modelled milliseconds: one exchange = shard / world bytes over each xGMI link at 153 GB/s; one gather that does
  // not ride in a sweep = one read + one write of the shard at the measured copy rate
  double exchange_ms() const { return 1e3 * ((double)(dtype == QIP_C64 ? 16 : 8) * (double)(1ull << L) / (double)world) / 153e9; }
  double pack_ms() const { return 1e3 * (2.0 * (double)(dtype == QIP_C64 ? 16 : 8) * (double)(1ull << L)) / 6.2e12; }
  // what the rest of the circuit costs in remaps when the remap before op `from` is chosen with / without the row positions
  // (later remaps: the count-optimal rule); layout only.  `prev_local`: a local op precedes (its sweep can carry the gather).
  const std::vector<OpInfo>* all_infos = nullptr;
  const std::vector<std::vector<uint64_t>>* all_next = nullptr;
  uint64_t cur_op = 0;
  static constexpr uint64_t kRolloutHorizon = 4096;
  double rollout(uint64_t from, bool first_avoid, bool prev_local, uint64_t* exchanges) const {
    std::vector<uint32_t> ph = phys;
    std::vector<uint64_t> lu = last_use;
    uint64_t clk = clock;
    double cost = 0;
    uint64_t ex = 0;
    const uint64_t end = std::min<uint64_t>(all_infos->size(), from + kRolloutHorizon);
    for (uint64_t i = from; i < end; ++i) {
      const OpInfo& info = (*all_infos)[i];
      if (i > from) clk += 1;  // (the caller's clock already counts op `from`)
      if (info.antidiag1q && ph[n - 1 - info.tgt[0]] >= L) {
        lu[n - 1 - info.tgt[0]] = clk;
        continue;
      }
      if (info.relabel_swap) {
        const uint32_t h = (uint32_t)info.tgt.size() / 2;
        for (uint32_t j = 0; j < h; ++j) std::swap(ph[n - 1 - info.tgt[j]], ph[n - 1 - info.tgt[h + j]]);
        continue;
      }
      bool needs = false;
      for (uint32_t p : info.nondiag_bits) needs = needs || ph[p] >= L;
      if (needs) {
        Choice c;
        if (!choose(ph, lu, info.nondiag_bits, &(*all_next)[i], i == from && first_avoid, &c)) return 1e300;
        ex += 1;
        cost += exchange_ms();
        if (c.need_pack && !(prev_local && !c.rows)) cost += pack_ms();
        layout_after(ph, c);
        prev_local = false;
      }
      for (uint32_t qb : info.ctrl) lu[n - 1 - qb] = clk;
      for (uint32_t qb : info.tgt) lu[n - 1 - qb] = clk;
      prev_local = true;
    }
    if (exchanges) *exchanges = ex;
    return cost;
  }

  // The exchange: choose the g logical bits that leave (must stay: `must`), gather them on top, all-to-all.
  int remap(const std::vector<uint32_t>& must, const std::vector<uint64_t>* next_use, std::vector<Step>* out) {
    Choice ch;
    if (!choose(phys, last_use, must, next_use, false, &ch))
      return fail(QIP_ERR_UNSUPPORTED, "op touches too many qubits to keep local on this shard size");
    if (ch.need_pack && ch.rows && all_infos && g_dist_plan_cost) {
      Choice alt;
      const bool prev_local = !out->empty() && out->back().kind == Step::LOCAL;
      if (choose(phys, last_use, must, next_use, true, &alt) && !alt.rows) {
        const double cost_a = rollout(cur_op, false, prev_local, nullptr), cost_b = rollout(cur_op, true, prev_local, nullptr);
        if (cost_b < cost_a) ch = alt;
      }
    }
    if (ch.need_pack) {
      Step st;
      st.kind = Step::PACK;
      st.sel = ch.sel;
      out->push_back(std::move(st));
    }
    layout_after(phys, ch);
    Step ex;
    ex.kind = Step::EXCHANGE;
    out->push_back(std::move(ex));
    // the old rank bits are local positions L-g .. L-1 now and hold the senders' PHYSICAL rank values: where a flip was
    // pending, physical and logical differ — one local X puts that right; the new rank bits start unflipped
    for (uint32_t j = 0; j < g; ++j)
      if ((flip >> j) & 1u) {
        Step st;
        st.kind = Step::LOCAL;
        st.op.reset(new LocalOp());
        st.op->kind = QIP_OP_MATRIX;
        st.op->indices.push_back(local_qubit(L - g + j));
        st.op->dense = {0.0, 0.0, 1.0, 0.0, 1.0, 0.0, 0.0, 0.0};
        out->push_back(std::move(st));
      }
    flip = 0;
    return QIP_OK;
  }

  // The op this rank runs on its shard (local qubit indices); null when it is the identity here.
  int localize(const OpInfo& info, std::unique_ptr<LocalOp>* out) {
    out->reset();
    const FlatOp& f = info.f;
    std::vector<uint64_t> local_ctrl;
    for (uint32_t qb : info.ctrl) {
      const uint32_t pp = phys[n - 1 - qb];
      if (pp >= L) {
        if (rank_bit(pp) == 0) return QIP_OK;  // a control on a rank bit that reads 0 here: identity on this shard
      } else {
        local_ctrl.push_back(local_qubit(pp));
      }
    }
    const uint32_t k = (uint32_t)info.tgt.size();
    std::vector<uint32_t> tpp(k);
    std::vector<uint32_t> glob, keep;
    for (uint32_t j = 0; j < k; ++j) {
      tpp[j] = phys[n - 1 - info.tgt[j]];
      (tpp[j] >= L ? glob : keep).push_back(j);
    }
    std::unique_ptr<LocalOp> loc(new LocalOp());
    if (glob.empty()) {
      loc->kind = f.inner->kind;
      for (uint32_t j = 0; j < k; ++j) loc->indices.push_back(local_qubit(tpp[j]));
      if (f.inner->kind == QIP_OP_MATRIX) {
        const uint64_t cnt = 1ull << (2 * k);
        loc->dense.resize(2 * cnt);
        for (uint64_t e = 0; e < cnt; ++e) entry(dtype, f.inner->dense, e, &loc->dense[2 * e], &loc->dense[2 * e + 1]);
      } else if (f.inner->kind == QIP_OP_SPARSE) {
        const uint64_t rows = 1ull << k, nnz = f.inner->sparse_rowptr[rows];
        loc->rowptr.assign(f.inner->sparse_rowptr, f.inner->sparse_rowptr + rows + 1);
        loc->cols.assign(f.inner->sparse_cols, f.inner->sparse_cols + nnz);
        loc->vals.resize(2 * nnz);
        for (uint64_t e = 0; e < nnz; ++e) entry(dtype, f.inner->sparse_vals, e, &loc->vals[2 * e], &loc->vals[2 * e + 1]);
      }
    } else {
      // diagonal targets on rank bits: keep the block of the matrix selected by this rank's bits
      if (f.inner->kind == QIP_OP_SWAP) return fail(QIP_ERR_INVALID, "internal: swap target left on a rank bit");
      const uint64_t side = 1ull << k;
      auto at = [&](uint64_t r, uint64_t c, double* re, double* im) {
        *re = *im = 0.0;
        if (f.inner->kind == QIP_OP_MATRIX) {
          entry(dtype, f.inner->dense, r * side + c, re, im);
        } else {
          for (uint64_t p = f.inner->sparse_rowptr[r]; p < f.inner->sparse_rowptr[r + 1]; ++p)
            if (f.inner->sparse_cols[p] == c) {
              double a, b;
              entry(dtype, f.inner->sparse_vals, p, &a, &b);
              *re += a;
              *im += b;
            }
        }
      };
      const uint32_t kk = (uint32_t)keep.size();
      std::vector<uint64_t> sel(1ull << kk);
      for (uint64_t sidx = 0; sidx < sel.size(); ++sidx) {
        uint64_t full = 0;
        for (uint32_t j = 0; j < k; ++j) {
          uint64_t bit;
          const auto it = std::find(keep.begin(), keep.end(), j);
          if (it == keep.end()) bit = rank_bit(tpp[j]);
          else bit = (sidx >> (kk - 1 - (uint32_t)(it - keep.begin()))) & 1ull;
          full |= bit << (k - 1 - j);
        }
        sel[sidx] = full;
      }
      if (kk > 0) {
        loc->kind = QIP_OP_MATRIX;
        for (uint32_t j : keep) loc->indices.push_back(local_qubit(tpp[j]));
        loc->dense.resize(2 * sel.size() * sel.size());
        for (uint64_t r = 0; r < sel.size(); ++r)
          for (uint64_t c = 0; c < sel.size(); ++c) at(sel[r], sel[c], &loc->dense[2 * (r * sel.size() + c)], &loc->dense[2 * (r * sel.size() + c) + 1]);
      } else {
        double re, im;
        at(sel[0], sel[0], &re, &im);
        if (re == 1.0 && im == 0.0) return QIP_OK;  // multiplies this shard by one
        loc->kind = QIP_OP_MATRIX;
        if (!local_ctrl.empty()) {  // a phase on the all-controls-one subspace: the last control becomes the target
          loc->indices.push_back(local_ctrl.back());
          local_ctrl.pop_back();
          loc->dense = {1.0, 0.0, 0.0, 0.0, 0.0, 0.0, re, im};
        } else {  // a global factor on this shard
          loc->indices.push_back(0);
          loc->dense = {re, im, 0.0, 0.0, 0.0, 0.0, re, im};
        }
      }
    }
    if (!local_ctrl.empty()) {
      std::unique_ptr<LocalOp> c(new LocalOp());
      c->kind = QIP_OP_CONTROL;
      c->n_controls = (uint32_t)local_ctrl.size();
      c->indices = local_ctrl;
      for (uint64_t t : loc->indices) c->indices.push_back(t);
      c->inner = std::move(loc);
      loc = std::move(c);
    }
    *out = std::move(loc);
    return QIP_OK;
  }

  // one op: remap first if it has to be, then the localized op
  int step(const OpInfo& info, const std::vector<uint64_t>* next_use, std::vector<Step>* out) {
    clock += 1;
    if (info.antidiag1q && phys[n - 1 - info.tgt[0]] >= L) {
      // new[v'] = M[v'][v] * old[v] with v' = v ^ 1: this rank's shard keeps its amplitudes, answers to the other value
      // of the bit from now on, and is scaled by the one non-zero entry of its new row (b for v' = 1, a for v' = 0)
      const uint32_t pp = phys[n - 1 - info.tgt[0]];
      flip ^= 1u << (pp - L);
      const bool one = rank_bit(pp) != 0;
      const double re = one ? info.b_re : info.a_re, im = one ? info.b_im : info.a_im;
      last_use[n - 1 - info.tgt[0]] = clock;
      if (!(re == 1.0 && im == 0.0)) {
        Step st;
        st.kind = Step::LOCAL;
        st.op.reset(new LocalOp());
        st.op->kind = QIP_OP_MATRIX;
        st.op->indices.push_back(0);
        st.op->dense = {re, im, 0.0, 0.0, 0.0, 0.0, re, im};
        out->push_back(std::move(st));
      }
      return QIP_OK;
    }
    if (info.relabel_swap) {
      // An uncontrolled Swap(h, A ++ B) is a permutation of the index bits and nothing else (SwapOpIterator,
      // qubit_iterators.rs:176-219: one (col, 1) per row): with a logical -> physical map already kept per qubit, the two
      // halves trade physical positions and no amplitude moves — wherever the qubits live, rank bits included (a pending
      // rank renaming belongs to the physical bit and stays with it).
      const uint32_t h = (uint32_t)info.tgt.size() / 2;
      for (uint32_t j = 0; j < h; ++j) std::swap(phys[n - 1 - info.tgt[j]], phys[n - 1 - info.tgt[h + j]]);
      return QIP_OK;
    }
    bool needs = false;
    for (uint32_t p : info.nondiag_bits) needs = needs || phys[p] >= L;
    if (needs) QCHK(remap(info.nondiag_bits, next_use, out));
    for (uint32_t qb : info.ctrl) last_use[n - 1 - qb] = clock;
    for (uint32_t qb : info.tgt) last_use[n - 1 - qb] = clock;
    std::unique_ptr<LocalOp> loc;
    QCHK(localize(info, &loc));
    if (loc) {
      Step st;
      st.kind = Step::LOCAL;
      st.op = std::move(loc);
      out->push_back(std::move(st));
    }
    return QIP_OK;
  }

  // a whole circuit: next_use[i][p] = first op index >= i that needs logical bit p local (~0 = never)
  int plan(const qip_op* ops, uint64_t count, std::vector<Step>* out) {
    std::vector<OpInfo> infos(count);
    for (uint64_t i = 0; i < count; ++i) {
      int rc = analyse(n, dtype, &ops[i], &infos[i]);
      if (rc != QIP_OK) {
        std::string msg = g_last_error;
        return fail(rc, "op %llu: %s", (unsigned long long)i, msg.c_str());
      }
    }
    std::vector<std::vector<uint64_t>> nxt(count);
    std::vector<uint64_t> cur(n, ~0ull);
    for (uint64_t i = count; i-- > 0;) {
      if (!infos[i].antidiag1q && !infos[i].relabel_swap)  // (an X-like gate / a swap is served wherever its qubits live)
        for (uint32_t p : infos[i].nondiag_bits) cur[p] = i;
      nxt[i] = cur;
    }
    all_infos = &infos;
    all_next = &nxt;
    int rc = QIP_OK;
    for (uint64_t i = 0; i < count && rc == QIP_OK; ++i) {
      cur_op = i;
      rc = step(infos[i], &nxt[i], out);
    }
    all_infos = nullptr;
    all_next = nullptr;
    return rc;
  }
};

// ---- JSON of a plan (test hook) ---------------------------------------------------------------------------------------------
static void json_op(const LocalOp& o, std::string* js) {
  char buf[64];
  auto num = [&](double v) {
    snprintf(buf, sizeof buf, "%.17g", v);
    return std::string(buf);
  };
  static const char* names[] = {"Matrix", "SparseMatrix", "Swap", "Control"};
  *js += std::string("{\"kind\":\"") + names[o.kind] + "\",\"indices\":[";
  for (size_t i = 0; i < o.indices.size(); ++i) *js += (i ? "," : "") + std::to_string(o.indices[i]);
  *js += "]";
  if (o.kind == QIP_OP_MATRIX) {
    *js += ",\"data\":[";
    for (size_t e = 0; e < o.dense.size() / 2; ++e) *js += std::string(e ? "," : "") + "[" + num(o.dense[2 * e]) + "," + num(o.dense[2 * e + 1]) + "]";
    *js += "]";
  } else if (o.kind == QIP_OP_SPARSE) {
    *js += ",\"rows\":[";
    for (size_t r = 0; r + 1 < o.rowptr.size(); ++r) {
      *js += r ? ",[" : "[";
      for (uint64_t p = o.rowptr[r]; p < o.rowptr[r + 1]; ++p)
        *js += std::string(p > o.rowptr[r] ? "," : "") + "[" + std::to_string(o.cols[p]) + "," + num(o.vals[2 * p]) + "," + num(o.vals[2 * p + 1]) + "]";
      *js += "]";
    }
    *js += "]";
  } else if (o.kind == QIP_OP_SWAP) {
    *js += ",\"half\":" + std::to_string(o.indices.size() / 2);
  } else {
    *js += ",\"n_controls\":" + std::to_string(o.n_controls) + ",\"inner\":";
    json_op(*o.inner, js);
  }
  *js += "}";
}

// ---- transports -----------------------------------------------------------------------------------------------------------------
struct Rccl {
  typedef struct { char internal[QIP_HIP_UNIQUE_ID_BYTES]; } UniqueId;
  typedef void* Comm;
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
  int (*CommDestroy)(Comm) = nullptr;
  int (*CommCount)(Comm, int*) = nullptr;
  int (*CommUserRank)(Comm, int*) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  void* handle = nullptr;
  static constexpr int kUint8 = 1, kFloat64 = 8, kSum = 0;  // ncclDataType_t / ncclRedOp_t values (rccl.h:448-467)
};
static Rccl g_rccl;
static std::mutex g_rccl_mutex;  // handles are per thread (one thread per GPU is the natural in-process use): the first
                                 // multi-GPU call of each may arrive at the same time
static int rccl_load() {
  std::lock_guard<std::mutex> lock(g_rccl_mutex);
  if (g_rccl.handle) return QIP_OK;
  void* h = nullptr;
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (h) break;
  }
  if (!h) return fail(QIP_ERR_UNSUPPORTED, "cannot load librccl: %s", dlerror());
#define SYM(field, name)                                                           \
  do {                                                                             \
    *(void**)(&g_rccl.field) = dlsym(h, name);                                     \
    if (!g_rccl.field) return fail(QIP_ERR_UNSUPPORTED, "librccl lacks %s", name); \
  } while (0)
  SYM(GetUniqueId, "ncclGetUniqueId");
  SYM(CommInitRank, "ncclCommInitRank");
  SYM(CommDestroy, "ncclCommDestroy");
  SYM(CommCount, "ncclCommCount");
  SYM(CommUserRank, "ncclCommUserRank");
  SYM(GroupStart, "ncclGroupStart");
  SYM(GroupEnd, "ncclGroupEnd");
  SYM(Send, "ncclSend");
  SYM(Recv, "ncclRecv");
  SYM(AllReduce, "ncclAllReduce");
  SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
  g_rccl.handle = h;
  return QIP_OK;
}
#define NCHK(expr)                                                                                                   \
  do {                                                                                                               \
    int r_ = (expr);                                                                                                 \
    if (r_ != 0) return fail(QIP_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, qipd::g_rccl.GetErrorString(r_), __FILE__, __LINE__); \
  } while (0)

struct RcclTransport {
  Rccl::Comm comm = nullptr;
  int rank = 0, world = 1;
  double* d_red = nullptr;  // small device buffer for the all-reduce
  size_t red_cap = 0;
  hipStream_t stream = nullptr;
  uint64_t piece_bytes = 1ull << 30;  // option "piece_bytes"
  uint64_t pieces_sent = 0;
};

// How one rank's all-to-all is cut into sends: every peer in rank order, each chunk in pieces of at most `piece_bytes` (a
// 16-GiB shard over 8 ranks makes 2-GiB chunks: keep every count far from 2^31).  Several sends to one peer inside a group
// are matched with its receives in order, so both sides walk the same list.  Pure: qip_hip_dist_debug_pieces exports it.
struct Piece {
  int peer;
  uint64_t offset, length;
};
static std::vector<Piece> plan_pieces(int rank, int world, uint64_t chunk_bytes, uint64_t piece_bytes) {
  std::vector<Piece> out;
  if (piece_bytes == 0) piece_bytes = chunk_bytes ? chunk_bytes : 1;
  for (int p = 0; p < world; ++p) {
    if (p == rank) continue;
    for (uint64_t off = 0; off < chunk_bytes; off += piece_bytes) out.push_back({p, off, std::min<uint64_t>(piece_bytes, chunk_bytes - off)});
  }
  return out;
}

static int rccl_all_to_all(void* ctx, const void* send, void* recv, uint64_t chunk_bytes, void* stream) {
  RcclTransport* t = static_cast<RcclTransport*>(ctx);
  hipStream_t st = (hipStream_t)stream;
  if (t->world > 1) {
    // every peer: one send + one receive per piece, all in ONE group so that all links run at once.  An error inside the
    // group must not leave the communicator in group mode (a later collective would hang instead of failing): remember the
    // first one, always close the group, then report.
    int first = 0;
    const char* what = "";
    int r = g_rccl.GroupStart();
    if (r != 0) return fail(QIP_ERR_DEVICE, "ncclGroupStart failed: %s", g_rccl.GetErrorString(r));
    for (const Piece& pc : plan_pieces(t->rank, t->world, chunk_bytes, t->piece_bytes)) {
      r = g_rccl.Send((const char*)send + (size_t)pc.peer * chunk_bytes + pc.offset, (size_t)pc.length, Rccl::kUint8, pc.peer, t->comm, st);
      if (r != 0 && first == 0) first = r, what = "ncclSend";
      r = g_rccl.Recv((char*)recv + (size_t)pc.peer * chunk_bytes + pc.offset, (size_t)pc.length, Rccl::kUint8, pc.peer, t->comm, st);
      if (r != 0 && first == 0) first = r, what = "ncclRecv";
      if (first != 0) break;
      t->pieces_sent += 1;
    }
    r = g_rccl.GroupEnd();
    if (first != 0) return fail(QIP_ERR_DEVICE, "%s failed inside the exchange group: %s", what, g_rccl.GetErrorString(first));
    if (r != 0) return fail(QIP_ERR_DEVICE, "ncclGroupEnd failed: %s", g_rccl.GetErrorString(r));
  }
  // own chunk: a device copy
  HIPCHK(hipMemcpyAsync((char*)recv + (size_t)t->rank * chunk_bytes, (const char*)send + (size_t)t->rank * chunk_bytes, chunk_bytes,
                        hipMemcpyDeviceToDevice, st));
  return QIP_OK;
}
// r5: bytes [slice_off, slice_off + slice_bytes) of EVERY chunk — one group over all peers, so that all links run at once in every
// slice (cutting the exchange by peer instead would use the links one after the other); the slices of one remap are issued in order
// on the handle's communication stream and overlap with the tile sweeps on the shard's stream (dist_run_steps)
static int rccl_all_to_all_slice(void* ctx, const void* send, void* recv, uint64_t chunk_bytes, uint64_t slice_off, uint64_t slice_bytes, void* stream) {
  RcclTransport* t = static_cast<RcclTransport*>(ctx);
  hipStream_t st = (hipStream_t)stream;
  if (slice_off > chunk_bytes || slice_bytes > chunk_bytes - slice_off) return fail(QIP_ERR_INVALID, "slice outside the chunk");
  if (t->world > 1) {
    int first = 0;
    const char* what = "";
    int r = g_rccl.GroupStart();
    if (r != 0) return fail(QIP_ERR_DEVICE, "ncclGroupStart failed: %s", g_rccl.GetErrorString(r));
    for (const Piece& pc : plan_pieces(t->rank, t->world, slice_bytes, t->piece_bytes)) {
      const size_t at = (size_t)pc.peer * chunk_bytes + slice_off + pc.offset;
      r = g_rccl.Send((const char*)send + at, (size_t)pc.length, Rccl::kUint8, pc.peer, t->comm, st);
      if (r != 0 && first == 0) first = r, what = "ncclSend";
      r = g_rccl.Recv((char*)recv + at, (size_t)pc.length, Rccl::kUint8, pc.peer, t->comm, st);
      if (r != 0 && first == 0) first = r, what = "ncclRecv";
      if (first != 0) break;
      t->pieces_sent += 1;
    }
    r = g_rccl.GroupEnd();
    if (first != 0) return fail(QIP_ERR_DEVICE, "%s failed inside the exchange group: %s", what, g_rccl.GetErrorString(first));
    if (r != 0) return fail(QIP_ERR_DEVICE, "ncclGroupEnd failed: %s", g_rccl.GetErrorString(r));
  }
  const size_t own = (size_t)t->rank * chunk_bytes + slice_off;
  HIPCHK(hipMemcpyAsync((char*)recv + own, (const char*)send + own, slice_bytes, hipMemcpyDeviceToDevice, st));
  return QIP_OK;
}
static int rccl_all_reduce_on(RcclTransport* t, double* values, uint64_t count, hipStream_t stream);
static int rccl_all_reduce(void* ctx, double* values, uint64_t count) {
  RcclTransport* t = static_cast<RcclTransport*>(ctx);
  return rccl_all_reduce_on(t, values, count, t->stream);
}
// (r5: the ranks' agreement on where an overlapped exchange is cut runs on the COMMUNICATION stream — waiting for it does not drain
// the sweeps queued on the shard's stream)
static int rccl_all_reduce_on(RcclTransport* t, double* values, uint64_t count, hipStream_t stream) {
  if (count == 0) return QIP_OK;
  if (count > t->red_cap) {
    if (t->d_red) HIPCHK(hipFree(t->d_red));
    t->d_red = nullptr;
    t->red_cap = 0;
    const size_t cap = std::max<size_t>(count, 4096);
    HIPCHK(hipMalloc((void**)&t->d_red, cap * sizeof(double)));
    t->red_cap = cap;
  }
  HIPCHK(hipMemcpyAsync(t->d_red, values, count * sizeof(double), hipMemcpyHostToDevice, stream));
  NCHK(g_rccl.AllReduce(t->d_red, t->d_red, count, Rccl::kFloat64, Rccl::kSum, t->comm, stream));
  HIPCHK(hipMemcpyAsync(values, t->d_red, count * sizeof(double), hipMemcpyDeviceToHost, stream));
  return wait_stream(stream, t->rank, t->world, "an all-reduce");
}

}  // namespace qipd

// ---- the pack sweep: gather g local bit positions into the top g positions, out of place ------------------------------------
// dst index j = (c << (L-g)) | rest  <-  src index = insert_bits(rest, sel sorted) | bits of c spread over sel (slot order).
// Reads follow the source's runs (contiguous for 2^(lowest selected position) amplitudes), writes are fully contiguous.
namespace qipk {
struct PackDesc {
  uint32_t g, Lg;      // Lg = L - g
  uint32_t sel[8];     // sel[t]: source position of destination bit Lg + t
};
template <typename T>
__global__ __launch_bounds__(kBlock) void k_pack_bits(const amp_t<T>* __restrict__ in, amp_t<T>* __restrict__ out, uint64_t namps,
                                                     Ins ins, PackDesc d) {
  const uint64_t blk = blockIdx.x + (uint64_t)blockIdx.y * gridDim.x;
  // 4 destination amplitudes per lane, 1 KiB wave rows 4 KiB apart per step
  const uint64_t j0 = blk * (kBlock * 4) + threadIdx.x;
  amp_t<T> v[4];
  uint64_t jj[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    jj[u] = j0 + (uint64_t)u * kBlock;
    if (jj[u] < namps) {
      const uint64_t rest = jj[u] & ((1ull << d.Lg) - 1ull), c = jj[u] >> d.Lg;
      uint64_t src = insert_bits<-1>(rest, ins);
      for (uint32_t t = 0; t < d.g; ++t) src |= ((c >> t) & 1ull) << d.sel[t];
      v[u] = __builtin_nontemporal_load(in + src);
    }
  }
#pragma unroll
  for (int u = 0; u < 4; ++u)
    if (jj[u] < namps) __builtin_nontemporal_store(v[u], out + jj[u]);
}
}  // namespace qipk

// ---- the handle -----------------------------------------------------------------------------------------------------------------
struct qip_hip_dist {
  qipd::DistPlanner pl;
  qip_hip_state* shard = nullptr;
  qip_hip_transport transport{};
  qipd::RcclTransport* rccl = nullptr;  // owned when the built-in transport is in use
  qip_hip_dist_stats stats{};
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_exchange, ev_pack;
  // A batch that fails half way (a kernel launch, the transport, an allocation) leaves the shard somewhere between two
  // layouts while the planner already describes the end of the batch — and the other ranks may sit in a collective.  Nothing
  // computed from such a handle can be trusted: it is poisoned and every later call fails with the original message.
  bool poisoned = false;
  std::string poison_msg;
  // r5: the exchange overlapped with the tile sweeps either side of it (dist_run_steps): cut into `overlap` slices (option
  // "dist_overlap": 0 / 1 = off, 2 / 4 / 8), each issued on `comm_stream` as soon as the sweep before the remap has stored it,
  // the sweep after the remap starting on a slice as soon as it has landed.  `third`: the receive buffer of a remap whose packed
  // store and exchange are both in flight (source, packed copy and receive buffer are three different arrays then).
  int64_t overlap = 0;
  qip_hip_all_to_all_slice_fn slice_fn = nullptr;
  hipStream_t comm_stream = nullptr;
  std::vector<hipEvent_t> ev_pre, ev_rx;
  void* third = nullptr;
  bool rx_pending = false;  // slices of the last overlapped exchange may still be landing: the shard's stream has not waited yet
  uint32_t rx_slices = 0;
  size_t agreed_for = (size_t)-1;   // index of the EXCHANGE step the ranks have already voted on (dist_run_steps)
  std::vector<uint32_t> rx_packed;  // the positions that cut it
  bool rx_after = false;            // ... and whether the first sweep of the next batch can take them
};


// ---- r5: the overlapped exchange — what the tile sweeps either side of a remap look like, and where the exchange can be cut ----
// The exchange is cut into 2^p slices by p index positions of the buffer that is exchanged ("packed" positions, below the g chunk-
// selecting ones); a sweep can run slice by slice when NONE of those positions is a tile position of it.  The edge sweeps of the two
// batches are found by scheduling them exactly as apply_ops will (host work, no device), then the p highest packed positions that the
// LAST sweep before the remap leaves alone — preferring those the FIRST sweep after it leaves alone too — are taken.  A slice is then
// 2^(L-g-p-q0) runs of 2^q0 consecutive amplitudes per chunk (q0 = the lowest chosen position, >= kSliceMinPos): one transport call
// per run, every call a group over all peers.  Shared by the executor (dist_run_steps) and the host-only predicate
// (qip_hip_dist_debug_overlap), so what the model prices is what runs.
struct EdgeTile {
  bool segment = false;          // the step is a multi-gate tile sweep
  bool wide = false;
  std::vector<uint32_t> high;    // its tile's positions above the rows
  uint32_t p5 = 5;
  bool ends_in_callers_order = false, starts_plain = false;
  size_t nsteps = 0;
};
int state_tile_mode(const qip_hip_state* s);  // qip_circuit.hip
template <typename T>
static EdgeTile edge_tile_t(int dtype, uint32_t L, const qip_op* ops, uint64_t count, int tile_mode, bool last) {
  EdgeTile e;
  TileSchedule sc;
  if (L < (uint32_t)kTileBits || make_tile_schedule(dtype, L, ops, count, tile_mode & (1 | 2 | 4 | 8 | 16), true, &sc) != QIP_OK || sc.steps.empty()) return e;
  e.nsteps = sc.steps.size();
  e.ends_in_callers_order = true;
  for (uint32_t p = 0; p < sc.final_phys.size(); ++p) e.ends_in_callers_order = e.ends_in_callers_order && sc.final_phys[p] == p;
  e.starts_plain = sc.init_phys.empty() && sc.inserted == 0 && sc.absorbed == 0;
  const TileStep& st = last ? sc.steps.back() : sc.steps.front();
  if (st.ops.size() < 2 || !st.perm.empty()) return e;
  std::vector<const TileItem*> seg;
  for (uint64_t i : st.ops) seg.push_back(&sc.items[i]);
  if ((tile_mode & 16) && L > (uint32_t)kWideBits) {
    WidePlan<T> plan;
    if (build_wide_segment<T>(L, seg, st.high, &plan, tile_mode & 3) != QIP_OK) return e;
    e.high = plan.high;
    e.p5 = plan.p5;
    e.wide = true;
  } else {
    TileSegmentPlan<T> plan;
    if (build_tile_segment<T>(L, true, seg, st.high, &plan, tile_mode & 3) != QIP_OK) return e;
    e.high = plan.high;
    e.p5 = plan.p5;
  }
  e.segment = true;
  return e;
}
static EdgeTile edge_tile(int dtype, uint32_t L, const qip_op* ops, uint64_t count, int tile_mode, bool last) {
  (void)hipGetLastError();
  return dtype == QIP_C64 ? edge_tile_t<double>(dtype, L, ops, count, tile_mode, last) : edge_tile_t<float>(dtype, L, ops, count, tile_mode, last);
}
constexpr uint32_t kSliceMinPos = 12;  // (slice positions lie above the rows and above the split-row position 11)
// packed position q < L-g  <->  position of the buffer the sweep before the remap addresses (a gather in between moves it)
static uint32_t packed_to_source(uint32_t q, const std::vector<uint32_t>* sel, uint32_t L) {
  if (!sel) return q;
  uint32_t seen = 0;
  for (uint32_t pp = 0; pp < L; ++pp) {
    if (std::find(sel->begin(), sel->end(), pp) != sel->end()) continue;
    if (seen == q) return pp;
    ++seen;
  }
  return L;
}
static bool in_tile(const EdgeTile& e, uint32_t pos) { return tile_is_low(pos, e.p5) || std::find(e.high.begin(), e.high.end(), pos) != e.high.end(); }
// EVERY rank must cut a remap's exchange at the same positions, but the ranks' local batches differ (a control on a rank bit drops
// the op on half of them), so their edge sweeps hold different tiles: each rank states what it could live with — v[0] = "my last
// sweep can run in parts at all", v[1 + q] = "packed position q is outside its tile", v[1 + W + q] = "... and outside my first
// sweep's after the remap" — the vectors are summed over the ranks, and the choice is made from what ALL of them accept (identical
// input on every rank, so identical output): the p highest positions, preferring those the sweeps after the remap accept too.
static void slice_votes(uint32_t L, uint32_t g, const std::vector<uint32_t>* pack_sel, const EdgeTile& pre, const EdgeTile* post, bool usable,
                        std::vector<double>* v) {
  const uint32_t W = L - g;
  v->assign(1 + 2 * (size_t)W, 0.0);
  bool ok = usable && pre.segment && pre.ends_in_callers_order;
  if (ok && pack_sel)  // the gather must ride in that sweep: a packed store, which needs the rows to survive
    for (uint32_t p : *pack_sel) ok = ok && !tile_is_low(p, pre.p5);
  if (!ok) return;
  (*v)[0] = 1.0;
  const bool post_usable = post && post->segment && post->starts_plain;
  for (uint32_t q = kSliceMinPos; q < W; ++q) {
    const uint32_t src = packed_to_source(q, pack_sel, L);
    if (src >= L || src <= 11u || in_tile(pre, src)) continue;
    (*v)[1 + q] = 1.0;
    if (post_usable && !in_tile(*post, q)) (*v)[1 + W + q] = 1.0;
  }
}
// from the summed votes of `world` ranks: *packed = the p chosen packed positions (ascending); false = this remap runs the serial way
static bool choose_slices(uint32_t L, uint32_t g, uint32_t pbits, const std::vector<double>& sum, int world, std::vector<uint32_t>* packed) {
  const uint32_t W = L - g;
  packed->clear();
  if (sum.size() != 1 + 2 * (size_t)W || sum[0] != (double)world || W < pbits + kSliceMinPos) return false;
  for (int pass = 0; pass < 2; ++pass) {
    packed->clear();
    for (uint32_t q = W; q-- > kSliceMinPos && packed->size() < pbits;)
      if (sum[1 + (pass == 0 ? W : 0) + q] == (double)world) packed->push_back(q);
    if (packed->size() == pbits) break;
  }
  if (packed->size() != pbits) return false;
  std::sort(packed->begin(), packed->end());
  return true;
}
// the byte runs of slice k inside ONE chunk: offsets (bytes) and the common run length
static void slice_runs(uint32_t L, uint32_t g, const std::vector<uint32_t>& packed, uint32_t k, uint64_t amp_bytes, std::vector<uint64_t>* offsets,
                       uint64_t* run_bytes) {
  const uint32_t q0 = packed[0], top = L - g;
  std::vector<uint32_t> free_bits;  // the chunk-offset bits above q0 that are not slice positions
  for (uint32_t q = q0 + 1; q < top; ++q)
    if (std::find(packed.begin(), packed.end(), q) == packed.end()) free_bits.push_back(q);
  uint64_t fixed = 0;
  for (size_t j = 0; j < packed.size(); ++j) fixed |= (uint64_t)((k >> j) & 1u) << packed[j];
  offsets->clear();
  for (uint64_t c = 0; c < (1ull << free_bits.size()); ++c) {
    uint64_t off = fixed;
    for (size_t b = 0; b < free_bits.size(); ++b) off |= ((c >> b) & 1ull) << free_bits[b];
    offsets->push_back(off * amp_bytes);
  }
  *run_bytes = (1ull << q0) * amp_bytes;
}

static int dist_drain_events(qip_hip_dist* d) {
  qip_hip_state* s = d->shard;
  if (d->ev_exchange.empty() && d->ev_pack.empty()) return QIP_OK;
  if (d->comm_stream) QCHK(qipd::wait_stream(d->comm_stream, d->pl.rank, d->pl.world, "the exchange on the communication stream"));
  QCHK(qipd::wait_stream(s->stream, d->pl.rank, d->pl.world, "the queued batch (sweeps + exchange)"));
  for (auto* list : {&d->ev_exchange, &d->ev_pack}) {
    for (auto& e : *list) {
      float ms = 0;
      HIPCHK(hipEventElapsedTime(&ms, e.first, e.second));
      (list == &d->ev_exchange ? d->stats.exchange_ms : d->stats.pack_ms) += ms;
      (void)hipEventDestroy(e.first);
      (void)hipEventDestroy(e.second);
    }
    list->clear();
  }
  return QIP_OK;
}

template <typename F> static int dist_timed(qip_hip_dist* d, std::vector<std::pair<hipEvent_t, hipEvent_t>>* list, F&& fn) {
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0));
  HIPCHK(hipEventCreate(&e1));
  HIPCHK(hipEventRecord(e0, d->shard->stream));
  int rc = fn();
  if (rc == QIP_OK) {
    hipError_t e = hipEventRecord(e1, d->shard->stream);
    if (e != hipSuccess) rc = fail(QIP_ERR_DEVICE, "hipEventRecord failed: %s", hipGetErrorString(e));
  }
  if (rc != QIP_OK) {
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
  }
  list->push_back({e0, e1});
  if (list->size() > 1024) QCHK(dist_drain_events(d));
  return QIP_OK;
}

// the shard's stream waits for every slice of the last overlapped exchange that it has not waited for yet
static int dist_wait_rx(qip_hip_dist* d) {
  if (!d->rx_pending) return QIP_OK;
  for (uint32_t k = 0; k < d->rx_slices; ++k) HIPCHK(hipStreamWaitEvent(d->shard->stream, d->ev_rx[k], 0));
  d->rx_pending = false;
  return QIP_OK;
}
// sum over the ranks, without touching the shard's stream (RCCL: on the communication stream; a caller's transport: its host reduction)
static int dist_overlap_setup(qip_hip_dist* d, uint32_t P);
static int dist_agree(qip_hip_dist* d, std::vector<double>* v) {
  if (d->rccl) {
    QCHK(dist_overlap_setup(d, 1));
    return qipd::rccl_all_reduce_on(d->rccl, v->data(), v->size(), d->comm_stream);
  }
  const int rc = d->transport.all_reduce_sum(d->transport.ctx, v->data(), v->size());
  if (rc != 0) return g_last_error.empty() ? fail(QIP_ERR_DEVICE, "transport all_reduce_sum failed with status %d", rc) : QIP_ERR_DEVICE;
  return QIP_OK;
}
static int dist_overlap_setup(qip_hip_dist* d, uint32_t P) {
  if (!d->comm_stream) HIPCHK(hipStreamCreateWithFlags(&d->comm_stream, hipStreamNonBlocking));
  while (d->ev_pre.size() < P) {
    hipEvent_t a, b;
    HIPCHK(hipEventCreateWithFlags(&a, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&b, hipEventDisableTiming));
    d->ev_pre.push_back(a);
    d->ev_rx.push_back(b);
  }
  return QIP_OK;
}

static int dist_run_steps_inner(qip_hip_dist* d, std::vector<qipd::Step>& steps);
static int dist_run_steps(qip_hip_dist* d, std::vector<qipd::Step>& steps) {
  d->agreed_for = (size_t)-1;
  const int rc = dist_run_steps_inner(d, steps);
  // whatever comes next on the shard's stream (a download, a measurement, the next batch) is ordered after the last slices
  if (rc == QIP_OK) return dist_wait_rx(d);
  // (the requests point at objects of the failed batch's frame — none may outlive it)
  d->shard->slice_first = d->shard->slice_last = d->shard->slice_now = nullptr;
  d->shard->fold_request = nullptr;
  d->shard->fold_now = d->shard->fold_done = false;
  return rc;
}
static int dist_run_steps_inner(qip_hip_dist* d, std::vector<qipd::Step>& steps) {
  qip_hip_state* s = d->shard;
  const uint32_t g = d->pl.g, L = d->pl.L;
  size_t i = 0;
  while (i < steps.size()) {
    if (steps[i].kind == qipd::Step::LOCAL) {
      // the run of local ops up to the next remap goes to the shard as ONE batch (tile sweeps see it whole)
      qipd::Marshalled m;
      std::vector<const qip_op*> ptrs;
      size_t jx = i;
      for (; jx < steps.size() && steps[jx].kind == qipd::Step::LOCAL; ++jx) ptrs.push_back(qipd::marshal_one(s->dtype, *steps[jx].op, &m));
      m.flat.reserve(ptrs.size());
      for (const qip_op* p : ptrs) m.flat.push_back(*p);
      // r4: when a remap follows, its gather (the leaving qubits' positions to the top g local positions) rides in the store
      // phase of this batch's last tile sweep — every sweep writes whole rows anyway, so it writes them to their packed places
      // in the second buffer — and the PACK step is skipped.  Not when a gathered position lies inside a row (the rows would
      // break into 16-byte pieces: the bit-permutation sweep handles that), nor when the batch does not end in a tile sweep.
      TileStorePerm sp;
      memset(&sp, 0, sizeof sp);
      const bool pack_next = g_dist_fold_pack && jx < steps.size() && steps[jx].kind == qipd::Step::PACK && s->layout.empty();
      if (pack_next) {
        sp.g = g;
        sp.Lg = L - g;
        std::vector<uint32_t> desc = steps[jx].sel;
        std::sort(desc.begin(), desc.end(), std::greater<uint32_t>());
        for (uint32_t t = 0; t < g; ++t) {
          sp.sel[t] = steps[jx].sel[t];
          sp.sel_desc[t] = desc[t];
        }
        s->fold_request = &sp;
        s->fold_done = false;
      }
      // r5, option "dist_overlap": the exchange that follows this batch in P slices on the communication stream, slice k sent as
      // soon as the batch's LAST tile sweep — launched in P parts — has stored it; and the slices of the exchange BEFORE this
      // batch awaited one by one by its FIRST sweep (TileSlicing, qip_internal.h).  The slice bits are the index positions right
      // below the chunk-selecting ones of the buffer that is exchanged: packed positions L-g-p .. L-g-1.
      const uint32_t P = (d->overlap >= 2 && d->slice_fn && d->pl.world > 1 && s->tile >= 1 && s->tile_passes && !s->force_generic && !g_force_generic) ? (uint32_t)d->overlap : 0u;
      uint32_t pbits = 0;
      while ((1u << pbits) < P) ++pbits;
      TileSlicing sl_post, sl_pre;
      const bool post = d->rx_pending && d->rx_after && P && d->rx_packed.size() == pbits;
      if (post) {
        sl_post.nbits = pbits;
        sl_post.in_place_only = true;
        for (uint32_t j = 0; j < pbits; ++j) sl_post.pos[j] = d->rx_packed[j];
        sl_post.before = [d](uint32_t k, bool) -> int {
          HIPCHK(hipStreamWaitEvent(d->shard->stream, d->ev_rx[k], 0));
          return QIP_OK;
        };
        sl_post.fallback = [d]() -> int { return dist_wait_rx(d); };
        s->slice_first = &sl_post;
      } else {
        QCHK(dist_wait_rx(d));
      }
      const size_t ex_at = jx + (jx < steps.size() && steps[jx].kind == qipd::Step::PACK ? 1 : 0);
      const bool ex_next = ex_at < steps.size() && steps[ex_at].kind == qipd::Step::EXCHANGE;
      const bool pack_in_plan = ex_next && ex_at != jx;
      // (with a PERSISTENT relabelling, tile_relabel = 3, the batch is scheduled from and to layouts edge_tile does not
      // model — the prediction "ends in the caller's order" could be wrong on every rank alike: such states vote no)
      bool pre = P && ex_next && s->layout.empty() && s->tile_relabel < 3 && (!pack_in_plan || pack_next);
      const uint64_t chunk_bytes = (s->namps >> g) * s->amp_bytes;
      std::vector<uint32_t> packed;
      if (P && ex_next) {
        // the sweeps either side, as apply_ops will schedule them (the batch after the remap is marshalled early for that); the vote
        // is cast for EVERY exchange of the plan, feasible here or not: the agreement is a collective
        std::vector<double> votes;
        // everything that can fail on ONE rank only is settled BEFORE the vote and goes into it: a rank that found out afterwards
        // that it cannot follow would leave the others in a sliced exchange it never joins
        if (pre && (dist_overlap_setup(d, P) != QIP_OK || ensure_alt(s) != QIP_OK)) pre = false;
        if (pre && pack_in_plan && !d->third && hipMalloc(&d->third, s->namps * s->amp_bytes) != hipSuccess) {
          (void)hipGetLastError();  // no room for the receive buffer: this rank votes no, every rank runs the remap the serial way
          d->third = nullptr;
          pre = false;
        }
        if (pre) {
          const int mode = state_tile_mode(s);
          const EdgeTile e_pre = edge_tile(s->dtype, L, m.flat.data(), m.flat.size(), mode, true);
          EdgeTile e_post;
          qipd::Marshalled m2;
          std::vector<const qip_op*> p2;
          for (size_t t = ex_at + 1; t < steps.size() && steps[t].kind == qipd::Step::LOCAL; ++t) p2.push_back(qipd::marshal_one(s->dtype, *steps[t].op, &m2));
          for (const qip_op* q2 : p2) m2.flat.push_back(*q2);
          if (!m2.flat.empty()) e_post = edge_tile(s->dtype, L, m2.flat.data(), m2.flat.size(), mode, false);
          slice_votes(L, g, pack_in_plan ? &steps[jx].sel : nullptr, e_pre, m2.flat.empty() ? nullptr : &e_post, true, &votes);
        } else {
          slice_votes(L, g, nullptr, EdgeTile(), nullptr, false, &votes);
        }
        QCHK(dist_agree(d, &votes));
        d->agreed_for = ex_at;
        pre = pre && choose_slices(L, g, pbits, votes, d->pl.world, &packed);
      } else {
        pre = false;
      }
      if (pre) {
        sl_pre.nbits = pbits;
        sl_pre.need_fold = pack_in_plan;
        for (uint32_t j = 0; j < pbits; ++j) sl_pre.pos[j] = packed_to_source(packed[j], pack_in_plan ? &steps[jx].sel : nullptr, L);
        const uint64_t amp_bytes = s->amp_bytes;
        sl_pre.after = [d, s, chunk_bytes, amp_bytes, pack_in_plan, packed, L, g](uint32_t k, bool folding) -> int {
          if (folding != pack_in_plan) return fail(QIP_ERR_DEVICE, "internal: overlapped exchange and packed store disagree");
          HIPCHK(hipEventRecord(d->ev_pre[k], s->stream));
          HIPCHK(hipStreamWaitEvent(d->comm_stream, d->ev_pre[k], 0));
          // (the part has been enqueued, the buffers have not been exchanged yet: a packed store went to `alt`)
          const void* send = folding ? s->alt : s->cur;
          void* recv = folding ? d->third : s->alt;
          std::vector<uint64_t> offs;
          uint64_t run_bytes = 0;
          slice_runs(L, g, packed, k, amp_bytes, &offs, &run_bytes);
          for (uint64_t off : offs) {
            const int rc = d->slice_fn(d->transport.ctx, send, recv, chunk_bytes, off, run_bytes, (void*)d->comm_stream);
            if (rc != 0) return g_last_error.empty() ? fail(QIP_ERR_DEVICE, "transport all_to_all_slice failed with status %d", rc) : QIP_ERR_DEVICE;
          }
          HIPCHK(hipEventRecord(d->ev_rx[k], d->comm_stream));
          return QIP_OK;
        };
        s->slice_last = &sl_pre;
      }
      const int rc_batch = qip_hip_state_apply_ops(s, m.flat.data(), m.flat.size());
      const bool folded = pack_next && s->fold_done;
      s->fold_request = nullptr;
      s->fold_done = false;
      s->slice_first = s->slice_last = s->slice_now = nullptr;
      QCHK(rc_batch);
      if (post) {
        QCHK(dist_wait_rx(d));  // (a no-op in stream order when the first sweep has awaited every slice; otherwise the safety net)
        if (sl_post.parts_done == P) d->stats.remaps_overlapped_after += 1;
      }
      if (folded) {
        d->stats.packs_folded += 1;
        ++jx;  // the PACK step is done
      }
      // (the ranks AGREED to cut this exchange: a sweep that ran uncut here would leave the peers in slices this rank never sends —
      // fail loudly instead of falling back to the serial exchange alone)
      if (pre && sl_pre.parts_done == 0) return fail(QIP_ERR_DEVICE, "internal: the sweep before an exchange the ranks agreed to cut ran uncut on rank %d", d->pl.rank);
      if (pre) {
        if (sl_pre.parts_done != P || sl_pre.folded != pack_in_plan) return fail(QIP_ERR_DEVICE, "internal: overlapped exchange left incomplete");
        // every slice is on its way: the receive buffer becomes the shard (builder.rs:514 analogue), the EXCHANGE step is done
        if (pack_in_plan) {
          std::swap(s->cur, d->third);  // cur was the packed copy (folded_swap): it is what the communication stream still reads
        } else {
          std::swap(s->cur, s->alt);
          std::swap(s->owns_cur, s->owns_alt);
        }
        d->rx_pending = true;
        d->rx_slices = P;
        d->rx_packed = packed;  // (after the exchange the packed positions ARE the shard's positions)
        d->rx_after = true;     // (whether the next batch's first sweep can take them is decided when it is launched: fallback = full wait)
        d->stats.remaps += 1;
        d->stats.remaps_overlapped += 1;
        d->stats.slices_overlapped += P;
        d->stats.bytes_sent += chunk_bytes * (uint64_t)(d->pl.world - 1);
        jx = ex_at + 1;
      }
      // payload buffers of this batch die with `m`: the uploads above were staged by the runtime before returning
      i = jx;
      continue;
    }
    QCHK(dist_wait_rx(d));  // (a gather or an exchange of its own reads the whole shard)
    {  // an exchange this rank reaches without a local batch before it: the other ranks may have one and vote — vote "no" with them
      const size_t ex_here = i + (steps[i].kind == qipd::Step::PACK ? 1 : 0);
      const bool votes_held = d->overlap >= 2 && d->slice_fn && d->pl.world > 1 && s->tile >= 1 && s->tile_passes && !s->force_generic && !g_force_generic;
      if (votes_held && ex_here < steps.size() && steps[ex_here].kind == qipd::Step::EXCHANGE && d->agreed_for != ex_here) {
        std::vector<double> votes;
        slice_votes(L, g, nullptr, EdgeTile(), nullptr, false, &votes);
        QCHK(dist_agree(d, &votes));
        d->agreed_for = ex_here;
      }
    }
    if (!s->layout.empty()) QCHK(state_settle(s));  // (a shard with a persistent relabelling: the caller's order first)
    if (steps[i].kind == qipd::Step::PACK) {
      QCHK(ensure_alt(s));
      // a selected position inside a 1-KiB row would make k_pack_bits read 16-byte pieces: such a gather goes through the
      // LDS-tiled bit-permutation sweep (k_permute_bits), which streams whole rows on both sides whatever the positions
      bool low_sel = false;
      for (uint32_t t = 0; t < g; ++t) low_sel = low_sel || steps[i].sel[t] < 6;
      if (low_sel) {
        std::vector<uint32_t> pi(L);
        uint32_t next = 0;
        for (uint32_t pp = 0; pp < L; ++pp)
          if (std::find(steps[i].sel.begin(), steps[i].sel.end(), pp) == steps[i].sel.end()) pi[next++] = pp;
        for (uint32_t t = 0; t < g; ++t) pi[L - g + t] = steps[i].sel[t];
        QCHK(dist_timed(d, &d->ev_pack, [&]() -> int { return launch_permute(s, pi.data()); }));  // (swaps cur / alt itself)
        d->stats.pack_sweeps += 1;
        d->stats.packs_via_permute += 1;
        ++i;
        continue;
      }
      qipk::PackDesc pd;
      memset(&pd, 0, sizeof pd);
      pd.g = g;
      pd.Lg = L - g;
      for (uint32_t t = 0; t < g; ++t) pd.sel[t] = steps[i].sel[t];
      Ins ins = make_ins(steps[i].sel, 0);
      const dim3 grid = grid2d(s->namps, (uint64_t)kBlock * 4);
      QCHK(dist_timed(d, &d->ev_pack, [&]() -> int {
        if (s->dtype == QIP_C64)
          hipLaunchKernelGGL((qipk::k_pack_bits<double>), grid, dim3(kBlock), 0, s->stream, (const amp_t<double>*)s->cur,
                             (amp_t<double>*)s->alt, s->namps, ins, pd);
        else
          hipLaunchKernelGGL((qipk::k_pack_bits<float>), grid, dim3(kBlock), 0, s->stream, (const amp_t<float>*)s->cur,
                             (amp_t<float>*)s->alt, s->namps, ins, pd);
        HIPCHK(hipGetLastError());
        return QIP_OK;
      }));
      std::swap(s->cur, s->alt);
      std::swap(s->owns_cur, s->owns_alt);
      d->stats.pack_sweeps += 1;
      ++i;
      continue;
    }
    // EXCHANGE: cur -> alt through the transport, then alt becomes current (builder.rs:514 analogue)
    QCHK(ensure_alt(s));
    const uint64_t chunk_bytes = (s->namps >> g) * s->amp_bytes;
    QCHK(dist_timed(d, &d->ev_exchange, [&]() -> int {
      int rc = d->transport.all_to_all(d->transport.ctx, s->cur, s->alt, chunk_bytes, (void*)s->stream);
      if (rc != 0 && g_last_error.empty()) return fail(QIP_ERR_DEVICE, "transport all_to_all failed with status %d", rc);
      return rc == 0 ? QIP_OK : QIP_ERR_DEVICE;
    }));
    std::swap(s->cur, s->alt);
    std::swap(s->owns_cur, s->owns_alt);
    d->stats.remaps += 1;
    d->stats.bytes_sent += chunk_bytes * (uint64_t)(d->pl.world - 1);
    ++i;
  }
  return QIP_OK;
}

#define DIST_ENTER(d)                                                                                                       \
  if (!(d)) return fail(QIP_ERR_INVALID, "null dist handle");                                                                \
  if ((d)->poisoned) return fail(QIP_ERR_DEVICE, "sharded state unusable after an earlier failure: %s", (d)->poison_msg.c_str()); \
  HIPCHK(hipSetDevice((d)->shard->device))

// run a planned batch; on failure the handle is poisoned (see qip_hip_dist::poisoned)
static int dist_run_steps(qip_hip_dist* d, std::vector<qipd::Step>& steps);
static int dist_all_reduce(qip_hip_dist* d, double* v, uint64_t count);
// Every rank plans for itself, from its own copy of the circuit and of the process-global options the planner reads
// (dist_plan_cost, tile_row_split through row_p5(), dist_fold_pack): these MUST be identical on all ranks (include/qip_hip.h).
// Ranks whose plans differ would post different exchanges — a deadlock or a silently wrong all-to-all — so a batch
// that contains an exchange first compares a fingerprint of its communication steps (which exchanges, which positions each
// gather selects) across the ranks: one 16-byte all-reduce, and every rank sees the same verdict, so all of them refuse together.
static int dist_plans_agree(qip_hip_dist* d, const std::vector<qipd::Step>& steps) {
  if (d->pl.world <= 1) return QIP_OK;
  uint64_t h = 1469598103934665603ull, comm_steps = 0;
  auto mix = [&](uint64_t v) {
    h ^= v;
    h *= 1099511628211ull;
  };
  for (const qipd::Step& st : steps) {
    if (st.kind == qipd::Step::LOCAL) continue;  // (local ops legitimately differ between ranks: controls on rank bits)
    comm_steps += 1;
    mix(st.kind == qipd::Step::PACK ? 0x50u : 0x45u);
    for (uint32_t p : st.sel) mix(0x100u + p);
  }
  if (comm_steps == 0) return QIP_OK;
  // (whether the slice votes of an overlapped exchange are held — collectives too — follows from per-handle options:
  // ranks that differ in them must be refused here, not hang in a different number of all-reduces)
  mix(0x4f00u + (uint64_t)d->overlap);
  mix(0x5400u + (uint64_t)(d->shard->tile >= 1) + 2 * (uint64_t)(d->shard->tile_passes != 0) + 4 * (uint64_t)(d->shard->force_generic != 0) +
      8 * (uint64_t)(d->shard->tile_relabel >= 3));
  const double v = (double)((h ^ (h >> 20) ^ (h >> 40)) & 0xfffffull);  // 20 bits: v^2 * world stays exact in a double
  double r[2] = {v, v * v};
  QCHK(dist_all_reduce(d, r, 2));
  const double w = (double)d->pl.world;
  if (r[0] != w * v || r[1] != w * v * v)
    return fail(QIP_ERR_INVALID, "the ranks planned different exchanges for this batch (do they see the same circuit and the same "
                                 "options dist_plan_cost / dist_fold_pack / dist_overlap / tile?)");
  return QIP_OK;
}
static int dist_run_or_poison(qip_hip_dist* d, std::vector<qipd::Step>& steps) {
  int rc = dist_plans_agree(d, steps);
  if (rc == QIP_OK) rc = dist_run_steps(d, steps);
  if (rc != QIP_OK) {
    d->poisoned = true;
    d->poison_msg = g_last_error;
  }
  return rc;
}

extern "C" int qip_hip_dist_unique_id(void* id_out) try {
  if (!id_out) return fail(QIP_ERR_INVALID, "null output");
  QCHK(qipd::rccl_load());
  qipd::Rccl::UniqueId id;
  NCHK(qipd::g_rccl.GetUniqueId(&id));
  memcpy(id_out, &id, sizeof id);
  return QIP_OK;
} QIP_CATCH_ALL

extern "C" int qip_hip_dist_destroy(qip_hip_dist* d) try {
  if (!d) return QIP_OK;
  if (d->shard) {
    (void)hipSetDevice(d->shard->device);
    (void)dist_drain_events(d);
  }
  if (d->rccl) {
    if (d->rccl->comm) (void)qipd::g_rccl.CommDestroy(d->rccl->comm);
    if (d->rccl->d_red) (void)hipFree(d->rccl->d_red);
    delete d->rccl;
  }
  if (d->comm_stream) (void)hipStreamSynchronize(d->comm_stream);
  for (hipEvent_t e : d->ev_pre) (void)hipEventDestroy(e);
  for (hipEvent_t e : d->ev_rx) (void)hipEventDestroy(e);
  if (d->comm_stream) (void)hipStreamDestroy(d->comm_stream);
  if (d->third) (void)hipFree(d->third);
  if (d->shard) qip_hip_state_destroy(d->shard);
  delete d;
  return QIP_OK;
} QIP_CATCH_ALL

extern "C" int qip_hip_dist_set_slice_transport(qip_hip_dist* d, qip_hip_all_to_all_slice_fn fn) try {
  if (!d) return fail(QIP_ERR_INVALID, "null dist handle");
  if (d->rccl) return fail(QIP_ERR_UNSUPPORTED, "the built-in RCCL transport brings its own slice entry point");
  d->slice_fn = fn;
  return QIP_OK;
} QIP_CATCH_ALL

extern "C" int qip_hip_dist_create(uint32_t n, int dtype, int device, int rank, int world, const void* unique_id,
                                   const qip_hip_transport* transport, qip_hip_dist** out) try {
  if (!out) return fail(QIP_ERR_INVALID, "null output handle");
  *out = nullptr;
  if (dtype != QIP_C64 && dtype != QIP_C32) return fail(QIP_ERR_INVALID, "bad dtype %d", dtype);
  std::unique_ptr<qip_hip_dist> d(new qip_hip_dist());
  QCHK(d->pl.init(n, dtype, rank, world));
  if (d->pl.g > 8) return fail(QIP_ERR_UNSUPPORTED, "more than 256 ranks");
  QCHK(qip_hip_state_create(d->pl.L, dtype, device, &d->shard));
  auto bail = [&](int rc) {
    std::string keep = g_last_error;
    qip_hip_dist_destroy(d.release());
    g_last_error = keep;
    return rc;
  };
  if (transport) {
    if (!transport->all_to_all || !transport->all_reduce_sum) return bail(fail(QIP_ERR_INVALID, "transport with a null callback"));
    d->transport = *transport;
  } else {
    int rc = qipd::rccl_load();
    if (rc != QIP_OK) return bail(rc);
    if (!unique_id) return bail(fail(QIP_ERR_INVALID, "the RCCL transport needs the unique id from qip_hip_dist_unique_id"));
    d->rccl = new qipd::RcclTransport();
    d->rccl->rank = rank;
    d->rccl->world = world;
    d->rccl->stream = d->shard->stream;
    qipd::Rccl::UniqueId id;
    memcpy(&id, unique_id, sizeof id);
    const int r = qipd::g_rccl.CommInitRank(&d->rccl->comm, world, id, rank);
    if (r != 0) return bail(fail(QIP_ERR_DEVICE, "ncclCommInitRank failed: %s", qipd::g_rccl.GetErrorString(r)));
    d->transport.ctx = d->rccl;
    d->transport.all_to_all = qipd::rccl_all_to_all;
    d->transport.all_reduce_sum = qipd::rccl_all_reduce;
    d->slice_fn = qipd::rccl_all_to_all_slice;
  }
  // the ranks of one node share its CPUs: the run-time compiler's automatic helper-process count is divided by the world size
  if (world > 1 && (int64_t)world > g_jit_world) g_jit_world = world;
  if (world > 1) {
    // every remap goes through the second 2^L buffer: get it now, so that a state too large for two buffers fails here and
    // not in the middle of a circuit, after local ops have already changed the shard
    const int rc = ensure_alt(d->shard);
    if (rc != QIP_OK) return bail(rc);
  }
  *out = d.release();
  return QIP_OK;
} QIP_CATCH_ALL

extern "C" int64_t qip_hip_dist_debug_pieces(int rank, int world, uint64_t chunk_bytes, uint64_t piece_bytes, uint64_t cap, int32_t* peer,
                                             uint64_t* offset, uint64_t* length) try {
  if (world < 1 || rank < 0 || rank >= world) return fail(QIP_ERR_INVALID, "rank %d of %d", rank, world) < 0 ? -1 : -1;
  const std::vector<qipd::Piece> ps = qipd::plan_pieces(rank, world, chunk_bytes, piece_bytes);
  for (uint64_t i = 0; i < ps.size() && i < cap; ++i) {
    if (peer) peer[i] = ps[i].peer;
    if (offset) offset[i] = ps[i].offset;
    if (length) length[i] = ps[i].length;
  }
  return (int64_t)ps.size();
} catch (...) { return -1; }

extern "C" int qip_hip_dist_local_state(qip_hip_dist* d, qip_hip_state** shard) try {
  if (!d || !shard) return fail(QIP_ERR_INVALID, "null argument");
  *shard = d->shard;
  return QIP_OK;
} QIP_CATCH_ALL

extern "C" int qip_hip_dist_layout(qip_hip_dist* d, uint32_t* phys) try {
  if (!d || !phys) return fail(QIP_ERR_INVALID, "null argument");
  for (uint32_t p = 0; p < d->pl.n; ++p) phys[p] = d->pl.phys[p];
  return QIP_OK;
} QIP_CATCH_ALL

extern "C" int qip_hip_dist_rank_flip(qip_hip_dist* d, uint32_t* mask) try {
  if (!d || !mask) return fail(QIP_ERR_INVALID, "null argument");
  *mask = d->pl.flip;
  return QIP_OK;
} QIP_CATCH_ALL

extern "C" int qip_hip_dist_set_option(qip_hip_dist* d, const char* key, int64_t value) try {
  if (!d) return fail(QIP_ERR_INVALID, "null dist handle");
  if (key && !strcmp(key, "piece_bytes")) {  // largest single ncclSend / ncclRecv of the built-in transport
    if (value < 16 || (value & 15)) return fail(QIP_ERR_INVALID, "piece_bytes must be a positive multiple of 16");
    if (!d->rccl) return fail(QIP_ERR_UNSUPPORTED, "piece_bytes belongs to the built-in RCCL transport; this handle uses caller-supplied callbacks");
    d->rccl->piece_bytes = (uint64_t)value;
    return QIP_OK;
  }
  if (key && !strcmp(key, "dist_overlap")) {
    if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8) return fail(QIP_ERR_INVALID, "dist_overlap is 0, 1 (off), 2, 4 or 8 slices");
    d->overlap = value;
    return QIP_OK;
  }
  return qip_hip_state_set_option(d->shard, key, value);
} QIP_CATCH_ALL

extern "C" int qip_hip_dist_sync(qip_hip_dist* d) try {
  DIST_ENTER(d);
  if (d->comm_stream) QCHK(qipd::wait_stream(d->comm_stream, d->pl.rank, d->pl.world, "the exchange on the communication stream"));
  const int rc = qipd::wait_stream(d->shard->stream, d->pl.rank, d->pl.world, "the queued batch (sweeps + exchange)");
  if (rc != QIP_OK) {  // what is queued behind a collective that never completes can never run: the handle is done
    d->poisoned = true;
    d->poison_msg = g_last_error;
  }
  return rc;
} QIP_CATCH_ALL

extern "C" int qip_hip_dist_take_stats(qip_hip_dist* d, qip_hip_dist_stats* out) try {
  DIST_ENTER(d);
  if (!out) return fail(QIP_ERR_INVALID, "null output");
  QCHK(dist_drain_events(d));
  *out = d->stats;
  out->rccl_ranks = 0;
  out->rccl_rank = -1;
  out->piece_bytes = 0;
  if (d->rccl) {
    // read back from the communicator itself, not from what the caller passed to qip_hip_dist_create
    int cnt = 0, ur = -1;
    NCHK(qipd::g_rccl.CommCount(d->rccl->comm, &cnt));
    NCHK(qipd::g_rccl.CommUserRank(d->rccl->comm, &ur));
    out->rccl_ranks = cnt;
    out->rccl_rank = ur;
    out->pieces_sent = d->rccl->pieces_sent;
    out->piece_bytes = d->rccl->piece_bytes;
    d->rccl->pieces_sent = 0;
  }
  d->stats = qip_hip_dist_stats{};
  return QIP_OK;
} QIP_CATCH_ALL

extern "C" int qip_hip_dist_init_basis(qip_hip_dist* d, uint64_t logical_index) try {
  DIST_ENTER(d);
  const qipd::DistPlanner& pl = d->pl;
  if (pl.n < 64 && (logical_index >> pl.n) != 0) return fail(QIP_ERR_INVALID, "basis index out of range");
  uint64_t P = 0;
  for (uint32_t p = 0; p < pl.n; ++p) P |= ((logical_index >> p) & 1ull) << pl.phys[p];
  const uint64_t owner = (P >> pl.L) ^ pl.flip, local = P & ((1ull << pl.L) - 1ull);  // (pending rank renamings)
  if ((int)owner == pl.rank) return qip_hip_state_init_basis(d->shard, local);
  qip_hip_state* s = d->shard;
  s->layout.clear();  // (an all-zero shard is the same in every order)
  HIPCHK(hipMemsetAsync(s->cur, 0, s->namps * s->amp_bytes, s->stream));
  HIPCHK(hipStreamSynchronize(s->stream));
  return QIP_OK;
} QIP_CATCH_ALL

extern "C" int qip_hip_dist_apply_ops(qip_hip_dist* d, const qip_op* ops, uint64_t count) try {
  DIST_ENTER(d);
  if (count && !ops) return fail(QIP_ERR_INVALID, "null op array");
  std::vector<qipd::Step> steps;
  qipd::DistPlanner trial = d->pl;  // the layout only advances if planning succeeds
  QCHK(trial.plan(ops, count, &steps));
  d->pl = trial;
  return dist_run_or_poison(d, steps);
} QIP_CATCH_ALL

extern "C" int qip_hip_dist_apply_op(qip_hip_dist* d, const qip_op* op) try {
  DIST_ENTER(d);
  qipd::OpInfo info;
  QCHK(qipd::analyse(d->pl.n, d->pl.dtype, op, &info));
  std::vector<qipd::Step> steps;
  qipd::DistPlanner trial = d->pl;
  QCHK(trial.step(info, nullptr, &steps));
  d->pl = trial;
  return dist_run_or_poison(d, steps);
} QIP_CATCH_ALL

static int dist_all_reduce(qip_hip_dist* d, double* v, uint64_t count) {
  const int rc = d->transport.all_reduce_sum(d->transport.ctx, v, count);
  if (rc != 0) return g_last_error.empty() ? fail(QIP_ERR_DEVICE, "transport all_reduce_sum failed with status %d", rc) : QIP_ERR_DEVICE;
  return QIP_OK;
}

extern "C" int qip_hip_dist_norm_sqr(qip_hip_dist* d, double* out) try {
  DIST_ENTER(d);
  if (!out) return fail(QIP_ERR_INVALID, "null output");
  double v = 0;
  QCHK(qip_hip_state_norm_sqr(d->shard, &v));
  QCHK(dist_all_reduce(d, &v, 1));
  *out = v;
  return QIP_OK;
} QIP_CATCH_ALL

// measure_probs (measurement_ops.rs:115-127): bit i of the outcome <-> indices[i].  Measured qubits on rank bits
// contribute this rank's bit; the local ones go through the shard's measure_probs; one all-reduce adds the ranks up.
static int dist_measure_probs(qip_hip_dist* d, const uint64_t* indices, uint32_t k, std::vector<double>* out) {
  const qipd::DistPlanner& pl = d->pl;
  if (k == 0 || k > pl.n || !indices) return fail(QIP_ERR_INVALID, "bad measurement index list");
  if (k > 30) return fail(QIP_ERR_UNSUPPORTED, "measure_probs over %u qubits", k);
  uint64_t seen = 0, fixed = 0;
  std::vector<uint32_t> loc_i;
  std::vector<uint64_t> loc_q;
  for (uint32_t i = 0; i < k; ++i) {
    if (indices[i] >= pl.n) return fail(QIP_ERR_INVALID, "measured qubit index out of range");
    if (seen & (1ull << indices[i])) return fail(QIP_ERR_INVALID, "repeated measured qubit index");
    seen |= 1ull << indices[i];
    const uint32_t pp = pl.phys[pl.n - 1 - (uint32_t)indices[i]];
    if (pp >= pl.L) {
      fixed |= (uint64_t)pl.rank_bit(pp) << i;
    } else {
      loc_i.push_back(i);
      loc_q.push_back(pl.local_qubit(pp));
    }
  }
  out->assign(1ull << k, 0.0);
  if (!loc_i.empty()) {
    std::vector<double> part(1ull << loc_i.size());
    QCHK(qip_hip_state_measure_probs(d->shard, loc_q.data(), (uint32_t)loc_q.size(), part.data()));
    for (uint64_t sidx = 0; sidx < part.size(); ++sidx) {
      uint64_t m = fixed;
      for (size_t b = 0; b < loc_i.size(); ++b) m |= ((sidx >> b) & 1ull) << loc_i[b];
      (*out)[m] += part[sidx];
    }
  } else {
    double v = 0;
    QCHK(qip_hip_state_norm_sqr(d->shard, &v));
    (*out)[fixed] = v;
  }
  return dist_all_reduce(d, out->data(), out->size());
}

extern "C" int qip_hip_dist_measure_probs(qip_hip_dist* d, const uint64_t* indices, uint32_t k, double* out) try {
  DIST_ENTER(d);
  if (!out) return fail(QIP_ERR_INVALID, "null output");
  std::vector<double> probs;
  QCHK(dist_measure_probs(d, indices, k, &probs));
  memcpy(out, probs.data(), probs.size() * sizeof(double));
  return QIP_OK;
} QIP_CATCH_ALL

// soft_measure (measurement_ops.rs:153-176) over the WHOLE vector in LOGICAL index order: r -= |amp_i|^2 until r <= 0.
// Instead of walking 2^n amplitudes across ranks, descend the index bit by bit from the top: the mass of the half-block
// whose next bit is 0 is one masked norm (local reduction over a sub-space that halves at every local bit, so ~2 sweeps in
// all) + one all-reduce; the crossing lies in it iff r - mass <= 0.  Same sample -> outcome map as the reference up to the
// rounding of block sums (the caveat of the single-GPU soft_measure; tests/dist_worker_gpu.py sweeps samples against the
// oracle and counts the disagreements).  Rank 0's sample decides, so every rank arrives at the same outcome.
static int dist_soft_measure(qip_hip_dist* d, const uint64_t* indices, uint32_t k, double rand_u01, uint64_t* measured) {
  const qipd::DistPlanner& pl = d->pl;
  double r = pl.rank == 0 ? rand_u01 : 0.0;
  QCHK(dist_all_reduce(d, &r, 1));
  uint64_t logical_index = 0;
  std::vector<uint64_t> cq;      // local qubits constrained so far
  uint64_t cval = 0;             // their required values (bit j <-> cq[j])
  bool rank_ok = true;           // this rank's bits agree with the prefix chosen so far
  for (uint32_t qb = 0; qb < pl.n; ++qb) {
    const uint32_t p_log = pl.n - 1 - qb, pp = pl.phys[p_log];
    // mass of (prefix, bit = 0)
    double mass = 0.0;
    if (pp >= pl.L) {
      if (rank_ok && pl.rank_bit(pp) == 0) {
        if (cq.empty()) QCHK(qip_hip_state_norm_sqr(d->shard, &mass));
        else QCHK(qip_hip_state_measure_prob(d->shard, cval, cq.data(), (uint32_t)cq.size(), &mass));
      }
    } else if (rank_ok) {
      cq.push_back(pl.local_qubit(pp));  // bit value 0 for this probe: cval unchanged
      QCHK(qip_hip_state_measure_prob(d->shard, cval, cq.data(), (uint32_t)cq.size(), &mass));
    } else {
      cq.push_back(pl.local_qubit(pp));
    }
    QCHK(dist_all_reduce(d, &mass, 1));
    uint64_t bit = 0;
    if (!(r - mass <= 0.0)) {
      r -= mass;
      bit = 1;
    }
    logical_index |= bit << p_log;
    if (pp >= pl.L) rank_ok = rank_ok && pl.rank_bit(pp) == bit;
    else cval |= bit << (cq.size() - 1);
  }
  // the walk never crossed before the LAST amplitude: the reference then tests that one too and, if the sample is
  // still positive (it exceeds the norm by rounding), leaves measured_indx at 0 (measurement_ops.rs:166-173)
  if (logical_index == (pl.n < 64 ? (1ull << pl.n) - 1ull : ~0ull)) {
    double last = 0.0;
    if (rank_ok) {
      if (cq.empty()) QCHK(qip_hip_state_norm_sqr(d->shard, &last));
      else QCHK(qip_hip_state_measure_prob(d->shard, cval, cq.data(), (uint32_t)cq.size(), &last));
    }
    QCHK(dist_all_reduce(d, &last, 1));
    if (!(r - last <= 0.0)) logical_index = 0;
  }
  uint64_t m = 0;
  for (uint32_t i = 0; i < k; ++i) m |= ((logical_index >> (pl.n - 1 - (uint32_t)indices[i])) & 1ull) << i;
  *measured = m;
  return QIP_OK;
}

extern "C" int qip_hip_dist_soft_measure(qip_hip_dist* d, const uint64_t* indices, uint32_t k, double rand_u01, uint64_t* measured) try {
  DIST_ENTER(d);
  if (!measured || (k && !indices)) return fail(QIP_ERR_INVALID, "null argument");
  for (uint32_t i = 0; i < k; ++i)
    if (indices[i] >= d->pl.n) return fail(QIP_ERR_INVALID, "measured index %llu out of range", (unsigned long long)indices[i]);
  return dist_soft_measure(d, indices, k, rand_u01, measured);
} QIP_CATCH_ALL

extern "C" int qip_hip_dist_measure(qip_hip_dist* d, const uint64_t* indices, uint32_t k, int64_t forced, double rand_u01,
                                    uint64_t* measured, double* prob) try {
  DIST_ENTER(d);
  if (!measured || !prob) return fail(QIP_ERR_INVALID, "null output");
  const qipd::DistPlanner& pl = d->pl;
  std::vector<double> probs;
  QCHK(dist_measure_probs(d, indices, k, &probs));
  uint64_t m = 0;
  if (forced >= 0) {
    m = (uint64_t)forced;
    if (k < 64 && (m >> k) != 0) return fail(QIP_ERR_INVALID, "forced outcome has more than k bits");
  } else {
    QCHK(dist_soft_measure(d, indices, k, rand_u01, &m));
  }
  const double p = probs[m];
  *measured = m;
  *prob = p;
  if (p == 0.0) return QIP_OK;  // measure_state is a no-op (:230)
  bool agrees = true;
  std::vector<uint64_t> loc_q;
  uint64_t lm = 0;
  for (uint32_t i = 0; i < k; ++i) {
    const uint32_t pp = pl.phys[pl.n - 1 - (uint32_t)indices[i]];
    if (pp >= pl.L) {
      agrees = agrees && pl.rank_bit(pp) == ((m >> i) & 1ull);
    } else {
      lm |= ((m >> i) & 1ull) << loc_q.size();
      loc_q.push_back(pl.local_qubit(pp));
    }
  }
  qip_hip_state* s = d->shard;
  if (!agrees) {  // this rank's bits contradict the outcome: the whole shard goes to zero
    s->layout.clear();
    HIPCHK(hipMemsetAsync(s->cur, 0, s->namps * s->amp_bytes, s->stream));
    return QIP_OK;
  }
  return qip_hip_state_measure_state(s, loc_q.data(), (uint32_t)loc_q.size(), lm, p);
} QIP_CATCH_ALL

// Host-only (r5): which remaps the overlapped exchange (option "dist_overlap") serves when the local batches run as tile sweeps in
// scheduler mode `tile_mode` — decided by the very functions the executor uses (edge_tile, slice_votes, choose_slices), with the votes of
// ALL ranks (each rank's plan is made here, their votes summed as the executor's all-reduce sums them); "after" is `rank`'s own view.
extern "C" const char* qip_hip_dist_debug_overlap(uint32_t n, int dtype, int rank, int world, const qip_op* ops, uint64_t count, int tile_mode,
                                                  int slices) {
  static thread_local std::string json;
  try {
    if (count && !ops) return fail(QIP_ERR_INVALID, "null op array"), nullptr;
    if (dtype != QIP_C64 && dtype != QIP_C32) return fail(QIP_ERR_INVALID, "bad dtype %d", dtype), nullptr;
    if (slices != 2 && slices != 4 && slices != 8) return fail(QIP_ERR_INVALID, "slices is 2, 4 or 8"), nullptr;
    if (rank < 0 || rank >= world) return fail(QIP_ERR_INVALID, "rank %d of %d", rank, world), nullptr;
    uint32_t pbits = 0;
    while ((1 << pbits) < slices) ++pbits;
    struct Remap {
      bool pack = false, after = false;
      size_t sweeps = 0;
      std::vector<double> votes;
    };
    std::vector<std::vector<Remap>> per_rank((size_t)world);
    uint32_t g = 0, L = 0;
    for (int r = 0; r < world; ++r) {
      qipd::DistPlanner pl;
      if (pl.init(n, dtype, r, world) != QIP_OK) return nullptr;
      std::vector<qipd::Step> steps;
      if (pl.plan(ops, count, &steps) != QIP_OK) return nullptr;
      g = pl.g;
      L = pl.L;
      auto batch = [&](size_t from, qipd::Marshalled* m) {
        std::vector<const qip_op*> ptrs;
        size_t t = from;
        for (; t < steps.size() && steps[t].kind == qipd::Step::LOCAL; ++t) ptrs.push_back(qipd::marshal_one(dtype, *steps[t].op, m));
        for (const qip_op* p : ptrs) m->flat.push_back(*p);
        return t;
      };
      size_t i = 0;
      while (i < steps.size()) {
        qipd::Marshalled m;
        const size_t jx = steps[i].kind == qipd::Step::LOCAL ? batch(i, &m) : i;
        const size_t ex_at = jx + (jx < steps.size() && steps[jx].kind == qipd::Step::PACK ? 1 : 0);
        if (!(ex_at < steps.size() && steps[ex_at].kind == qipd::Step::EXCHANGE)) {
          i = jx > i ? jx : i + 1;
          continue;
        }
        Remap rm;
        rm.pack = ex_at != jx;
        const bool usable = world > 1 && !m.flat.empty() && (!rm.pack || g_dist_fold_pack);
        EdgeTile e_pre, e_post;
        qipd::Marshalled m2;
        batch(ex_at + 1, &m2);
        if (usable) {
          e_pre = edge_tile(dtype, L, m.flat.data(), m.flat.size(), tile_mode, true);
          rm.sweeps = e_pre.nsteps;
          if (!m2.flat.empty()) e_post = edge_tile(dtype, L, m2.flat.data(), m2.flat.size(), tile_mode, false);
        }
        slice_votes(L, g, rm.pack ? &steps[jx].sel : nullptr, e_pre, m2.flat.empty() ? nullptr : &e_post, usable, &rm.votes);
        rm.after = !m2.flat.empty() && e_post.segment && e_post.starts_plain;  // (refined below against the chosen positions)
        if (rm.after) {
          // remember the first sweep's tile through the votes' second half: position q acceptable <=> votes[1 + W + q] on THIS rank
        }
        per_rank[(size_t)r].push_back(std::move(rm));
        i = ex_at + 1;
      }
    }
    json = "{\"n\":" + std::to_string(n) + ",\"slices\":" + std::to_string(slices) + ",\"remaps\":[";
    const size_t nrem = per_rank[0].size();
    for (int r = 1; r < world; ++r)
      if (per_rank[(size_t)r].size() != nrem) return fail(QIP_ERR_INVALID, "internal: the ranks plan different numbers of exchanges"), nullptr;
    const uint32_t W = L - g;
    for (size_t k = 0; k < nrem; ++k) {
      std::vector<double> sum(1 + 2 * (size_t)W, 0.0);
      for (int r = 0; r < world; ++r)
        for (size_t t = 0; t < sum.size(); ++t) sum[t] += per_rank[(size_t)r][k].votes[t];
      std::vector<uint32_t> packed;
      const bool pre = choose_slices(L, g, pbits, sum, world, &packed);
      const Remap& mine = per_rank[(size_t)rank][k];
      bool after = pre && mine.after;
      for (uint32_t q : packed) after = after && mine.votes[1 + W + q] == 1.0;
      json += std::string(k ? "," : "") + "{\"pack\":" + (mine.pack ? "1" : "0") + ",\"before\":" + (pre ? "1" : "0") + ",\"after\":" + (after ? "1" : "0") +
              ",\"batch_sweeps_before\":" + std::to_string(mine.sweeps) + ",\"positions\":[";
      for (size_t j = 0; j < packed.size(); ++j) json += (j ? "," : "") + std::to_string(packed[j]);
      json += "]}";
    }
    json += "]}";
    return json.c_str();
  } catch (const std::exception& e) {
    fail(QIP_ERR_INVALID, "internal error: %s", e.what());
    return nullptr;
  } catch (...) {
    fail(QIP_ERR_INVALID, "internal error");
    return nullptr;
  }
}

extern "C" const char* qip_hip_dist_debug_plan(uint32_t n, int dtype, int rank, int world, const qip_op* ops, uint64_t count) {
  static thread_local std::string json;
  try {
    if (count && !ops) return fail(QIP_ERR_INVALID, "null op array"), nullptr;
    if (dtype != QIP_C64 && dtype != QIP_C32) return fail(QIP_ERR_INVALID, "bad dtype %d", dtype), nullptr;
    qipd::DistPlanner pl;
    if (pl.init(n, dtype, rank, world) != QIP_OK) return nullptr;
    std::vector<qipd::Step> steps;
    if (pl.plan(ops, count, &steps) != QIP_OK) return nullptr;
    json = "{\"n\":" + std::to_string(n) + ",\"g\":" + std::to_string(pl.g) + ",\"L\":" + std::to_string(pl.L) + ",\"rank\":" +
           std::to_string(rank) + ",\"steps\":[";
    for (size_t i = 0; i < steps.size(); ++i) {
      if (i) json += ",";
      if (steps[i].kind == qipd::Step::LOCAL) {
        json += "{\"t\":\"local\",\"op\":";
        qipd::json_op(*steps[i].op, &json);
        json += "}";
      } else if (steps[i].kind == qipd::Step::PACK) {
        json += "{\"t\":\"pack\",\"sel\":[";
        for (size_t t = 0; t < steps[i].sel.size(); ++t) json += (t ? "," : "") + std::to_string(steps[i].sel[t]);
        json += "]}";
      } else {
        json += "{\"t\":\"exchange\"}";
      }
    }
    json += "],\"phys\":[";
    for (uint32_t p = 0; p < n; ++p) json += (p ? "," : "") + std::to_string(pl.phys[p]);
    json += "],\"flip\":" + std::to_string(pl.flip);
    {  // the planner's cost model (ms): what an exchange and a gather that does not ride in a tile sweep are priced at, and which
       // of this plan's gathers select a position inside a wave row (those can never ride)
      uint64_t ex = 0, packs = 0, rows = 0;
      const uint32_t p5 = pl.row_p5();
      for (const auto& st : steps) {
        ex += st.kind == qipd::Step::EXCHANGE;
        if (st.kind == qipd::Step::PACK) {
          packs += 1;
          bool r = false;
          for (uint32_t p : st.sel) r = r || tile_is_low(p, p5);
          rows += r;
        }
      }
      char buf[256];
      snprintf(buf, sizeof buf, ",\"model\":{\"exchange_ms\":%.4f,\"pack_ms\":%.4f,\"exchanges\":%llu,\"packs\":%llu,\"packs_from_row_positions\":%llu,\"row_p5\":%u}",
               pl.exchange_ms(), pl.pack_ms(), (unsigned long long)ex, (unsigned long long)packs, (unsigned long long)rows, p5);
      json += buf;
    }
    json += "}";
    return json.c_str();
  } catch (const std::exception& e) {
    fail(QIP_ERR_INVALID, "internal error: %s", e.what());
    return nullptr;
  }
}

