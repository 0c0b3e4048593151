// qip_tile.h — the host-side tile plan types shared by the scheduler (qip_tile_sched.hip) and the code that runs plans
// (qip_circuit.hip).
#pragma once
#include "qip_internal.h"

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

// Everything the host decides about one segment before anything touches the device: which amplitude-index
// positions the tile's free bits 6..10 stand for, the gate descriptors, the passes and each gate's resolution
// against its pass.  Pure host code: qip_hip_debug_tile_plan serialises it so that tests can replay a plan on
// the CPU (tests/test_tile_plan_cpu.py) and check it against the oracle without a GPU.
template <typename T> struct TileSegmentPlan {
  std::vector<uint32_t> high;  // amplitude-index position of tile bit 6 + j
  std::vector<TileGate<T>> gates;
  std::vector<amp_t<T>> mats;  // 4x4 matrices of the dense 2-qubit gates (kind 3), 16 entries each
  TilePassDesc pd;             // passes (only when `passes`)
  uint32_t p5 = 5;             // amplitude-index position of tile bit 5
  std::vector<uint32_t> order; // gates[i] is the segment's order[i]-th op (build_tile_segment may reorder inside the segment)
};

// r5: what the INTERPRETER kernel (k_tile_passes) is handed for a segment: the plan's gate list with every run of >= 2
// consecutive diagonal gates of a pass replaced by one TOP_DIAG_RUN entry + its TileDiagItem steps (qip_kernels.h).  The plan
// itself — what the run-time generators, the CPU replay and the one-op sweeps consume — is unchanged.
template <typename T> struct TileInterpPlan {
  std::vector<TileGate<T>> gates;
  std::vector<TileDiagItem<T>> items;
  TilePassDesc pd;
  uint32_t runs = 0, gates_in_runs = 0;
};
template <typename T> void tile_merge_diag_runs(const TileSegmentPlan<T>& plan, TileInterpPlan<T>* out, uint32_t min_run = 2);
extern int64_t g_tile_diag_runs;  // global option "tile_diag_runs" (1 = on)

// One step of a tiled schedule: a segment of >= 2 gates applied in one sweep, or a single op applied by
// its own kernel (not tileable, or alone — a lone gate's own kernel touches only what can change).
struct TileStep {
  std::vector<uint64_t> ops;   // indices into the circuit, in application order
  std::vector<uint32_t> high;  // the free bit positions the segment claimed (<= kTileHigh)
  std::vector<uint32_t> perm;  // non-empty: the ops are a run of uncontrolled Swap ops applied as ONE bit-permutation
                               // sweep, new[j] = old[src(j)], bit perm[d] of src(j) = bit d of j (launch_permute)
};

struct TileSchedule {
  const qip_op* circuit = nullptr;  // what the steps' op numbers index: the caller's array, or `owned`
  uint64_t count = 0;
  std::vector<qip_op> owned;               // relabelled: the caller's ops under the labels in force when they run + inserted swaps
  std::deque<std::vector<uint64_t>> idx;   // their index lists
  std::vector<int64_t> origin;             // relabelled: position in the caller's circuit, -1 = inserted swap
  std::vector<TileItem> items;             // one per entry of `circuit`
  std::vector<TileStep> steps;
  uint64_t absorbed = 0, inserted = 0;     // Swap ops turned into label exchanges / in-tile swaps added
  // persistent relabelling (option tile_relabel = 3): the layout the state is in when the plan starts (empty = identity), whether
  // the plan leaves the state relabelled instead of closing with the restoring sweep, and the layout it ends in
  std::vector<uint32_t> init_phys, final_phys;
  bool keep_layout = false;
};

// Which amplitude-index positions are the tile's six LOW bits (lane id at load / store time): 0..4 and p5 (qip_kernels.h,
// tile_block_base).  p5 = 11 ("split rows": two 512-byte halves 32 KiB apart per wave-level access) for Complex<f64> states
// with n >= 12, else 5 (one contiguous row); global option "tile_row_split" (default 11; 5 = contiguous rows everywhere).
extern int64_t g_tile_row_split, g_tile_row_split_f32;
static inline uint32_t tile_p5(int dtype, uint32_t n) {
  // (Complex<f32>: 8-byte amplitudes, a wave-level access is 512 bytes; its own switch, position 12 = the same byte-address bit 15)
  const uint32_t p = (uint32_t)(dtype == QIP_C64 ? g_tile_row_split : g_tile_row_split_f32);
  return (p > 5u && p < n && n >= 13u) || (dtype == QIP_C64 && p > 5u && p < n && n >= 12u) ? p : 5u;
}
template <typename T> static inline uint32_t tile_p5_of(uint32_t n) { return tile_p5(std::is_same<T, double>::value ? QIP_C64 : QIP_C32, n); }
static inline bool tile_is_low(uint32_t pos, uint32_t p5) { return pos < 5u || pos == p5; }
static inline uint32_t tile_low_bit(uint32_t pos) { return pos < 5u ? pos : 5u; }  // (only for positions that are low)
static inline uint64_t tile_low_mask(uint32_t p5) { return 31ull | (1ull << p5); }
// the Ins a tile kernel takes: the high positions opened in the space where p5 and 5 have traded places
Ins tile_ins(const std::vector<uint32_t>& high, uint32_t p5);

// ---- wide tiles (r4, option "tile_wide"): 2^13 amplitudes per block held in REGISTERS (32 per lane = five register bits),
// seven free positions per sweep; LDS is a 32-KiB transposition buffer.  Run-time-compiled segments only.
constexpr int kWideHigh = 7;
constexpr int kWideBits = kTileLow + kWideHigh;  // 13
constexpr int kWideRegBits = 5;
struct WidePass {
  uint32_t R[kWideRegBits];  // tile bit held by register-index bit j
  uint32_t L[8];             // tile bit filled by thread-id bit k
  bool transposed = false;   // the arrangement differs from the previous pass's: the tile goes through LDS in four quarters
  uint32_t q[2] = {0, 0};    // the two tile bits (register bits before AND after) that select the quarter
  uint32_t bufpos[kWideBits] = {0};  // tile bit -> bit of the buffer index (the 11 bits that are not quarter bits)
  uint32_t first = 0, count = 0;     // gates[first .. first + count)
};
template <typename T> struct WidePlan {
  std::vector<uint32_t> high;  // amplitude-index position of tile bit 6 + j (7 of them; high[0], high[1] = the wave bits at load / store)
  uint32_t p5 = 5;
  std::vector<TileGate<T>> gates;  // b0 / b1 / tpos_out / cmask in the 13-bit tile-index space; op, cm_reg, cm_lane unused
  std::vector<amp_t<T>> mats;
  std::vector<WidePass> passes;    // passes.back() may hold no gate: the way back to the load arrangement
  std::vector<uint32_t> order;
};
extern thread_local int t_tile_high;  // free positions per segment the scheduler plans for (kTileHigh, or kWideHigh in wide mode)
template <typename T>
int build_wide_segment(uint32_t n, const std::vector<const TileItem*>& seg, std::vector<uint32_t> high, WidePlan<T>* out, int order_rule = 0);

int classify_tile_item(int dtype, uint32_t n, const qip_op* op, TileItem* it);
template <typename T>
int build_tile_segment(uint32_t n, bool passes, const std::vector<const TileItem*>& seg, std::vector<uint32_t> high, TileSegmentPlan<T>* out,
                       int order_rule = 0, uint32_t p5_override = 0);  // p5_override: 5 / 11 = this tile's sixth low position (one-op sweeps choose)  // order_rule: 0 = gates in the given order, 1 / 2 = fewest passes under tile = 1 / 2's commutation rule
int make_tile_schedule(int dtype, uint32_t n, const qip_op* ops, uint64_t count, int mode, bool allow_2q, TileSchedule* out,
                       bool allow_permute = true);
