/*
 * qip_hip_debug.h — host-only TEST HOOKS of libqip_hip.so.  Not part of the binding contract (include/qip_hip.h is): a
 * binding for the reference (bindings/rust/qip-hip/src/sys.rs) does not mirror these, and they may change with any build.
 * They serialise what the host half of the library DECIDES (tile plans, permutation descriptors, the sharded state's
 * communication plan, generated kernel sources) so that the CPU test-suite can replay those decisions against the CPU
 * oracle without a GPU.  None of them touches a device.  Exported under the version-script node QIP_HIP_DEBUG.
 */
#ifndef QIP_HIP_DEBUG_H
#define QIP_HIP_DEBUG_H

#include "qip_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Host-only test hook: the descriptor of that sweep as JSON (tile bit positions, LDS swizzle), NULL on error.
 * row_bits = 0 asks for the shape the library launches for 16-byte elements (512-byte rows, thread bit 5 = index position 11
 * on both sides when n >= 12: "tile_bits" 11 or 12); row_bits = 5 / 6 the contiguous-row shapes of 2 * row_bits tile bits;
 * row_bits = 100 the pair form of 8-byte elements whose index bit 0 moves (k_permute_pairs: "fits" false when the tile would
 * need 14 bits or n < 14; fold_bits is ignored). */
const char* qip_hip_debug_permute_plan(uint32_t n, const uint32_t* pi, uint32_t row_bits, uint32_t fold_bits);

/* Host-only: how one pass of a tile sweep lays the thread id over the tile.  pass_bits = the pass's three exchange
 * bits (tile-index space 0..10, ascending); *lanepos gets nibble k = tile-index bit filled by bit k of the 8-bit
 * thread id.  The tile is stored in LDS at slot(t) = t ^ ((t >> S) & (2^S - 1)), S = 4 (QIP_C64) / 5 (QIP_C32);
 * together the two make every pass free of LDS bank conflicts unless it holds both bits of a pair (j, j+S).  Exposed
 * so the claim can be checked without a GPU (tests/test_host_ops.py). */
int qip_hip_tile_lane_assignment(int dtype, const uint32_t* pass_bits, uint64_t* lanepos);

/* Host-only test hook: the complete tile plan of a circuit as a JSON string (owned by the library, valid until
 * the calling thread's next call; NULL on error): the schedule of qip_hip_plan_tiles and, for every multi-gate
 * step, the free bit positions, the passes (exchange bits, lane-bit assignment) and the gate descriptors exactly
 * as they are shipped to k_tile_passes.  tests/test_tile_plan_cpu.py replays it with a numpy model of the kernel
 * and checks the result against the CPU oracle, so the host half of the tile path is covered without a GPU. */
const char* qip_hip_debug_tile_plan(int dtype, uint32_t n, const qip_op* ops, uint64_t count, int mode);

/* Host-only test hook (r4): what the host decides about applying ONE SparseMatrix op in place through the LDS-staged tile
 * kernel (k_sparse_tile) on a state of n qubits — the tile's positions, the block-base descriptor, the row table (entries per
 * row, each stored column's place in the tile, the values) — as JSON, or {"applies":0} when the op takes another kernel.
 * tests/test_tile_plan_cpu.py replays it with a numpy model of the kernel against the CPU oracle.  NULL on error. */
const char* qip_hip_debug_sparse_tile(int dtype, uint32_t n, const qip_op* op);

/* Host-only test hook: generate AND compile (hiprtc cross-compiles for gfx950 without a device) the run-time source
 * of every multi-gate step of the circuit's tile schedule.  *first_source (may be NULL) points at the first segment's
 * source text, owned by the library until the calling thread's next call. */
int qip_hip_debug_tile_jit(int dtype, uint32_t n, const qip_op* ops, uint64_t count, int mode, uint64_t* segments,
                           uint64_t* source_bytes, uint64_t* code_bytes, const char** first_source);

/* Host-only: how one rank's all-to-all of `chunk_bytes` per peer is cut into sends of at most `piece_bytes` — the list the
 * built-in RCCL transport walks inside ONE ncclGroupStart / ncclGroupEnd (peer, byte offset inside the chunk, length; the
 * matching receive has the same three numbers).  Returns the number of pieces; fills at most `cap` entries of each array
 * (any may be NULL).  Test transports use the same list, so the loop is exercised without a second GPU. */
int64_t qip_hip_dist_debug_pieces(int rank, int world, uint64_t chunk_bytes, uint64_t piece_bytes, uint64_t cap,
                                  int32_t* peer, uint64_t* offset, uint64_t* length);

/* Host-only (r5): which of that plan's remaps the overlapped exchange (option "dist_overlap" = `slices`) serves when the local
 * batches run as tile sweeps in scheduler mode `tile_mode` (1 / 2 = "tile", + 16 = wide tiles): JSON
 * {"remaps":[{"pack":0|1,"before":0|1,"after":0|1,"batch_sweeps_before":k}, ...]} — "before": the batch's last sweep is cut into
 * slices and the exchange starts beside it, "after": the next batch's first sweep awaits the slices one by one.  The predicate
 * the executor itself applies; tools/model_scaling.py prices the overlap with it.  NULL on error. */
const char* qip_hip_dist_debug_overlap(uint32_t n, int dtype, int rank, int world, const qip_op* ops, uint64_t count, int tile_mode,
                                       int slices);

/* Host-only test hook: what rank `rank` of `world` would do for this circuit on a fresh state, as a JSON string
 * (owned by the library, valid until the calling thread's next call; NULL on error): the steps
 *   {"t":"local","op":{...}}   the op this rank applies to its shard, in LOCAL qubit indices
 *   {"t":"pack","sel":[...]}   gather these local bit positions into the top g positions (in this order)
 *   {"t":"exchange"}           all-to-all of the top g local bits with the g rank bits
 * and the final layout.  tests/test_distributed_cpu.py replays it with the CPU oracle as the shard and gloo as the
 * transport, so the planner and the per-rank localisation are covered without a GPU. */
const char* qip_hip_dist_debug_plan(uint32_t n, int dtype, int rank, int world, const qip_op* ops, uint64_t count);

#ifdef __cplusplus
}
#endif
#endif /* QIP_HIP_DEBUG_H */
