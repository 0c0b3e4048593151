"""Worker for the GPU leg of the N > 1 tests: several ranks share ONE MI355X ("virtual shards").
Everything goes through the C ABI's sharded state (qip_hip_dist_*: C++ planner, pack sweep, real HIP kernels on the
shard); the exchange uses the host-staged transport callbacks over gloo because RCCL refuses two ranks on one device.
With --nccl and world_size 1 the same script drives the built-in RCCL transport (librccl dlopen, unique id, communicator,
all-reduce) that bench.py --gpus N uses."""
import math
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import qip_oracle as O  # noqa: E402
import rustqip_amd as q  # noqa: E402
from rustqip_amd import circuits  # noqa: E402
from rustqip_amd.sharded import DistState  # noqa: E402


def fault_scenario(dist, rank, world):
    """a batch that fails half way must not leave a handle whose layout the data never reached.  The transport's
    all-to-all fails (on every rank alike, so nobody waits in a collective): the call reports it, and every later call on the
    handle — gates, measurement, layout — refuses with the original message instead of computing from a half-moved shard."""
    import ctypes as C

    from rustqip_amd import _ffi
    from rustqip_amd.sharded import HostStagedTransport
    from rustqip_amd.state import _check

    n = 12

    class Failing(HostStagedTransport):
        def _all_to_all(self, ctx, send, recv, chunk_bytes, stream):
            return 1

    t = Failing(dist)
    h = C.c_void_p()
    _check(_ffi.lib.qip_hip_dist_create(n, _ffi.QIP_C64, 0, rank, world, None, C.byref(t.struct), C.byref(h)))
    ops = circuits.h_layer(n)  # H on a rank bit: needs the exchange
    cops = [op.to_c(_ffi.QIP_C64) for op in ops]
    arr = (_ffi.QipOp * len(cops))(*cops)
    rc = _ffi.lib.qip_hip_dist_apply_ops(h, arr, len(cops))
    assert rc != 0 and "all_to_all" in _ffi.last_error(), _ffi.last_error()
    out = C.c_double()
    rc = _ffi.lib.qip_hip_dist_norm_sqr(h, C.byref(out))
    assert rc != 0 and "unusable after an earlier failure" in _ffi.last_error(), _ffi.last_error()
    rc = _ffi.lib.qip_hip_dist_apply_op(h, C.byref(cops[3]))
    assert rc != 0 and "unusable after an earlier failure" in _ffi.last_error()
    _ffi.lib.qip_hip_dist_destroy(h)
    if rank == 0:
        print("ok fault: a failed exchange poisons the handle")


def main():
    use_nccl = "--nccl" in sys.argv
    torch.cuda.set_device(0)
    if use_nccl:
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    g = int(math.log2(world))
    rng = np.random.default_rng(5)
    for n in (12,):
        x = circuits.random_state(n, n)
        u2 = np.linalg.qr(rng.standard_normal((4, 4)) + 1j * rng.standard_normal((4, 4)))[0]
        extra = [q.make_matrix_op([0, n - 1], u2.ravel()), q.make_swap_op([0, 3], [n - 1, 5]),
                 q.make_sparse_matrix_op([1, 0], [[(1, 0.6), (0, 0.8j)], [(0, 0.6), (1, 0.8j)], [(3, 1j)], [(2, -1)]]),
                 q.make_control_op([0], q.make_matrix_op([1], circuits.T)), q.make_matrix_op([0], circuits.rz(0.4))]
        for name, ops in (("c2", circuits.h_layer(n) + circuits.c2_random_circuit(n, 96, seed=28) + extra),
                          ("c4", circuits.c4_clifford_t(n, 96, seed=32)),
                          ("qft", circuits.c3_qft(n)),
                          ("grover_k3", circuits.c5_grover_iteration(n, dense_k3=True))):
            st = DistState(n, dist, 0, host_staged=not use_nccl)
            st.upload_global(x)
            st.apply_ops(ops)
            got = st.download_global()
            want = O.apply_ops_in_place(n, ops, x.copy())
            err = float(np.max(np.abs(got - want)))
            assert err < 1e-12, (name, n, world, err)
            assert abs(st.norm_sqr() - 1) < 1e-12
            for idx in ([0], [n - 1, 0], list(range(5))):
                assert np.max(np.abs(st.measure_probs(idx) - O.measure_probs(n, idx, want))) < 1e-12
            stats = st.comm_stats()
            if world > 1 and name != "grover_k3":
                assert stats["remaps"] >= 1 and stats["bytes_sent_per_rank"] > 0
            # one op at a time (least-recently-used choice instead of look-ahead): same state
            st1 = DistState(n, dist, 0, host_staged=not use_nccl)
            st1.upload_global(x)
            for op in ops[:40]:
                st1.apply_op(op)
            assert np.max(np.abs(st1.download_global() - O.apply_ops_in_place(n, ops[:40], x.copy()))) < 1e-12, name
            # the runs of local gates between remaps as LDS-resident tile sweeps on every shard (tile = 1):
            # IEEE-equal to the gate-by-gate shards
            stt = DistState(n, dist, 0, host_staged=not use_nccl)
            stt.set_option("tile", 1)
            stt.upload_global(x)
            stt.apply_ops(ops)
            if name == "grover_k3":  # its 8x8 gates ride in the sweeps (unfused register fold); gate by gate they ran on the matrix cores
                assert np.max(np.abs(stt.download_global() - got)) < 1e-12, (name, n, world)
            else:
                assert np.array_equal(stt.download_global(), got), (name, n, world)
            # collapsing measurement: forced outcome, then a sampled one (already collapsed, so it repeats)
            ref = O.apply_ops_in_place(n, ops[:30], x.copy())
            for idx, forced in (([0], 1), ([n - 1, 1], 2), ([2, 0, n - 1], 5)):
                st2 = DistState(n, dist, 0, host_staged=not use_nccl)
                st2.upload_global(x)
                st2.apply_ops(ops[:30])
                m, p = st2.measure(idx, measured=forced)
                out = np.zeros_like(ref)
                wm, wp = O.measure(n, idx, ref, out, forced=forced)
                if wp == 0:
                    out = ref
                assert m == wm and abs(p - wp) < 1e-12, (name, idx)
                assert np.max(np.abs(st2.download_global() - out)) < 1e-12, (name, idx)
                ms, ps = st2.measure(idx, rand_u01=0.37 + 0.1 * rank)  # ranks disagree on the sample: rank 0 decides
                if wp > 0:
                    assert ms == forced and abs(ps - 1) < 1e-12
            # sampled outcomes follow the reference's soft_measure map over the whole vector (logical index order)
            for r_u in (0.0137, 0.31, 0.5, 0.77, 0.993):
                for idx in ([0], [n - 1, 3], [5, 0, n - 2, 7]):
                    st3 = DistState(n, dist, 0, host_staged=not use_nccl)
                    st3.upload_global(x)
                    st3.apply_ops(ops[:30])  # a permuted layout whenever the circuit touched a global qubit
                    ms, ps = st3.measure(idx, rand_u01=r_u if rank == 0 else 0.123)
                    wm = O.soft_measure(n, idx, ref, r_u)
                    assert ms == wm and abs(ps - O.measure_prob(n, wm, idx, ref)) < 1e-12, (name, idx, r_u, ms, wm)
            if name in ("c2", "qft"):
                # ... and how OFTEN: 2000 samples through the sampling step alone (no collapse), in a permuted layout, against
                # the oracle's sequential scan.  The descent sums blocks in another order than the scan subtracts amplitudes, so
                # a disagreement needs the sample within rounding (~1e-16) of a boundary: the count is reported and must be 0.
                st4 = DistState(n, dist, 0, host_staged=not use_nccl)
                st4.upload_global(x)
                st4.apply_ops(ops[:30])
                srng = np.random.default_rng(99)
                samples = srng.uniform(0, 1, 400 if use_nccl else 2000)  # (the world-1 RCCL plumbing run repeats this for the transport only)
                idx = [3, 0, n - 1, 6]
                differ = sum(1 for r_u in samples if st4.soft_measure(idx, float(r_u) if rank == 0 else 0.5) != O.soft_measure(n, idx, ref, float(r_u)))
                assert differ == 0, (name, differ)
                if rank == 0:
                    print(f"soft_measure map n={n} world={world} {name}: {differ} of {len(samples)} samples differ from the reference's scan")
            st.init_basis(5)
            e = np.zeros(1 << n, dtype=np.complex128)
            e[5] = 1
            assert np.array_equal(st.download_global(), e)
            if rank == 0:
                print(f"ok n={n} world={world} {name}: err={err:.2e} stats={stats} {st.describe()['transport']}")
    if not use_nccl and world > 1:
        fault_scenario(dist, rank, world)
        # the exchange cut into ragged pieces (the list the RCCL transport walks, over the host-staged transport): same state
        n = 12
        x = circuits.random_state(n, n)
        ops = circuits.h_layer(n) + circuits.c2_random_circuit(n, 64, seed=5) + circuits.c3_qft(n)[:50]
        sp = DistState(n, dist, 0, host_staged=True, piece_bytes=(((1 << (n - g)) * 16 // world) // 3) // 16 * 16)
        sp.upload_global(x)
        sp.apply_ops(ops)
        assert np.max(np.abs(sp.download_global() - O.apply_ops_in_place(n, ops, x.copy()))) < 1e-12
        assert sp._transport.pieces_moved >= 3 * (world - 1)
        if rank == 0:
            print(f"ok pieces: {sp._transport.pieces_moved} pieces moved in {sp.comm_stats()['remaps']} exchanges")
    if not use_nccl and world > 1:
        # r4: the remap's gather folded into the store phase of the tile sweep before it (qip_hip_dist_stats.packs_folded):
        # same amplitudes bit for bit as with a gather sweep of its own, fewer sweeps.  n = 20: real tiles (2^11) on 2^19 shards.
        n = 20
        x = circuits.random_state(n, n)
        ops = circuits.h_layer(n) + circuits.c2_random_circuit(n, 160, seed=11) + circuits.c4_clifford_t(n, 96, seed=5)
        res = {}
        for fold in (0, 1):
            q.set_global_option("dist_fold_pack", fold)
            # (1, 0, 3): shards with a PERSISTENT relabelling (tile_relabel = 3): a batch may leave its qubits relabelled, and then
            # its last sweep addresses other qubits than the gather request names — no fold, the layout is settled first
            # (1, 1, 16): run-time-compiled WIDE segments (r5: they have the packed store too)
            for tile, jit, relabel in ((1, 0, 0), (1, 1, 0), (0, 0, 0), (1, 0, 3), (1, 1, 16)):
                sf = DistState(n, dist, 0, host_staged=True)
                sf.set_option("tile", tile)
                sf.set_option("tile_jit", jit)
                if relabel == 16:
                    sf.set_option("tile_wide", 1)
                elif relabel:
                    sf.set_option("tile_relabel", relabel)
                sf.upload_global(x)
                sf.apply_ops(ops)
                res[(fold, tile, jit, relabel)] = (sf.download_global(), sf.comm_stats())
                sf.close()
        q.set_global_option("dist_fold_pack", 1)
        want = O.apply_ops_in_place(n, ops, x.copy())
        folded_total = 0
        for tile, jit, relabel in ((1, 0, 0), (1, 1, 0), (0, 0, 0), (1, 0, 3), (1, 1, 16)):
            a, sa = res[(0, tile, jit, relabel)]
            b, sb = res[(1, tile, jit, relabel)]
            assert np.array_equal(a, b), (tile, jit, relabel)
            assert np.array_equal(b, res[(1, 1, 0, 0)][0]), (tile, jit, relabel)  # every form of the shard sweeps: the same amplitudes
            assert np.max(np.abs(b - want)) < 1e-12
            assert sa["packs_folded"] == 0 and sb["remaps"] == sa["remaps"]
            assert sb["pack_sweeps"] + sb["packs_folded"] == sa["pack_sweeps"], (sa, sb)
            if relabel == 16:
                assert sb["packs_folded"] >= 1, sb  # the wide sweeps really took the gather
            if not relabel:
                folded_total += sb["packs_folded"]
            if rank == 0:
                print(f"fold tile={tile} jit={jit} relabel={relabel}: remaps={sb['remaps']} pack_sweeps {sa['pack_sweeps']} -> {sb['pack_sweeps']} (folded {sb['packs_folded']})")
        assert folded_total >= 2, folded_total
        if rank == 0:
            print("ok fold: the remap's gather rides in the preceding tile sweep")
    if not use_nccl and world > 1:
        # r5: gate-by-gate shards at a size where pair_floor is active (n_local = 22): a pair in the MIDDLE of a batch that a remap
        # follows must not take the batch's packed-store request (found by the n = 29 twin: the gates after it ran on the packed buffer)
        n = 23
        x = circuits.random_state(n, n)
        ops = circuits.h_layer(n)[:6] + circuits.c4_clifford_t(n, 90, seed=32) + circuits.c2_random_circuit(n, 60, seed=28)
        want = O.apply_ops_in_place(n, ops, x.copy())
        got = {}
        for pair in (1, 0):
            sp = DistState(n, dist, 0, host_staged=True)
            sp.set_option("pair_floor", pair)
            sp.set_option("profile", 1)
            sp.upload_global(x)
            sp.apply_ops(ops)
            got[pair] = (sp.download_global(), sum(v["launches"] for v in sp.take_profile().values()), sp.comm_stats())
            sp.close()
        assert np.array_equal(got[1][0], got[0][0]) and np.max(np.abs(got[1][0] - want)) < 1e-12
        assert got[1][1] < got[0][1] and got[1][2]["remaps"] >= 1, (got[1][1], got[0][1], got[1][2])
        if rank == 0:
            print(f"ok pair_floor on shards: {got[0][1]} -> {got[1][1]} launches, {got[1][2]['remaps']} remaps, folded {got[1][2]['packs_folded']}")
    if not use_nccl and world > 1:
        # r5, option dist_overlap: the exchange in P slices on the communication stream, each sent as soon as the LAST tile sweep
        # before the remap (launched in P parts, packed store included) has stored it, the FIRST sweep after the remap starting
        # on each slice as soon as it has landed.  Same amplitudes bit for bit as the serial remap, for every form of the sweeps.
        n = 21
        x = circuits.random_state(n, n)
        ops = circuits.h_layer(n) + circuits.c2_random_circuit(n, 200, seed=13) + circuits.c4_clifford_t(n, 120, seed=7) + circuits.c3_qft(n)[:150]
        want = O.apply_ops_in_place(n, ops, x.copy())
        base = None
        seen_overlap = seen_after = 0
        for slices, tile, jit, wide in ((0, 1, 0, 0), (4, 1, 0, 0), (2, 1, 0, 0), (4, 1, 1, 0), (4, 1, 1, 1), (8, 2, 1, 0)):
            so = DistState(n, dist, 0, host_staged=True)
            so.set_option("tile", tile)
            so.set_option("tile_jit", jit)
            so.set_option("tile_wide", wide)
            so.set_option("dist_overlap", slices)
            so.upload_global(x)
            so.apply_ops(ops)
            got = so.download_global()
            cs = so.comm_stats()
            assert abs(so.norm_sqr() - 1) < 1e-12
            so.close()
            assert np.max(np.abs(got - want)) < 1e-12, (slices, tile, jit, wide)
            if slices == 0:
                base = (got, cs)
                assert cs["remaps_overlapped"] == 0 and cs["slices_overlapped"] == 0
            else:
                if tile == 1:
                    assert np.array_equal(got, base[0]), (slices, tile, jit, wide)  # IEEE-equal sweeps: the very same bits
                assert cs["remaps"] == base[1]["remaps"], (cs, base[1])
                assert cs["slices_overlapped"] == slices * cs["remaps_overlapped"]
                seen_overlap += cs["remaps_overlapped"]
                seen_after += cs["remaps_overlapped_after"]
            if rank == 0:
                print(f"overlap slices={slices} tile={tile} jit={jit} wide={wide}: remaps={cs['remaps']} overlapped={cs['remaps_overlapped']} "
                      f"(after too: {cs['remaps_overlapped_after']}) folded={cs['packs_folded']} pack_sweeps={cs['pack_sweeps']}")
        assert seen_overlap >= 3 and seen_after >= 1, (seen_overlap, seen_after)
        if rank == 0:
            print("ok overlap: the exchange in slices beside the neighbouring tile sweeps changes nothing")
    # f32 shards
    n = 11
    xf = circuits.random_state(n, 3, np.complex64)
    ops = circuits.h_layer(n) + circuits.c2_random_circuit(n, 48, seed=3)
    sf = DistState(n, dist, 0, np.complex64, host_staged=not use_nccl)
    sf.upload_global(xf)
    sf.apply_ops(ops)
    assert np.max(np.abs(sf.download_global() - O.apply_ops_in_place(n, ops, xf.copy()))) < 1e-5
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
