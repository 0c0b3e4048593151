#!/usr/bin/env python
"""bench.py — gate-application throughput on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched under torch.distributed.run)

A "step" is one pass of the hot path over one batch of synthetic input: the configs[1] circuit
(random single-qubit H / X / Rz + CNOT, 256 gates) applied once to a
2^n-amplitude Complex<f64> state that is already resident in HBM.  n = n_local + log2(N) with
n_local = 30 amplitudes-bits per GPU by default (the size BASELINE.json's target is quoted on:
n=30 on 1 GPU, n=33 on 8); weak scaling.  value = algorithmic GB/s of the whole job
(sum over gates of the bytes of SURVEY.md §8(d)'s table, / wall time), gates/s beside it.

Also on the same JSON line:
  roofline      dominant kernel: algorithmic bytes per launch / mean launch duration, measured with
                HIP events on the launching stream inside the timed region
  cpu_baseline  the CPU oracle (C restatement of qip-iterators apply_op_overwrite, OpenMP over all
                host cores) timed on a bounded sample of the same circuit (rank 0, N = 1 only)
  extras        configs[1] exactly (n = 28) and the per-target-qubit H sweep at n_local
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n-local", type=int, default=30, help="qubits per GPU shard (2^n_local amplitudes)")
    ap.add_argument("--gates", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--cpu-budget-s", type=float, default=20.0)
    return ap.parse_args()


def circuit_bytes(q, n, ops):
    return [q.algorithmic_bytes(n, op) for op in ops]


def dominant_kernel(profile):
    if not profile:
        return None, None
    name = max(profile, key=lambda k: profile[k]["total_ms"])
    return name, profile[name]


def load_traffic(kernel_name):
    """HBM bytes per launch from the committed rocprofv3 PMC summary (profiles/), or None."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(kernel_name, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def cpu_baseline(q, circuits, args):
    """Time the oracle on a bounded sample of the same workload: the first gates of the same
    seeded circuit at the largest n <= 28 whose predicted cost fits the budget."""
    import numpy as np

    from oracle import qip_oracle as O

    threads = O.max_threads()

    def run(n, ops):
        state = np.zeros(1 << n, dtype=np.complex128)
        state[0] = 1
        arena = np.zeros_like(state)  # touch both buffers before timing
        for op in circuits.h_layer(n)[:2]:
            O.apply_op_overwrite(n, op, state, arena)
            state, arena = arena, state
        t0 = time.perf_counter()
        for op in ops:
            O.apply_op_overwrite(n, op, state, arena)
            state, arena = arena, state
        return time.perf_counter() - t0

    n_cal, n_gates = 22, 16
    ops_cal = circuits.c2_random_circuit(n_cal, n_gates, seed=28)
    t_cal = run(n_cal, ops_cal)
    n_cpu = n_cal
    while n_cpu < 28 and t_cal * (2 ** (n_cpu + 1 - n_cal)) <= args.cpu_budget_s:
        n_cpu += 1
    ops = circuits.c2_random_circuit(n_cpu, n_gates, seed=28)
    t = run(n_cpu, ops) if n_cpu != n_cal else t_cal
    by = sum(circuit_bytes(q, n_cpu, ops))
    return {
        "value": by / t / 1e9, "unit": "GB/s", "cores": threads, "kind": "port",
        "gates_per_s": n_gates / t, "ms_per_gate": 1e3 * t / n_gates,
        "sample": f"first {n_gates} gates of the same seeded C2 circuit at n={n_cpu} (2 buffers x {16 * 2**n_cpu / 2**30:.2f} GiB), "
                  f"C restatement of qip-iterators 1.5.0 apply_op_overwrite, gcc -O2 -fopenmp, {threads} threads, {t:.1f} s",
    }


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    g = int(math.log2(world))
    if 1 << g != world:
        raise SystemExit("number of GPUs must be a power of two")

    import numpy as np
    import torch

    import rustqip_amd as q
    from rustqip_amd import circuits

    if not torch.cuda.is_available() or q.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device: rustqip_amd has no CPU fallback")
    # QIP_BENCH_DIST_BACKEND=gloo is a TEST hook: several ranks share one GPU and the remap all-to-all is
    # staged through host memory, so the N > 1 code path can be exercised where only one GPU exists.
    # Real multi-GPU runs use nccl (= RCCL) on device buffers.
    dist_backend = os.environ.get("QIP_BENCH_DIST_BACKEND", "nccl")
    device = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device)
    dist = None
    if world > 1:
        import torch.distributed as dist

        if dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend=dist_backend)

    n = args.n_local + g
    ops = circuits.c2_random_circuit(n, args.gates, seed=28)
    bytes_per_step = sum(circuit_bytes(q, n, ops))  # whole job (all ranks)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if world == 1:
        st = q.HipState(n, np.complex128, device=device)
        st.init_basis(0)
        st.apply_ops(circuits.h_layer(n))  # dense state, every amplitude 2^(-n/2)
        compiled = st.compile_ops(ops)
        run_step = lambda: st.apply_compiled(compiled)
        sync = st.sync
        set_profile = lambda v: st.set_option("profile", v)
        get_profile = lambda: (st.profile(), st.profile_reset())[0]
    else:
        from rustqip_amd.sharded import HipBackend, ShardedState

        st = ShardedState(n, dist, backend=HipBackend(args.n_local, device, host_staged_exchange=dist_backend != "nccl"))
        st.init_basis(0)
        st.apply_ops(circuits.h_layer(n))
        plan = st.plan(ops)
        run_step = lambda: st.run_plan(plan)
        sync = st.sync
        set_profile = st.set_profile
        get_profile = st.take_profile

    for _ in range(args.warmup):
        run_step()
    sync()
    set_profile(1)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_step()
    sync()
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if dist_backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    profile = get_profile()
    set_profile(0)
    norm = st.norm_sqr()

    ms_per_step = 1e3 * elapsed / args.steps
    value = bytes_per_step * args.steps / elapsed / 1e9
    kname, kstat = dominant_kernel(profile)
    roofline = None
    if kstat:
        per_launch_bytes = kstat["algorithmic_bytes"] / kstat["launches"]
        avg_ms = kstat["total_ms"] / kstat["launches"]
        achieved = per_launch_bytes / (avg_ms * 1e-3) / 1e9
        roofline = {
            "bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS, "traffic": load_traffic(kname),
            "avg_launch_ms": avg_ms, "launches": kstat["launches"], "algorithmic_bytes_per_launch": per_launch_bytes,
        }
    kernels = {
        k: {"launches": v["launches"], "avg_ms": v["total_ms"] / v["launches"],
            "GBps": (v["algorithmic_bytes"] / v["launches"]) / (v["total_ms"] / v["launches"] * 1e-3) / 1e9}
        for k, v in (profile or {}).items() if v["launches"] and v["total_ms"] > 0
    }

    extras = {}
    if world == 1 and not args.no_extras:
        # per-target-qubit H sweep at n_local: GB/s by target qubit
        sweep = []
        st.set_option("profile", 0)
        for tq in range(n):
            op = st.compile_ops([q.make_matrix_op([tq], circuits.H)] * 4)
            st.apply_compiled(op)
            st.sync()
            t = time.perf_counter()
            st.apply_compiled(op)
            st.apply_compiled(op)
            st.sync()
            dt = (time.perf_counter() - t) / 8
            sweep.append(round(32.0 * 2**n / dt / 1e9, 1))
        extras["h_sweep_GBps_by_target_qubit"] = sweep
        extras["h_sweep_min_median_GBps"] = [min(sweep), float(np.median(sweep))]
        single = circuits.c2_random_circuit(n, args.gates, seed=30, single_only=True)
        cs = st.compile_ops(single)
        st.apply_compiled(cs)
        st.sync()
        t = time.perf_counter()
        st.apply_compiled(cs)
        st.sync()
        dt = time.perf_counter() - t
        extras["single_qubit_only"] = {"n": n, "gates": len(single), "GBps": sum(circuit_bytes(q, n, single)) / dt / 1e9,
                                       "gates_per_s": len(single) / dt, "frac_of_8TBps": sum(circuit_bytes(q, n, single)) / dt / 1e9 / HBM_PEAK_GBPS}
        # gate fusion: the same circuit with option fuse = 5 — one sweep per fused gate
        st.set_option("fuse", 5)
        st.set_option("profile", 1)
        st.profile_reset()
        cf = st.compile_ops(ops)
        st.apply_compiled(cf)
        st.sync()
        st.profile_reset()
        t = time.perf_counter()
        st.apply_compiled(cf)
        st.sync()
        dt = time.perf_counter() - t
        prof_f = st.profile()
        sweeps = sum(v["launches"] for v in prof_f.values())
        sweep_bytes = sum(v["algorithmic_bytes"] for v in prof_f.values())
        extras["fused_k5"] = {"gates": len(ops), "sweeps": sweeps, "gates_per_s": len(ops) / dt, "ms_per_step": 1e3 * dt,
                              "sweep_GBps": sweep_bytes / dt / 1e9,
                              "note": "per-sweep bytes (32*2^n per fused dense gate), never per-gate bytes over sweep time"}
        st.set_option("fuse", 0)
        # LDS-resident multi-gate sweeps: tile = 1 (circuit order up to exact commutations, IEEE-equal) and tile = 2 (commuting reorder)
        for mode in (1, 2):
            st.set_option("tile", mode)
            ct = st.compile_ops(ops)
            st.apply_compiled(ct)
            st.sync()
            st.profile_reset()
            t = time.perf_counter()
            st.apply_compiled(ct)
            st.sync()
            dt = time.perf_counter() - t
            prof_t = st.profile()
            sweeps = sum(v["launches"] for v in prof_t.values())
            extras["tiled_mode%d" % mode] = {"gates": len(ops), "sweeps": sweeps, "gates_per_s": len(ops) / dt, "ms_per_step": 1e3 * dt,
                                             "sweep_GBps": sum(v["algorithmic_bytes"] for v in prof_t.values()) / dt / 1e9}
        st.set_option("tile", 0)
        st.set_option("profile", 0)
        # the other single-GPU configs of BASELINE.json on the same resident state size
        for cname, cops in (("configs2_qft_n%d" % n, circuits.c3_qft(n)),
                            ("configs4_grover_iteration_n%d" % n, circuits.c5_grover_iteration(n)),
                            ("configs4_grover_dense_k3_n%d" % n, circuits.c5_grover_iteration(n, dense_k3=True))):
            cc = st.compile_ops(cops)
            st.apply_compiled(cc)
            st.sync()
            t = time.perf_counter()
            st.apply_compiled(cc)
            st.sync()
            dt = time.perf_counter() - t
            by = sum(circuit_bytes(q, n, cops))
            extras[cname] = {"ops": len(cops), "ms": 1e3 * dt, "ops_per_s": len(cops) / dt, "algorithmic_GBps": by / dt / 1e9}
            st.set_option("tile", 1)  # IEEE-equal multi-gate sweeps
            st.set_option("profile", 1)
            st.apply_compiled(cc)
            st.sync()
            st.profile_reset()
            t = time.perf_counter()
            st.apply_compiled(cc)
            st.sync()
            dt = time.perf_counter() - t
            extras[cname]["tile1"] = {"ms": 1e3 * dt, "ops_per_s": len(cops) / dt,
                                      "sweeps": sum(v["launches"] for v in st.profile().values())}
            st.set_option("tile", 0)
            st.set_option("profile", 0)
        extras["norm_sqr_end"] = st.norm_sqr()
        st.close()
        # configs[1] exactly: n = 28
        n28 = 28
        ops28 = circuits.c2_random_circuit(n28, args.gates, seed=28)
        with q.HipState(n28) as s28:
            s28.init_basis(0)
            s28.apply_ops(circuits.h_layer(n28))
            c28 = s28.compile_ops(ops28)
            s28.apply_compiled(c28)
            s28.sync()
            t = time.perf_counter()
            for _ in range(3):
                s28.apply_compiled(c28)
            s28.sync()
            dt = (time.perf_counter() - t) / 3
            extras["configs1_n28"] = {"GBps": sum(circuit_bytes(q, n28, ops28)) / dt / 1e9, "gates_per_s": len(ops28) / dt,
                                      "ms_per_step": 1e3 * dt, "norm_sqr": s28.norm_sqr()}

    if world > 1 and not args.no_extras:
        # the same step with the runs of local gates between remaps applied as LDS-resident tile sweeps on every
        # shard (tile = 1: IEEE-equal to gate by gate).  Reported beside the headline, never as `value`; guarded so
        # that nothing here can take the bench line down.
        try:
            st.backend.state.set_option("tile", 1)
            st.run_plan(plan, batched=True)
            sync()
            barrier()
            t = time.perf_counter()
            st.run_plan(plan, batched=True)
            sync()
            barrier()
            dt = time.perf_counter() - t
            tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if dist_backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            extras["tiled_mode1"] = {"gates": len(ops), "ms_per_step": 1e3 * float(tt.item()),
                                     "gates_per_s": len(ops) / float(tt.item())}
            st.backend.state.set_option("tile", 0)
        except Exception as exc:  # noqa: BLE001
            extras["tiled_mode1"] = {"error": repr(exc)}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(q, circuits, args)

    if rank == 0:
        line = {
            "metric": "single-qubit gate apply GB/s (algorithmic bytes, random H/X/Rz+CNOT circuit)",
            "value": value, "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": f"configs[1] generator (random H/X/Rz 3/4 + CNOT 1/4, seed 28, {args.gates} gates) at n={n} "
                            f"({args.n_local} qubits = {16 * 2**args.n_local / 2**30:.0f} GiB per GPU), Complex<f64>, state = H^n|0>",
                "n_qubits": n, "n_local": args.n_local, "gates_per_step": args.gates,
                "algorithmic_bytes_per_step": bytes_per_step,
                "parallelism": "single GPU" if world == 1 else f"state sharded by top {g} index bits over {world} GPUs, RCCL all-to-all qubit remap",
            },
            "gates_per_s": args.gates * args.steps / elapsed,
            "frac_of_hbm_peak_per_gpu": value / world / HBM_PEAK_GBPS,
            "norm_sqr_after": norm,
            "roofline": roofline,
            "kernels": kernels,
            "cpu_baseline": cpu,
        }
        if extras:
            line["extras"] = extras
        if world > 1:
            line["comm"] = st.comm_stats()
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
