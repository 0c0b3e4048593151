#!/usr/bin/env python
"""apply_op<P> for a REAL P on device slices (qip_hip_apply_op_device): the reference's own f64 bench shapes
(qip-iterators/benches/matmul_bench.rs:19-33 n = 12, :163-177 n = 20: a 2 x 2 matrix of ones on qubit 0, ones in, accumulate)
and the same op at HBM sizes, against the CPU oracle on this box's cores.  Algorithmic bytes per row: sizeof(P) x (input read +
output write, + output read when accumulating).  HIP events on the stream the kernel is launched on (torch's current = null stream)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustqip_amd as q  # noqa: E402
from oracle import qip_oracle as O  # noqa: E402  (the CPU column only)
from rustqip_amd.ops import MatrixOp  # noqa: E402


def gpu_time(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def profile_workload():
    """the fixed sequence rocprofv3 watches (tools/profile_real_p.sh): n = 28, three calls per shape, nothing else on the device"""
    from rustqip_amd import _ffi

    n = 28
    rng = np.random.default_rng(3)
    for dt in (np.float64, np.float32):
        x = rng.standard_normal(1 << n).astype(dt)
        d_in = torch.from_numpy(x).cuda()
        d_out = torch.zeros(1 << n, dtype=d_in.dtype, device="cuda")
        for op, acc in ((MatrixOp.new_matrix([0], [1.0, 1.0, 1.0, 1.0]), True), (MatrixOp.new_matrix([n - 1], rng.standard_normal(4)), False),
                        (MatrixOp.new_control([5], [20], MatrixOp.new_matrix([20], [0, 1, 1, 0])), False),
                        (MatrixOp.new_matrix([2, 9, 17], rng.standard_normal(64)), False)):
            cop = op.to_c(_ffi.QIP_F64 if dt == np.float64 else _ffi.QIP_F32)
            for _ in range(3):
                q.apply_op_device(n, cop, d_in, d_out, accumulate=acc)
            torch.cuda.synchronize()
        del d_in, d_out


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--profile":
        return profile_workload()
    from oracle.bench_parity import host_threads  # (OpenMP default, usable by affinity, cgroup quota, what the CPU legs use)

    threads = host_threads()[3]
    print(f"| n | P | op | accumulate | GPU us / call | algorithmic GB/s | of 8 TB/s | CPU oracle us / call ({threads} threads = the box's cgroup quota; n <= 20: the better of 1 and {threads}) | check |\n|---|---|---|---|---|---|---|---|---|")
    ones = [1.0, 1.0, 1.0, 1.0]
    shapes = [(12, np.float64, "ones on qubit 0 (matmul_bench.rs:19-33)", MatrixOp.new_matrix([0], ones), True),
              (20, np.float64, "ones on qubit 0 (matmul_bench.rs:163-177)", MatrixOp.new_matrix([0], ones), True)]
    rng = np.random.default_rng(3)
    for n in (26, 28):
        for dt in (np.float64, np.float32):
            shapes += [(n, dt, "ones on qubit 0", MatrixOp.new_matrix([0], ones), True),
                       (n, dt, "dense on qubit n-1", MatrixOp.new_matrix([n - 1], rng.standard_normal(4)), False),
                       (n, dt, "dense on qubits 3, n-2", MatrixOp.new_matrix([3, n - 2], rng.standard_normal(16)), False),
                       (n, dt, "CNOT(5 -> 20)", MatrixOp.new_control([5], [20], MatrixOp.new_matrix([20], [0, 1, 1, 0])), False),
                       (n, dt, "Swap(1, n-1)", MatrixOp.new_swap([1], [n - 1]), False)]
            if n == 28:  # what the literal kernel (k_gather_real) still takes: wider ops, SparseMatrix (windows: the GPU tests)
                shapes += [(n, dt, "dense on qubits 2, 6, 11, 17, 23 (literal)", MatrixOp.new_matrix([2, 6, 11, 17, 23], rng.standard_normal(1024)), False),
                           (n, dt, "SparseMatrix on qubits 4, 13, 21, two entries per row (literal)",
                            MatrixOp.new_sparse([4, 13, 21], [[(int(r), 0.5), (int(r ^ 5), -1.25)] for r in range(8)]), False)]
    for n, dt, name, op, acc in shapes:
        N = 1 << n
        x = np.ones(N, dtype=dt) if "ones" in name else rng.standard_normal(N).astype(dt)
        d_in = torch.from_numpy(x).cuda()
        d_out = torch.zeros(N, dtype=d_in.dtype, device="cuda")
        from rustqip_amd import _ffi
        cop = op.to_c(_ffi.QIP_F64 if dt == np.float64 else _ffi.QIP_F32)  # (built once, as the reference's benches do)
        sec = gpu_time(lambda: q.apply_op_device(n, cop, d_in, d_out, accumulate=acc), 200 if n <= 20 else 20)
        by = np.dtype(dt).itemsize * N * (3 if acc else 2)
        one_row_us = None
        if "(literal)" in name:  # A/B inside the run: buffers that start one element off a 16-byte boundary keep one row per lane
            u_in = torch.empty(N + 1, dtype=d_in.dtype, device="cuda")
            u_out = torch.zeros(N + 1, dtype=d_in.dtype, device="cuda")
            u_in[1:].copy_(d_in)
            one_row_us = gpu_time(lambda: q.apply_op_device(n, cop, u_in[1:], u_out[1:], accumulate=acc), 5) * 1e6
            del u_in, u_out
        graph_us = None
        if n <= 20:  # launch-bound sizes: 64 calls recorded into ONE hipGraph (the call is a plain kernel launch on the given stream)
            side = torch.cuda.Stream()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                q.apply_op_device(n, cop, d_in, d_out, accumulate=acc, stream=side.cuda_stream)
                side.synchronize()
                with torch.cuda.graph(g, stream=side):
                    for _ in range(64):
                        q.apply_op_device(n, cop, d_in, d_out, accumulate=acc, stream=side.cuda_stream)
            graph_us = gpu_time(g.replay, 50) / 64 * 1e6
        # check: one more call from a known output against the oracle (bit-equal)
        d_out.zero_()
        q.apply_op_device(n, op, d_in, d_out, accumulate=acc)
        torch.cuda.synchronize()
        want = np.zeros(N, dtype=dt)
        O.apply_op(n, op, x, want, accumulate=acc)
        ok = np.array_equal(d_out.cpu().numpy(), want)
        cpu = None
        for nt in ((1, threads) if n <= 20 else (threads,)):  # (small vectors: one thread or all of them, whichever is faster)
            reps = 20 if n <= 20 else 2
            O.apply_op(n, op, x, want, accumulate=acc, nthreads=nt)
            t = time.perf_counter()
            for _ in range(reps):
                O.apply_op(n, op, x, want, accumulate=acc, nthreads=nt)
            dt_cpu = (time.perf_counter() - t) / reps
            cpu = dt_cpu if cpu is None else min(cpu, dt_cpu)
        print(f"| {n} | {np.dtype(dt).name} | {name} | {int(acc)} | {sec*1e6:.1f} | {by/sec/1e9:.0f} | {by/sec/8e12*100:.1f} % | {cpu*1e6:.0f} | {'bit-equal' if ok else 'DIFFERS'}{'' if graph_us is None else '; %.2f us / call inside a 64-call hipGraph' % graph_us}{'' if one_row_us is None else '; one row per lane (buffers off a 16-byte boundary): %.0f us' % one_row_us} |")
        del d_in, d_out


if __name__ == "__main__":
    main()
