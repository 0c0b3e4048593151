"""TEST-ONLY per-rank backend for rustqip_amd.sharded.ShardedState: the CPU oracle stands in for
the HIP kernels so the N > 1 logic (qubit mapping, remap all-to-all, rank-bit controls/diagonals,
reductions) can be exercised with gloo on a box without GPUs.  Never imported by the product."""
from __future__ import annotations

import numpy as np
import torch

from oracle import qip_oracle as O


class OracleBackend:
    def __init__(self, n_local: int):
        self.n_local = n_local
        N = 1 << n_local
        self.bufs = [torch.zeros(N, dtype=torch.complex128) for _ in range(2)]
        self.cur = 0
        self.applied = []

    def _np(self, i):
        return self.bufs[i].numpy()

    def apply_op(self, op):
        self.applied.append(op)
        O.apply_op_overwrite(self.n_local, op, self._np(self.cur), self._np(1 - self.cur))
        self.cur = 1 - self.cur

    def apply_ops(self, ops):
        self.batches = getattr(self, "batches", []) + [len(ops)]
        for op in ops:
            self.apply_op(op)

    def exchange_buffers(self):
        return torch.view_as_real(self.bufs[self.cur]), torch.view_as_real(self.bufs[1 - self.cur])

    def adopt_recv(self):
        self.cur = 1 - self.cur

    def all_to_all(self, dist, recv, send):
        dist.all_to_all_single(recv, send)

    def timed_collective(self, fn):
        fn()

    def collective_ms(self):
        return 0.0

    def init_basis(self, index):
        self.bufs[self.cur].zero_()
        if index is not None:
            self.bufs[self.cur][index] = 1

    def upload(self, x):
        self._np(self.cur)[:] = x

    def download(self):
        return self._np(self.cur).copy()

    def norm_sqr(self):
        return O.prob_magnitude(self._np(self.cur))

    def measure_probs(self, local_qubits):
        return O.measure_probs(self.n_local, list(local_qubits), self._np(self.cur))

    def measure_state(self, local_qubits, measured, prob):
        cur = self._np(self.cur)
        out = self._np(1 - self.cur)
        if O.measure_state(self.n_local, list(local_qubits), (int(measured), float(prob)), cur, out):
            self.cur = 1 - self.cur

    def sync(self):
        pass

    def set_profile(self, v):
        pass

    def take_profile(self):
        return {}

    def reduce_tensor(self, arr):
        return torch.as_tensor(arr, dtype=torch.float64)
