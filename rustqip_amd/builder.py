"""HipBuilder — the run loop of LocalBuilder<P> with the amplitudes on the GPU.

Mirrors the part of the reference's circuit API that feeds the hot path:
  qip/src/builder.rs:325-519   LocalBuilder: registers, pipeline recording, broadcast of 1-qubit
                               objects (:382-387), calculate_state_with_init (:400-519) incl. the
                               gate -> MatrixOp lowering table (:436-498)
  qip/src/builder_traits.rs    CliffordTBuilder defaults: s_dagger/t_dagger (:408-422), cnot
                               (:425-451), swap via 3 CNOTs (:454-482); basic_toffoli (:505-538),
                               toffoli for <= 2 controls (:541-568)
  qip/src/builder.rs:663-815   try_apply_with_condition: X, Y, Z, CNOT, SWAP arms
  qip/src/conditioning.rs      Conditioned wrapper (condition_with / dissolve)
Only calculate_state_with_init touches amplitudes, and it runs entirely through HipState
(HIP kernels).  Registers are plain index lists: Rust's move semantics are not modelled.

Deliberate deviation: the reference swaps its two buffers after
*every* stage, including GlobalPhase / stochastic measurements that wrote nothing, so later
stages read a stale buffer.  The device state is updated in place and has no such hazard.
"""
from __future__ import annotations

import cmath
import math
from fractions import Fraction
from dataclasses import dataclass
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np

from .ops import CircuitError, MatrixOp, make_control_op, make_matrix_op, make_swap_op
from .state import HipState


@dataclass(frozen=True)
class Register:
    """qip::builder::Qudit (builder.rs:60-108): an ordered list of qubit indices."""

    indices: Tuple[int, ...]

    def n(self) -> int:
        return len(self.indices)


@dataclass
class PipelineEntry:
    indices: List[int]
    kind: str  # X Y Z H S T CNOT SWAP Rz MAT GlobalPhase Measurement StochasticMeasurement
    param: object = None  # Rz: theta (radians); MAT: 4^k complex entries


@dataclass
class Measurements:
    """qip::builder::Measurements (builder.rs:262-323)."""

    results: List[tuple]

    def get_measurement(self, handle: int):
        kind, a, b = self.results[handle]
        return (a, b) if kind == "Single" else None

    def get_stochastic_measurement(self, handle: int):
        kind, a, _ = self.results[handle]
        return a if kind == "Stochastic" else None


def lower_to_matrix_op(entry: PipelineEntry) -> Optional[MatrixOp]:
    """The lowering table of the reference run loop (builder.rs:436-498)."""
    idx = list(entry.indices)
    l, o, i = 1.0 + 0.0j, 0.0 + 0.0j, 1j
    k = entry.kind
    if k == "X":
        return make_matrix_op(idx, [o, l, l, o])
    if k == "Y":
        return make_matrix_op(idx, [o, -i, i, o])
    if k == "Z":
        return make_matrix_op(idx, [l, o, o, -l])
    if k == "H":
        # nl = Complex::one() * FRAC_1_SQRT_2 = (s, 0); -nl = (-s, -0.0)   (:448-450)
        s = math.sqrt(0.5)
        nl = complex(s, 0.0)
        return make_matrix_op(idx, [nl, nl, nl, -nl])
    if k == "S":
        return make_matrix_op(idx, [l, o, o, i])
    if k == "T":
        return make_matrix_op(idx, [l, o, o, cmath.rect(1.0, math.pi / 4)])  # from_polar (:453-458)
    if k == "CNOT":
        return make_control_op([idx[0]], make_matrix_op(idx[1:], [o, l, l, o]))  # :460-467
    if k == "MAT":
        return make_matrix_op(idx, entry.param)  # :468-470
    if k == "SWAP":
        if len(idx) % 2 != 0:
            raise CircuitError("SWAP needs an even number of indices")
        x = len(idx) // 2
        return make_swap_op(idx[:x], idx[x:])  # :471-478
    if k == "Rz":
        theta = entry.param
        if isinstance(theta, Fraction):
            # RotationObject::PiRational(r) is lowered as r radians — WITHOUT pi — by the reference
            # (builder.rs:480-486).  Mirrored as written, so results match LocalBuilder.
            theta = float(theta)
        h_theta = float(theta) * 0.5  # :486-495
        return make_matrix_op(idx, [cmath.rect(1.0, -h_theta), o, o, cmath.rect(1.0, h_theta)])
    if k == "GlobalPhase":
        return None  # :431-432
    raise CircuitError(f"cannot lower pipeline entry {k!r}")


class HipBuilder:
    """Drop-in for the LocalBuilder<P> call sequence of the reference, GPU-backed."""

    def __init__(self, dtype=np.complex128, device: int = 0, tile: int = 1, tile_relabel: int = 1):
        self.dtype = np.dtype(dtype)
        self.device = device
        # tile_relabel = 1: the scheduler may relabel the qubits when that shortens the plan (only moves are added: bit-identical
        # for tile = 1).  It uses the second buffer — the reference's run loop holds `state` and `arena` as well
        # (builder.rs:406-407); a state too large for it keeps the plain plan.
        self.tile_relabel = tile_relabel
        # tile = 1: runs of gates between measurements are applied as LDS-resident multi-gate sweeps in circuit
        # order — IEEE-equal to one sweep per gate, several times fewer passes over HBM.  0 = one sweep per
        # gate, 2 = also hoist gates over gates they commute with (1e-12 instead of bit equality).
        self.tile = tile
        self._n = 0
        self.pipeline: List[PipelineEntry] = []

    # ---- registers (builder.rs:338-372) ------------------------------------------------
    def n(self) -> int:
        return self._n

    def register(self, n: int) -> Register:
        if n <= 0:
            raise CircuitError("register size must be non-zero")
        r = Register(tuple(range(self._n, self._n + n)))
        self._n += n
        return r

    def qubit(self) -> Register:
        return self.register(1)

    @staticmethod
    def merge_two_registers(r1: Register, r2: Register) -> Register:
        return Register(r1.indices + r2.indices)

    @staticmethod
    def split_all_register(r: Register) -> List[Register]:
        return [Register((i,)) for i in r.indices]

    # ---- recording (builder.rs:374-398) ---------------------------------------------------
    def _apply(self, r: Register, kind: str, width: int, param=None) -> Register:
        """apply_circuit_object: a 1-qubit object on a wider register is broadcast (:382-387)."""
        if width == 1 and r.n() > 1:
            for i in r.indices:
                self.pipeline.append(PipelineEntry([i], kind, param))
        elif width in (0, r.n()):
            self.pipeline.append(PipelineEntry(list(r.indices), kind, param))
        else:
            raise CircuitError("Matrix has incorrect N and cannot be broadcast")
        return r

    # CliffordTBuilder
    def x(self, r): return self._apply(r, "X", 1)
    def y(self, r): return self._apply(r, "Y", 1)
    def z(self, r): return self._apply(r, "Z", 1)
    def h(self, r): return self._apply(r, "H", 1)
    def s(self, r): return self._apply(r, "S", 1)
    def t(self, r): return self._apply(r, "T", 1)
    def not_(self, r): return self.x(r)

    def s_dagger(self, r):  # builder_traits.rs:419-422
        return self.s(self.z(r))

    def t_dagger(self, r):  # :408-411
        return self.t(self.s_dagger(r))

    def cnot(self, cr: Register, r: Register):  # :425-451
        if cr.n() > 1:
            raise CircuitError("Clifford CNOT can only have a single control qubit.")
        for q in r.indices:
            self.pipeline.append(PipelineEntry([cr.indices[0], q], "CNOT"))
        return cr, r

    def swap(self, ra: Register, rb: Register):  # :454-482 (three CNOTs per qubit pair)
        if ra.n() != rb.n():
            raise CircuitError("Swap must be between registers of the same size.")
        for a, b in zip(ra.indices, rb.indices):
            qa, qb = Register((a,)), Register((b,))
            self.cnot(qa, qb)
            self.cnot(qb, qa)
            self.cnot(qa, qb)
        return ra, rb

    def swap_op(self, ra: Register, rb: Register):
        """A native SWAP pipeline object (UnitaryMatrixObject::SWAP, lowered at builder.rs:471-478)."""
        if ra.n() != rb.n():
            raise CircuitError("Swap must be between registers of the same size.")
        self.pipeline.append(PipelineEntry(list(ra.indices + rb.indices), "SWAP"))
        return ra, rb

    # RotationsBuilder / UnitaryBuilder
    def rz(self, r: Register, theta: float):
        return self._apply(r, "Rz", 1, float(theta))

    def rz_ratio(self, r: Register, theta: Fraction):
        """RotationsBuilder::rz_ratio: Rz with a PiRational angle (exported to QASM as k*pi/m)."""
        return self._apply(r, "Rz", 1, Fraction(theta))

    def rz_pi_by(self, r: Register, m: int):
        """RotationsBuilder::rz_pi_by: Rz(pi/m) as PiRational(1/m) (see the Q1 note in lower_to_matrix_op)."""
        if m == 0:
            raise CircuitError("rz_pi_by: denominator must be non-zero")
        return self.rz_ratio(r, Fraction(1, m))

    def apply_global_phase(self, r: Register, theta: float):
        """builder.rs:32-56: recorded, no effect on the state (builder.rs:431-432)."""
        self.pipeline.append(PipelineEntry(list(r.indices), "GlobalPhase", float(theta)))
        return r

    def apply_matrix(self, r: Register, data):  # UnitaryBuilder::apply_vec_matrix
        data = np.asarray(data, dtype=np.complex128).ravel()
        if data.size == 4 and r.n() > 1:
            return self._apply(r, "MAT", 1, data)
        if data.size != 4 ** r.n():
            raise CircuitError("Matrix has incorrect N and cannot be broadcast")
        return self._apply(r, "MAT", r.n(), data)

    # AdvancedCircuitBuilder
    def basic_toffoli(self, cr: Register, r: Register):  # :505-538
        if cr.n() != 2:
            raise CircuitError("Basic Toffoli can only be applied to two control qubits.")
        cra, crb = Register(cr.indices[:1]), Register(cr.indices[1:])
        r = self.h(r)
        self.cnot(crb, r)
        r = self.t_dagger(r)
        self.cnot(cra, r)
        r = self.t(r)
        self.cnot(crb, r)
        r = self.t_dagger(r)
        self.cnot(cra, r)
        crb = self.t(crb)
        r = self.t(r)
        self.cnot(cra, crb)
        r = self.h(r)
        cra = self.t(cra)
        crb = self.t_dagger(crb)
        self.cnot(cra, crb)
        return cr, r

    def toffoli(self, cr: Register, r: Register):  # :541-568
        if cr.n() == 1:
            return self.cnot(cr, r)
        if cr.n() == 2:
            return self.basic_toffoli(cr, r)
        raise CircuitError("toffoli with more than two controls needs temporary qubits; "
                           "build a native Control MatrixOp instead (ops.rs:19)")

    # Conditionable (builder.rs:663-815) — the arms the CSWAP example exercises, plus Y/Z
    def try_apply_with_condition(self, cr: Register, r: Register, kind: str):
        if kind == "X":
            return self.toffoli(cr, r)
        if kind == "Y":
            r = self.s(r)
            cr, r = self.toffoli(cr, r)
            return cr, self.s_dagger(r)
        if kind == "Z":
            r = self.h(r)
            cr, r = self.toffoli(cr, r)
            return cr, self.h(r)
        if kind == "CNOT":
            if r.n() != 2:
                raise CircuitError("conditioned CNOT needs a 2-qubit register")
            ra, rt = Register(r.indices[:1]), Register(r.indices[1:])
            self.toffoli(self.merge_two_registers(cr, ra), rt)
            return cr, r
        raise CircuitError(f"conditioning {kind} is not part of this mirror")

    def condition_with(self, cr: Register) -> "Conditioned":
        return Conditioned(self, cr)

    # MeasurementBuilder / StochasticMeasurementBuilder (builder.rs:599-637)
    def measure(self, r: Register):
        self.pipeline.append(PipelineEntry(list(r.indices), "Measurement"))
        return r, sum(1 for e in self.pipeline if e.kind in ("Measurement", "StochasticMeasurement")) - 1

    def measure_stochastic(self, r: Register):
        self.pipeline.append(PipelineEntry(list(r.indices), "StochasticMeasurement"))
        return r, sum(1 for e in self.pipeline if e.kind in ("Measurement", "StochasticMeasurement")) - 1

    # ---- run (builder.rs:400-519) -------------------------------------------------------
    def initial_index(self, init: Iterable[Tuple[Register, int]]) -> int:
        n = self._n
        index = 0
        for r, x in init:  # :409-421: bit k of x sets qubit r.indices[k]
            for k, q in enumerate(r.indices):
                index |= ((x >> k) & 1) << (n - 1 - q)
        return index

    def calculate_state_with_init(self, init: Iterable[Tuple[Register, int]] = (), *,
                                  forced_measurements: Optional[Sequence[int]] = None,
                                  rng: Optional[np.random.Generator] = None,
                                  return_state: bool = True):
        """Returns (state as numpy array or the live HipState, Measurements).

        forced_measurements plays MeasuredCondition (measurement_ops.rs:181-186) for the
        collapsing measurements, in order; otherwise a uniform sample is drawn from `rng`
        (the reference uses the unseeded thread RNG, :160).
        """
        n = self._n
        if n == 0:
            raise CircuitError("empty circuit")
        rng = rng or np.random.default_rng()
        forced = list(forced_measurements or [])
        state = HipState(n, self.dtype, self.device)
        state.set_option("tile", self.tile)
        state.set_option("tile_relabel", self.tile_relabel if self.tile >= 1 else 0)
        state.init_basis(self.initial_index(init))
        results: List[tuple] = []
        batch: List[MatrixOp] = []

        def flush():
            if batch:
                state.apply_ops(batch)
                batch.clear()

        for entry in self.pipeline:
            if entry.kind == "Measurement":
                flush()
                want = forced.pop(0) if forced else None
                m, p = state.measure(entry.indices, want, float(rng.random()))
                results.append(("Single", m, p))
            elif entry.kind == "StochasticMeasurement":
                flush()
                results.append(("Stochastic", state.measure_probs(entry.indices), None))
            else:
                op = lower_to_matrix_op(entry)
                if op is not None:
                    batch.append(op)
        flush()
        meas = Measurements(results)
        if return_state:
            out = state.download()
            state.close()
            return out, meas
        return state, meas

    def calculate_state(self, **kw):
        return self.calculate_state_with_init((), **kw)


class Conditioned:
    """qip::conditioning::Conditioned (conditioning.rs:20-128): every object applied through it
    goes to parent.try_apply_with_condition with the held control register."""

    def __init__(self, parent: HipBuilder, cr: Register):
        self.parent = parent
        self.cr = cr

    def dissolve(self) -> Register:
        return self.cr

    def _apply1(self, r: Register, kind: str) -> Register:
        for q in r.indices:  # 1-qubit objects broadcast
            self.parent.try_apply_with_condition(self.cr, Register((q,)), kind)
        return r

    def x(self, r): return self._apply1(r, "X")
    def not_(self, r): return self._apply1(r, "X")
    def y(self, r): return self._apply1(r, "Y")
    def z(self, r): return self._apply1(r, "Z")

    def cnot(self, cr: Register, r: Register):
        if cr.n() > 1:
            raise CircuitError("Clifford CNOT can only have a single control qubit.")
        for q in r.indices:
            self.parent.try_apply_with_condition(self.cr, Register((cr.indices[0], q)), "CNOT")
        return cr, r

    def swap(self, ra: Register, rb: Register):
        # default CliffordTBuilder::swap on the Conditioned builder: 3 conditioned CNOTs per pair
        if ra.n() != rb.n():
            raise CircuitError("Swap must be between registers of the same size.")
        for a, b in zip(ra.indices, rb.indices):
            qa, qb = Register((a,)), Register((b,))
            self.cnot(qa, qb)
            self.cnot(qb, qa)
            self.cnot(qa, qb)
        return ra, rb
