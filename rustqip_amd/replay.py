"""Flat circuit-replay format ("qipc 1"): the lowered pipeline of a RustQIP circuit as text.

SURVEY.md §8 row f2: until a Rust toolchain can compile the `qip-hip` shim crate (bindings/rust/qip-hip), a Rust
program dumps what `LocalBuilder::calculate_state_with_init` would have executed — the `MatrixOp`s its run loop
lowers every pipeline entry to (qip/src/builder.rs:436-498) and the measurement stages between them (:501-511) —
and this module (or the C++ twin rustqip_amd/host/qip_replay.hpp + tools/qip_replay) replays it on the GPU.
The Rust writer is bindings/rust/qip-hip/src/replay.rs.

One statement per line, tokens separated by blanks, `#` starts a comment:

    qipc 1                      header (version)
    n <qubits>
    init <basis index>          optional, default 0 (builder.rs:409-421)
    matrix <k> <i_1..i_k> <re im>*4^k                          MatrixOp::Matrix, row-major
    sparse <k> <i_1..i_k> (<nnz> (<col> <re> <im>)*nnz)*2^k     MatrixOp::SparseMatrix, rows in order
    swap <h> <a_1..a_h> <b_1..b_h>                             MatrixOp::Swap
    control <nc> <c_1..c_nc> <matrix|sparse|swap ...>           MatrixOp::Control (inner op follows on the line)
    measure <k> <i_1..i_k> <rand_u01>                          collapse measurement with the caller's uniform sample
    probs <k> <i_1..i_k>                                        stochastic measurement (probabilities only)

Numbers are decimal (shortest round-trip repr on both sides, so f64 values survive exactly)."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Sequence, Tuple, Union

import numpy as np

from .ops import CircuitError, MatrixOp, make_control_op, make_matrix_op, make_sparse_matrix_op, make_swap_op


@dataclass
class Measure:
    indices: List[int]
    rand_u01: float


@dataclass
class Probs:
    indices: List[int]


Item = Union[MatrixOp, Measure, Probs]


@dataclass
class Circuit:
    n: int
    init: int = 0
    items: List[Item] = field(default_factory=list)

    def ops(self) -> List[MatrixOp]:
        return [it for it in self.items if isinstance(it, MatrixOp)]


def _num(x: float) -> str:
    return repr(float(x))


def _op_tokens(op: MatrixOp) -> List[str]:
    if op.kind == "Matrix":
        k = len(op.indices)
        d = np.asarray(op.data, dtype=np.complex128).ravel()
        if d.size != 4**k:
            raise CircuitError(f"Matrix data has {d.size} entries versus expected 2^2*{k}")
        return ["matrix", str(k), *map(str, op.indices), *[t for z in d for t in (_num(z.real), _num(z.imag))]]
    if op.kind == "SparseMatrix":
        toks = ["sparse", str(len(op.indices)), *map(str, op.indices)]
        for row in op.rows:
            toks.append(str(len(row)))
            for col, v in row:
                toks += [str(col), _num(complex(v).real), _num(complex(v).imag)]
        return toks
    if op.kind == "Swap":
        return ["swap", str(op.half), *map(str, op.indices)]
    if op.kind == "Control":
        return ["control", str(op.n_controls), *map(str, op.indices[: op.n_controls]), *_op_tokens(op.inner)]
    raise CircuitError(f"unknown op kind {op.kind!r}")


def dumps(circuit: Circuit) -> str:
    lines = ["qipc 1", f"n {circuit.n}"]
    if circuit.init:
        lines.append(f"init {circuit.init}")
    for it in circuit.items:
        if isinstance(it, MatrixOp):
            lines.append(" ".join(_op_tokens(it)))
        elif isinstance(it, Measure):
            lines.append(" ".join(["measure", str(len(it.indices)), *map(str, it.indices), _num(it.rand_u01)]))
        elif isinstance(it, Probs):
            lines.append(" ".join(["probs", str(len(it.indices)), *map(str, it.indices)]))
        else:
            raise CircuitError(f"cannot serialise {it!r}")
    return "\n".join(lines) + "\n"


class _Tokens:
    def __init__(self, toks: Sequence[str], lineno: int):
        self.t, self.i, self.lineno = toks, 0, lineno

    def take(self) -> str:
        if self.i >= len(self.t):
            raise CircuitError(f"line {self.lineno}: unexpected end of statement")
        self.i += 1
        return self.t[self.i - 1]

    def uint(self) -> int:
        s = self.take()
        if not s.isdigit():
            raise CircuitError(f"line {self.lineno}: expected a non-negative integer, found {s!r}")
        return int(s)

    def num(self) -> float:
        s = self.take()
        try:
            return float(s)
        except ValueError:
            raise CircuitError(f"line {self.lineno}: expected a number, found {s!r}") from None

    def done(self) -> None:
        if self.i != len(self.t):
            raise CircuitError(f"line {self.lineno}: {len(self.t) - self.i} unexpected trailing token(s)")


def _parse_op(tk: _Tokens, word: str) -> MatrixOp:
    if word == "matrix":
        k = tk.uint()
        idx = [tk.uint() for _ in range(k)]
        dat = [complex(tk.num(), tk.num()) for _ in range(4**k)]
        return make_matrix_op(idx, dat)
    if word == "sparse":
        k = tk.uint()
        idx = [tk.uint() for _ in range(k)]
        rows = []
        for _ in range(1 << k):
            nnz = tk.uint()
            rows.append([(tk.uint(), complex(tk.num(), tk.num())) for _ in range(nnz)])
        return make_sparse_matrix_op(idx, rows)
    if word == "swap":
        h = tk.uint()
        a = [tk.uint() for _ in range(h)]
        b = [tk.uint() for _ in range(h)]
        return make_swap_op(a, b)
    if word == "control":
        nc = tk.uint()
        c = [tk.uint() for _ in range(nc)]
        return make_control_op(c, _parse_op(tk, tk.take()))
    raise CircuitError(f"line {tk.lineno}: unknown statement {word!r}")


def loads(text: str) -> Circuit:
    circ: Circuit | None = None
    seen_header = False
    for lineno, raw in enumerate(text.splitlines(), 1):
        toks = raw.split("#", 1)[0].split()
        if not toks:
            continue
        tk = _Tokens(toks, lineno)
        word = tk.take()
        if not seen_header:
            if word != "qipc" or tk.uint() != 1:
                raise CircuitError(f"line {lineno}: expected the header 'qipc 1'")
            seen_header = True
        elif word == "n":
            circ = Circuit(tk.uint())
        elif circ is None:
            raise CircuitError(f"line {lineno}: 'n <qubits>' must come before {word!r}")
        elif word == "init":
            circ.init = tk.uint()
            if circ.init >> circ.n:
                raise CircuitError(f"line {lineno}: init index {circ.init} does not fit {circ.n} qubits")
        elif word == "measure":
            k = tk.uint()
            circ.items.append(Measure([tk.uint() for _ in range(k)], tk.num()))
        elif word == "probs":
            k = tk.uint()
            circ.items.append(Probs([tk.uint() for _ in range(k)]))
        else:
            circ.items.append(_parse_op(tk, word))
        tk.done()
    if circ is None:
        raise CircuitError("no 'n <qubits>' statement")
    return circ


def dump(path: str, circuit: Circuit) -> None:
    with open(path, "w") as f:
        f.write(dumps(circuit))


def load(path: str) -> Circuit:
    with open(path) as f:
        return loads(f.read())


def from_builder(builder) -> Circuit:
    """The pipeline a HipBuilder recorded, lowered exactly as its run loop would (builder.rs:436-498)."""
    from .builder import lower_to_matrix_op

    circ = Circuit(builder.n())
    for entry in builder.pipeline:
        if entry.kind == "GlobalPhase":  # builder.rs:431-432: recorded, never applied
            continue
        if entry.kind == "Measurement":
            raise CircuitError("a collapse measurement needs the caller's uniform sample: append replay.Measure yourself")
        if entry.kind == "StochasticMeasurement":
            circ.items.append(Probs(list(entry.indices)))
        else:
            circ.items.append(lower_to_matrix_op(entry))
    return circ


def run(circuit: Circuit, dtype=np.complex128, tile: int = 1, device: int = 0):
    """Replay on the GPU.  Returns (state handle results): list of measurement results in statement order —
    (measured, prob) for `measure`, the probability vector for `probs` — and the final HipState (caller closes)."""
    from .state import HipState

    st = HipState(circuit.n, dtype, device)
    try:
        st.set_option("tile", tile)
        st.init_basis(circuit.init)
        results: List[Union[Tuple[int, float], np.ndarray]] = []
        batch: List[MatrixOp] = []
        for it in circuit.items:
            if isinstance(it, MatrixOp):
                batch.append(it)
                continue
            if batch:
                st.apply_ops(batch)
                batch = []
            if isinstance(it, Measure):
                results.append(st.measure(it.indices, rand_u01=it.rand_u01))
            else:
                results.append(st.measure_probs(it.indices))
        if batch:
            st.apply_ops(batch)
        return results, st
    except BaseException:
        st.close()
        raise
