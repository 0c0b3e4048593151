"""OpenQASM 2.0 export and ingest for the gate set the reference's exporter emits
(qip/src/qasm.rs:18-213: x y z h s t cx swap rz measure; comments for what OQ2 cannot say).

`to_openqasm` produces the same text as `ToOpenQasm for LocalBuilder` (header, one classical bit per
collapsed-measured qubit, per-index emission, `k*pi/m` for PiRational angles, 12-digit trimmed decimals).
`from_openqasm` is the inverse for that gate set: a circuit *input format* for the GPU run loop.
Host bookkeeping only.
"""
from __future__ import annotations

import math
import re
from fractions import Fraction
from typing import List

from .builder import HipBuilder, PipelineEntry, Register
from .ops import CircuitError


def _format_angle(theta) -> str:
    """format_angle (qasm.rs:191-213)"""
    if isinstance(theta, Fraction):
        numer, denom = theta.numerator, theta.denominator  # Fraction keeps the denominator positive (:217-224)
        return f"{numer}*pi" if denom == 1 else f"{numer}*pi/{denom}"
    return f"{float(theta):.12f}".rstrip("0").rstrip(".")


def to_openqasm(b: HipBuilder) -> str:
    """ToOpenQasm::to_openqasm (qasm.rs:28-95)"""
    measured = sorted({q for e in b.pipeline if e.kind == "Measurement" for q in e.indices})
    cmap = {q: c for c, q in enumerate(measured)}
    out: List[str] = ["OPENQASM 2.0;", 'include "qelib1.inc";', f"qreg q[{b.n()}];"]
    if measured:
        out.append(f"creg c[{len(measured)}];")
    for e in b.pipeline:
        idx, k = e.indices, e.kind
        if k in ("X", "Y", "Z", "H", "S", "T"):
            out += [f"{k.lower()} q[{q}];" for q in idx]
        elif k == "CNOT":
            out += [f"cx q[{idx[0]}],q[{t}];" for t in idx[1:]]
        elif k == "SWAP":
            if len(idx) == 2:
                out.append(f"swap q[{idx[0]}],q[{idx[1]}];")
            elif len(idx) >= 2 and len(idx) % 2 == 0:
                half = len(idx) // 2
                out += [f"swap q[{idx[i]}],q[{idx[i + half]}];" for i in range(half)]
            elif len(idx) > 1:
                out.append(f"// swap with odd arity {idx} not directly supported")
        elif k == "Rz":
            out += [f"rz({_format_angle(e.param)}) q[{q}];" for q in idx]
        elif k == "GlobalPhase":
            out.append(f"// global phase {_format_angle(e.param)} (ignored in OpenQASM 2.0)")
        elif k == "MAT":
            out.append(f"// generic unitary on {idx} (not emitted in OpenQASM 2.0)")
        elif k == "Measurement":
            out += [f"measure q[{q}] -> c[{cmap[q]}];" for q in idx if q in cmap]
        elif k == "StochasticMeasurement":
            out.append(f"// stochastic measurement over {idx} (not in OpenQASM 2.0)")
    return "\n".join(out) + "\n"


def write_openqasm_file(b: HipBuilder, path) -> None:
    with open(path, "w") as f:
        f.write(to_openqasm(b))


_TOKEN = re.compile(r"\s*(?:(\d+\.?\d*(?:[eE][+-]?\d+)?|\.\d+(?:[eE][+-]?\d+)?)|(pi)|([-+*/()]))")


def _parse_angle(text: str) -> float:
    """Arithmetic over numbers and `pi` with + - * / and parentheses (what OpenQASM 2 angle expressions of the
    exporter's gate set use).  OpenQASM semantics: `k*pi/m` means k*pi/m radians — the reference's *simulation*
    of PiRational drops the pi, but its exported text means what it says, and that is what
    an ingester must honour."""
    tokens = []
    pos = 0
    while pos < len(text):
        if text[pos:].strip() == "":
            break
        m = _TOKEN.match(text, pos)
        if not m:
            raise CircuitError(f"unsupported angle expression {text!r}")
        tokens.append(float(m.group(1)) if m.group(1) else math.pi if m.group(2) else m.group(3))
        pos = m.end()
    it = iter(tokens + [None])
    cur = [next(it)]

    def advance():
        cur[0] = next(it)

    def atom():
        t = cur[0]
        if isinstance(t, float):
            advance()
            return t
        if t == "(":
            advance()
            v = expr()
            if cur[0] != ")":
                raise CircuitError(f"unbalanced parentheses in {text!r}")
            advance()
            return v
        if t in ("-", "+"):
            advance()
            v = atom()
            return -v if t == "-" else v
        raise CircuitError(f"unsupported angle expression {text!r}")

    def term():
        v = atom()
        while cur[0] in ("*", "/"):
            op = cur[0]
            advance()
            w = atom()
            v = v * w if op == "*" else v / w
        return v

    def expr():
        v = term()
        while cur[0] in ("+", "-"):
            op = cur[0]
            advance()
            w = term()
            v = v + w if op == "+" else v - w
        return v

    value = expr()
    if cur[0] is not None:
        raise CircuitError(f"unsupported angle expression {text!r}")
    return value


def from_openqasm(text: str, dtype=None) -> HipBuilder:
    """Build a HipBuilder pipeline from OpenQASM 2.0 text limited to the exporter's gate set."""
    b = HipBuilder() if dtype is None else HipBuilder(dtype)
    qregs = {}
    for raw in text.splitlines():
        line = raw.split("//", 1)[0].strip()
        if not line:
            continue
        for stmt in filter(None, (s.strip() for s in line.split(";"))):
            if stmt.startswith("OPENQASM") or stmt.startswith("include") or stmt.startswith("creg") or stmt.startswith("barrier"):
                continue
            m = re.fullmatch(r"qreg\s+(\w+)\s*\[\s*(\d+)\s*\]", stmt)
            if m:
                qregs[m.group(1)] = b.register(int(m.group(2))).indices[0]
                continue
            m = re.fullmatch(r"measure\s+(\w+)\s*\[\s*(\d+)\s*\]\s*->\s*\w+\s*\[\s*\d+\s*\]", stmt)
            if m:
                b.pipeline.append(PipelineEntry([qregs[m.group(1)] + int(m.group(2))], "Measurement"))
                continue
            m = re.fullmatch(r"(\w+)\s*(?:\((.*)\))?\s+(.+)", stmt)
            if not m:
                raise CircuitError(f"cannot parse {stmt!r}")
            gate, arg, operands = m.group(1), m.group(2), m.group(3)
            qs = []
            for o in operands.split(","):
                mo = re.fullmatch(r"\s*(\w+)\s*\[\s*(\d+)\s*\]\s*", o)
                if not mo or mo.group(1) not in qregs:
                    raise CircuitError(f"unknown operand {o!r}")
                qs.append(qregs[mo.group(1)] + int(mo.group(2)))
            if gate in ("x", "y", "z", "h", "s", "t") and len(qs) == 1:
                b.pipeline.append(PipelineEntry(qs, gate.upper()))
            elif gate == "cx" and len(qs) == 2:
                b.pipeline.append(PipelineEntry(qs, "CNOT"))
            elif gate == "swap" and len(qs) == 2:
                b.pipeline.append(PipelineEntry(qs, "SWAP"))
            elif gate == "rz" and len(qs) == 1 and arg is not None:
                b.pipeline.append(PipelineEntry(qs, "Rz", _parse_angle(arg)))
            else:
                raise CircuitError(f"gate {gate!r} is outside the reference exporter's gate set")
    return b
