"""tests/golden/reference_vectors.json — the golden vectors the reference's own tests hold for the hot path, as DATA (inputs and the
outputs the reference asserts; written by tests/golden/make_vectors.py from the constants of those tests, nothing computed by this
repository's oracle or kernels) — replayed against (a) the CPU oracle, which is how the oracle is pinned, and (b) the HIP path
through the C ABI on a GPU.  tests/test_oracle_golden.py holds the same vectors inline plus the checks that go beyond them."""
import json
import os

import numpy as np
import pytest

from rustqip_amd.ops import MatrixOp

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = json.load(open(os.path.join(HERE, "golden", "reference_vectors.json")))
CASES = FIXTURE["cases"]


ELEMENT_TYPES = (np.complex128, np.int32, np.int64, np.float64, np.float32, np.complex64)


def op_of(d):
    if d["kind"] == "matrix":
        return MatrixOp.new_matrix(d["indices"], d["data"])
    if d["kind"] == "sparse":
        return MatrixOp.new_sparse(d["indices"], [[(c, complex(re, im)) for c, re, im in row] for row in d["rows"]])
    if d["kind"] == "swap":
        return MatrixOp.new_swap(d["a"], d["b"])
    inner = op_of(d["inner"])
    return MatrixOp.new_control(d["controls"], inner.indices, inner)


class OracleBackend:
    name = "oracle"

    def __init__(self):
        from oracle import qip_oracle as O

        self.O = O

    def op_matrix(self, n, op, dtype=np.complex128):
        return self.O.make_op_matrix(n, op, dtype=dtype)

    def apply_op(self, n, op, x):
        out = np.zeros_like(x)
        self.O.apply_op(n, op, x, out)
        return out

    def measure_prob(self, n, m, idx, x):
        return self.O.measure_prob(n, m, idx, x)

    def soft_measure(self, n, idx, x, r):
        return self.O.soft_measure(n, idx, x, r)

    def measure_probs(self, n, idx, x):
        return np.asarray(self.O.measure_probs(n, idx, x))

    def measure_state(self, n, idx, m, p, x):
        out = x.copy()
        assert self.O.measure_state(n, idx, (m, p), x, out)
        return out

    def run_builder(self, b, n, init_regs):
        unitary = [(e.indices, e.kind, e.param) for e in b.pipeline]
        return self.O.run_pipeline(n, unitary, b.initial_index(init_regs))[0]


class HipBackend:
    name = "hip"

    def __init__(self):
        import rustqip_amd as q

        self.q = q

    def op_matrix(self, n, op, dtype=np.complex128):
        return self.q.make_op_matrix(n, op, dtype=dtype)  # (apply_op's host-pointer twin on basis vectors, whatever the element type)

    def apply_op(self, n, op, x):
        out = np.zeros_like(x)
        self.q.apply_op(n, op, x, out)  # (the host-pointer twin of apply_op, through the C ABI)
        return out

    def _state(self, n, x):
        st = self.q.HipState(n)
        st.upload(np.asarray(x, dtype=np.complex128))
        return st

    def measure_prob(self, n, m, idx, x):
        with self._state(n, x) as st:
            return st.measure_prob(m, idx)

    def soft_measure(self, n, idx, x, r):
        with self._state(n, x) as st:
            return st.soft_measure(idx, r)

    def measure_probs(self, n, idx, x):
        with self._state(n, x) as st:
            return np.asarray(st.measure_probs(idx))

    def measure_state(self, n, idx, m, p, x):
        with self._state(n, x) as st:
            st.measure_state(idx, m, p)
            return st.download()

    def run_builder(self, b, n, init_regs):
        return b.calculate_state_with_init(init_regs)[0]  # HipBuilder's run loop: lowering table + apply_ops on the device


def replay(case, B):
    n = case["n"]
    kind = case["check"]
    if kind in ("op_matrix", "op_matrix_differs"):
        # qip-iterators' own tests: the reference runs them with P = i32 (integer literals); apply_op is generic over P
        # (matrix_ops.rs:98-107), so the same vectors are replayed in that type and in every other element type of the ABI
        for dtype in ELEMENT_TYPES:
            got = B.op_matrix(n, op_of(case["op"]), dtype)
            want = np.array(case["matrix"], dtype=np.complex128).real.astype(dtype) if dtype != np.complex128 else np.array(case["matrix"], dtype=dtype)
            assert got.dtype == dtype and np.array_equal(got, want) == (kind == "op_matrix"), (case["id"], dtype)
    elif kind == "row_columns":
        for dtype in ELEMENT_TYPES:
            m = B.op_matrix(n, op_of(case["op"]), dtype)
            assert [list(np.nonzero(m[r])[0]) for r in range(1 << n)] == case["columns"] and np.all(m[m != 0] == 1), (case["id"], dtype)
    elif kind == "apply_op":
        got = B.apply_op(n, op_of(case["op"]), np.array(case["input"], dtype=np.complex128))
        assert np.array_equal(got, np.array(case["output"], dtype=np.complex128)), case["id"]
    elif kind == "measure_prob":
        x = np.array(case["state"], dtype=np.complex128)
        for qy in case["queries"]:
            assert B.measure_prob(n, qy["measured"], qy["indices"], x) == qy["p"], (case["id"], qy)
    elif kind == "soft_measure":
        x = np.array(case["state"], dtype=np.complex128)
        for r in case["samples"]:
            for qy in case["queries"]:
                assert B.soft_measure(n, qy["indices"], x, r) == qy["m"], (case["id"], qy, r)
    elif kind == "measure_probs":
        assert list(B.measure_probs(n, case["indices"], np.array(case["state"], dtype=np.complex128))) == case["probs"], case["id"]
    elif kind == "measure_state":
        x = np.array(case["state"], dtype=np.complex128)
        for oc in case["outcomes"]:
            p = B.measure_prob(n, oc["m"], case["indices"], x)
            assert abs(p - oc["p"]) < np.finfo(float).eps, case["id"]
            got = B.measure_state(n, case["indices"], oc["m"], p, x)
            rt = case["round_to"]  # (the reference's approx_eq(.., 10): compare after rounding to 10 decimals)
            assert np.array_equal(np.round(got / rt) * rt, np.round(np.array(oc["after"], dtype=np.complex128) / rt) * rt), (case["id"], oc["m"])
    elif kind == "cswap_state":
        from rustqip_amd.builder import HipBuilder

        b = HipBuilder()  # README.md:26-63, recorded with the reference's builder calls
        qb = b.qubit()
        ra, rb = b.register(3), b.register(3)
        qb = b.h(qb)
        cb = b.condition_with(qb)
        ra, rb = cb.swap(ra, rb)
        qb = cb.dissolve()
        qb = b.h(qb)
        assert len(b.pipeline) == case["pipeline_entries"] and b.n() == n
        init_regs = [(ra, 0b000), (rb, 0b001)]
        assert b.initial_index(init_regs) == case["initial_index"]
        state = B.run_builder(b, n, init_regs)
        want = np.zeros(1 << n, dtype=np.complex128)
        for k, v in case["amplitudes"].items():
            want[int(k)] = v
        assert np.max(np.abs(state - want)) < case["tolerance"], case["id"]
    else:
        raise AssertionError("unknown check " + kind)


def test_fixture_file_is_what_the_script_writes(tmp_path):
    """the committed JSON is exactly the generator's output (no hand edits), and every case names where in the reference it comes from"""
    import subprocess
    import sys

    assert len(CASES) >= 21 and all(c["ref"] and c["id"] for c in CASES)
    src = open(os.path.join(HERE, "golden", "make_vectors.py")).read().replace('os.path.dirname(os.path.abspath(__file__))', repr(str(tmp_path)))
    script = tmp_path / "make_vectors.py"
    script.write_text(src)
    subprocess.run([sys.executable, str(script)], check=True, capture_output=True)
    assert json.load(open(tmp_path / "reference_vectors.json")) == FIXTURE


@pytest.mark.parametrize("case", CASES, ids=[c["id"] for c in CASES])
def test_oracle_against_the_reference_vectors(case):
    replay(case, OracleBackend())


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=[c["id"] for c in CASES])
def test_hip_path_against_the_reference_vectors(case):
    replay(case, HipBackend())
