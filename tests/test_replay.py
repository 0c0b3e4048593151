"""The flat circuit-replay format "qipc 1" (SURVEY.md §8 row f2): Python writer/reader, C++ reader + CLI.
Host logic only — no amplitudes are touched here; the GPU replay is in test_gpu_b_boundary.py."""
import cmath
import os
import subprocess

import numpy as np
import pytest

import rustqip_amd as q
from rustqip_amd import circuits, replay

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sample_circuit():
    ops = [
        q.make_matrix_op([0], circuits.H),
        q.make_control_op([1, 3], q.make_matrix_op([2], [0, 1, 1, 0])),
        q.make_swap_op([0, 1], [3, 4]),
        q.make_sparse_matrix_op([1, 2], [[(0, 1)], [(2, 0.5 + 0.25j), (1, 1j)], [(3, cmath.rect(1, 0.3))], [(1, -1)]]),
        q.make_control_op([4], q.make_swap_op([0], [1])),
        q.make_matrix_op([2, 4], np.arange(16) * (1 + 0.1j)),
        q.make_control_op([0], q.make_control_op([1], q.make_matrix_op([2], [1, 0, 0, 1j]))),  # collapses to 2 controls
    ]
    return replay.Circuit(5, 3, ops + [replay.Probs([0, 2]), replay.Measure([1], 0.375), ops[0]])


def same_op(a, b):
    if a.kind != b.kind or a.indices != b.indices or a.n_controls != b.n_controls or a.half != b.half:
        return False
    if a.kind == "Matrix":
        return np.array_equal(np.asarray(a.data), np.asarray(b.data))
    if a.kind == "SparseMatrix":
        return a.rows == b.rows
    if a.kind == "Control":
        return same_op(a.inner, b.inner)
    return True


def test_round_trip_is_exact():
    c = sample_circuit()
    text = replay.dumps(c)
    back = replay.loads(text)
    assert (back.n, back.init, len(back.items)) == (5, 3, len(c.items))
    for x, y in zip(c.items, back.items):
        if isinstance(x, q.MatrixOp):
            assert same_op(x, y)  # every f64 survives the decimal text bit for bit
        else:
            assert x == y
    assert replay.dumps(back) == text
    assert "control 2 0 1 matrix 1 2" in text  # nested controls were collapsed by make_control_op


def test_comments_blank_lines_and_builder_export():
    c = replay.loads("# a Bell pair\nqipc 1\n\nn 2   # two qubits\nmatrix 1 0 0.5 0 0.5 0 0.5 0 -0.5 0\n"
                     "control 1 0 matrix 1 1 0 0 1 0 1 0 0 0\nprobs 2 0 1\n")
    assert c.n == 2 and len(c.ops()) == 2 and isinstance(c.items[-1], replay.Probs)
    b = q.HipBuilder()
    r = b.register(3)
    b.h(r)
    b.apply_global_phase(r, 0.3)  # recorded, never applied (builder.rs:431-432)
    b.measure_stochastic(r)
    circ = replay.from_builder(b)
    assert [type(i).__name__ for i in circ.items] == ["MatrixOp"] * 3 + ["Probs"] and circ.n == 3
    assert replay.loads(replay.dumps(circ)).n == 3


@pytest.mark.parametrize("text, needle", [
    ("n 3\n", "header"),
    ("qipc 2\nn 3\n", "header"),
    ("qipc 1\nmatrix 1 0 1 0 0 0 0 0 1 0\n", "must come before"),
    ("qipc 1\nn 3\nmatrix 1 0 1 0 0 0\n", "unexpected end"),
    ("qipc 1\nn 3\nmatrix 1 0 1 0 0 0 0 0 1 0 9\n", "trailing"),
    ("qipc 1\nn 3\nswap 1 0\n", "unexpected end"),
    ("qipc 1\nn 3\nfrobnicate 1\n", "unknown statement"),
    ("qipc 1\nn 3\nmatrix 1 x 1 0 0 0 0 0 1 0\n", "non-negative integer"),
    ("qipc 1\nn 3\ninit 8\n", "does not fit"),
    ("qipc 1\nn 2\nsparse 1 0 0 1 0 1 0\n", "must have data"),  # the reference constructor's own error
    ("qipc 1\n", "no 'n"),
])
def test_malformed_input_is_rejected_with_the_line(text, needle):
    with pytest.raises(q.CircuitError) as e:
        replay.loads(text)
    assert needle in str(e.value)


@pytest.fixture(scope="module")
def cli():
    subprocess.run(["make", "-C", os.path.join(ROOT, "tools"), "qip_replay"], check=True, capture_output=True)
    return os.path.join(ROOT, "tools", "qip_replay")


def test_cpp_reader_rejects_what_python_rejects(cli, tmp_path):
    for i, (text, needle) in enumerate([("qipc 1\nn 3\nmatrix 1 0 1 0 0 0\n", "line 3: unexpected end"),
                                        ("qipc 1\nn 3\nfrobnicate 1\n", "unknown statement"),
                                        ("qipc 1\nn 2\nswap 1 0 1 7\n", "trailing"),
                                        ("qipc 1\nn 2\ncontrol 0 matrix 1 0 1 0 0 0 0 0 1 0\n", "at least one control")]):
        p = tmp_path / f"bad{i}.qipc"
        p.write_text(text)
        r = subprocess.run([cli, str(p)], capture_output=True, text=True)
        assert r.returncode == 1 and needle in r.stderr, (text, r.stderr)
    r = subprocess.run([cli], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr


def test_cpp_reader_accepts_the_python_writer_and_needs_a_gpu(cli, tmp_path):
    """Without a device the CLI parses the whole file, then fails loudly at state creation: no CPU fallback."""
    if q.device_count() > 0:
        pytest.skip("a GPU is present: the replay itself is checked in test_gpu_b_boundary.py")
    p = tmp_path / "ok.qipc"
    replay.dump(str(p), sample_circuit())
    r = subprocess.run([cli, str(p)], capture_output=True, text=True)
    assert r.returncode == 1 and "line" not in r.stderr and "device" in r.stderr.lower(), r.stderr
