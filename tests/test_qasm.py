"""OpenQASM 2.0 export/ingest — the five tests of qip/src/qasm.rs:239-338 against HipBuilder's host
bookkeeping (CPU only), plus the round trip."""
import math
import os
import tempfile
from fractions import Fraction

import pytest

import rustqip_amd as q
from rustqip_amd.builder import Register
from rustqip_amd.qasm import from_openqasm, to_openqasm, write_openqasm_file


def test_qasm_header_and_measure():
    b = q.HipBuilder()
    q0 = b.register(1)
    q0 = b.h(q0)
    b.measure(q0)
    qasm = to_openqasm(b)
    for needle in ("OPENQASM 2.0;", 'include "qelib1.inc";', "qreg q[1];", "creg c[1];", "h q[0];", "measure q[0] -> c[0];"):
        assert needle in qasm


def test_qasm_cnot_no_creg():
    b = q.HipBuilder()
    q0, q1 = b.register(1), b.register(1)
    b.cnot(q0, q1)
    qasm = to_openqasm(b)
    assert "qreg q[2];" in qasm and "cx q[0],q[1];" in qasm and "creg c[" not in qasm


def test_qasm_rz_pi_rational():
    b = q.HipBuilder()
    b.register(1)
    q1 = b.register(1)
    b.rz_pi_by(q1, 4)
    qasm = to_openqasm(b)
    assert "qreg q[2];" in qasm and "rz(1*pi/4) q[1];" in qasm
    # as written in the reference the SIMULATED angle is 0.25 rad, not pi/4
    op = q.lower_to_matrix_op(b.pipeline[0])
    assert op.data[3] == complex(math.cos(0.125), math.sin(0.125))


def test_qasm_global_phase_comment():
    b = q.HipBuilder()
    q0 = b.register(1)
    b.apply_global_phase(q0, 0.3)
    qasm = to_openqasm(b)
    assert "// global phase 0.3 (ignored" in qasm and "qreg q[1];" in qasm and "creg c[" not in qasm
    assert q.lower_to_matrix_op(b.pipeline[0]) is None


def test_qasm_write_file_roundtrip():
    b = q.HipBuilder()
    q0, q1 = b.register(1), b.register(1)
    r = b.merge_two_registers(q0, q1)
    b.h(r)
    b.pipeline.append(type(b.pipeline[0])(list(r.indices), "CNOT"))
    b.measure(r)
    b.measure(r)
    path = os.path.join(tempfile.gettempdir(), "rustqip_amd_test_export.qasm")
    write_openqasm_file(b, path)
    text = open(path).read()
    os.remove(path)
    for needle in ("OPENQASM 2.0;", "qreg q[2];", "creg c[2];", "h q[0];", "cx q[0],q[1];", "measure q[0] -> c[0];",
                   "measure q[1] -> c[1];"):
        assert needle in text


def test_ingest_round_trip_and_angles():
    b = q.HipBuilder()
    r = b.register(4)
    b.h(r)
    b.x(Register((1,)))
    b.t(Register((2,)))
    b.cnot(Register((0,)), Register((3,)))
    b.swap_op(Register((1,)), Register((2,)))
    b.rz(Register((3,)), 0.123456789)
    b.rz_ratio(Register((0,)), Fraction(-3, 8))
    b.measure(Register((2,)))
    text = to_openqasm(b)
    b2 = from_openqasm(text)
    assert b2.n() == 4
    assert [(e.indices, e.kind) for e in b2.pipeline] == [(e.indices, e.kind) for e in b.pipeline]
    assert to_openqasm(b2).replace("rz(-1.178097245096)", "rz(-3*pi/8)") == text  # pi-rationals come back as radians
    rz = [e.param for e in b2.pipeline if e.kind == "Rz"]
    assert abs(rz[0] - 0.123456789) < 1e-12 and abs(rz[1] + 3 * math.pi / 8) < 1e-15
    with pytest.raises(q.CircuitError):
        from_openqasm("OPENQASM 2.0;\nqreg q[2];\nccx q[0],q[1],q[0];")
    with pytest.raises(q.CircuitError):
        from_openqasm("OPENQASM 2.0;\nqreg q[1];\nrz(__import__('os')) q[0];")


@pytest.mark.gpu
def test_ingested_circuit_runs_on_gpu():
    import numpy as np
    from oracle import qip_oracle as O

    text = "OPENQASM 2.0;\ninclude \"qelib1.inc\";\nqreg q[5];\ncreg c[1];\nh q[0];\ncx q[0],q[1];\ncx q[1],q[2];\nrz(1*pi/4) q[2];\n" \
           "swap q[0],q[4];\nt q[3];\ny q[3];\nmeasure q[4] -> c[0];\n"
    b = from_openqasm(text)
    state, meas = b.calculate_state_with_init([], forced_measurements=[1])
    pipe = [(e.indices, e.kind, e.param) for e in b.pipeline]
    want, res = O.run_pipeline(5, pipe, 0, forced_measurements=[1])
    assert np.max(np.abs(state - want)) < 1e-12
    assert meas.get_measurement(0)[0] == 1 and abs(meas.get_measurement(0)[1] - 0.5) < 1e-12
