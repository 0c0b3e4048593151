#!/usr/bin/env python
"""Writes tests/golden/reference_vectors.json: the golden vectors the REFERENCE's own tests hold for the hot path (SURVEY.md
Appendix B), as data — inputs and the outputs the reference asserts.  The reference is Rust and cannot run here, so nothing in
this file is produced by running it, and nothing is produced by this repository's oracle or kernels either: every expected value
is either the literal the reference's test compares with, or (B1 / B2) the result of the reference's own test helper
`ndarray_kron_helper` (qip-iterators/src/matrix_ops.rs:257-269: Kronecker products with 2 x 2 identities), restated with numpy.

    python tests/golden/make_vectors.py        # rewrites reference_vectors.json next to this file

Op descriptors: {"kind": "matrix" | "sparse" | "swap" | "control", "indices": [...], "data": [...] | "rows": [[[col, re, im], ...], ...]
| "a": [...], "b": [...] | "controls": [...], "inner": {...}} — the constructors of iterators/ops.rs:49-91."""
import json
import math
import os

import numpy as np


def kron_helper(before, mat, after):
    eye = np.eye(2)
    for _ in range(before):
        mat = np.kron(eye, mat)
    for _ in range(after):
        mat = np.kron(mat, eye)
    return mat


def mat_json(m):
    return [[float(v) for v in row] for row in np.asarray(m, dtype=float)]


cases = []
# B1: qip-iterators/src/matrix_ops.rs:271-348 — make_op_matrix(n, Matrix([q], data)) == kron(I.., data, ..I), exact (integers)
for name, line, data, q in (("test_ident", "271-282", [1, 0, 0, 1], 0), ("test_flip", "284-295", [0, 1, 1, 0], 0),
                            ("test_flip_mid", "297-308", [0, 1, 1, 0], 1), ("test_flip_end", "310-321", [0, 1, 1, 0], 2),
                            ("test_counting", "337-348", [1, 2, 3, 4], 0)):
    cases.append({"id": "B1." + name, "ref": "qip-iterators/src/matrix_ops.rs:" + line, "check": "op_matrix", "n": 3,
                  "op": {"kind": "matrix", "indices": [q], "data": data},
                  "matrix": mat_json(kron_helper(q, np.array(data, float).reshape(2, 2), 2 - q)), "exact": True})
data = [1, 0, 0, 0, 0, 0, 1, 0, 0, 1, 0, 0, 0, 0, 0, 1]
cases.append({"id": "B1.test_flip_mid_twobody", "ref": "qip-iterators/src/matrix_ops.rs:323-335", "check": "op_matrix", "n": 4,
              "op": {"kind": "matrix", "indices": [1, 2], "data": data},
              "matrix": mat_json(kron_helper(1, np.array(data, float).reshape(4, 4), 1)), "exact": True})
# B2: matrix_ops.rs:350-375 — a 4 x 4 matrix of 0..15 on indices [0, 1] is itself; on [1, 0] it is NOT
data = list(range(16))
cases.append({"id": "B2.test_counting_order", "ref": "qip-iterators/src/matrix_ops.rs:350-361", "check": "op_matrix", "n": 2,
              "op": {"kind": "matrix", "indices": [0, 1], "data": data}, "matrix": mat_json(np.array(data, float).reshape(4, 4)), "exact": True})
cases.append({"id": "B2.test_counting_order_flipped", "ref": "qip-iterators/src/matrix_ops.rs:363-375", "check": "op_matrix_differs", "n": 2,
              "op": {"kind": "matrix", "indices": [1, 0], "data": data}, "matrix": mat_json(np.array(data, float).reshape(4, 4))})
# B3: iterators/qubit_iterators.rs:289-379 — the (column, value) each row iterator yields; values are all 1
x = {"kind": "matrix", "indices": [0], "data": [0, 1, 1, 0]}
cases.append({"id": "B3.test_mat_iterator", "ref": "qip-iterators/src/iterators/qubit_iterators.rs:289-307", "check": "row_columns", "n": 1, "op": x, "columns": [[1], [0]]})
cases.append({"id": "B3.test_sparse_mat_iterator", "ref": "qip-iterators/src/iterators/qubit_iterators.rs:309-328", "check": "row_columns", "n": 1,
              "op": {"kind": "sparse", "indices": [0], "rows": [[[1, 1.0, 0.0]], [[0, 1.0, 0.0]]]}, "columns": [[1], [0]]})
cases.append({"id": "B3.test_swap_iterator", "ref": "qip-iterators/src/iterators/qubit_iterators.rs:330-352", "check": "row_columns", "n": 2,
              "op": {"kind": "swap", "a": [0], "b": [1]}, "columns": [[0], [2], [1], [3]]})
cases.append({"id": "B3.test_c_iterator", "ref": "qip-iterators/src/iterators/qubit_iterators.rs:354-379", "check": "row_columns", "n": 2,
              "op": {"kind": "control", "controls": [0], "inner": {"kind": "matrix", "indices": [1], "data": [0, 1, 1, 0]}}, "columns": [[0], [1], [3], [2]]})
# B4: qip/src/state_ops/matrix_ops.rs:306-344 — apply_op on Complex<f64> vectors
cases.append({"id": "B4.test_apply_identity", "ref": "qip/src/state_ops/matrix_ops.rs:306-314", "check": "apply_op", "n": 1,
              "op": {"kind": "matrix", "indices": [0], "data": [1, 0, 0, 1]}, "input": [1, 0], "output": [1, 0]})
cases.append({"id": "B4.test_apply_swap_mat", "ref": "qip/src/state_ops/matrix_ops.rs:316-325", "check": "apply_op", "n": 1,
              "op": {"kind": "matrix", "indices": [0], "data": [0, 1, 1, 0]}, "input": [1, 0], "output": [0, 1]})
cases.append({"id": "B4.test_apply_swap_mat_first", "ref": "qip/src/state_ops/matrix_ops.rs:327-336", "check": "apply_op", "n": 2,
              "op": {"kind": "matrix", "indices": [0], "data": [0, 1, 1, 0]}, "input": [1, 0, 0, 0], "output": [0, 0, 1, 0]})
cases.append({"id": "B4.test_apply_swap_mat_first.second_half", "ref": "qip/src/state_ops/matrix_ops.rs:338-343", "check": "apply_op", "n": 2,
              "op": {"kind": "matrix", "indices": [1], "data": [0, 1, 1, 0]}, "input": [1, 0, 0, 0], "output": [0, 1, 0, 0]})
# B6: qip/src/state_ops/measurement_ops.rs — doctests :24-43 (measure_prob), :136-152 (soft_measure), tests :290-335
cases.append({"id": "B6.measure_prob_doctest", "ref": "qip/src/state_ops/measurement_ops.rs:24-43", "check": "measure_prob", "n": 2, "state": [0, 0, 1, 0],
              "queries": [{"measured": 0, "indices": [0], "p": 0.0}, {"measured": 1, "indices": [0], "p": 1.0},
                          {"measured": 1, "indices": [0, 1], "p": 1.0}, {"measured": 2, "indices": [1, 0], "p": 1.0}]})
cases.append({"id": "B6.soft_measure_doctest", "ref": "qip/src/state_ops/measurement_ops.rs:136-152", "check": "soft_measure", "n": 2, "state": [0, 0, 1, 0],
              "samples": [1e-12, 0.3, 0.999],
              "queries": [{"indices": [0], "m": 1}, {"indices": [1], "m": 0}, {"indices": [0, 1], "m": 1}, {"indices": [1, 0], "m": 2}]})
h = math.sqrt(0.5)
cases.append({"id": "B6.test_measure_state", "ref": "qip/src/state_ops/measurement_ops.rs:290-326 (test_measure_state, test_measure_state2)", "check": "measure_state", "n": 2, "state": [0.5, 0.5, 0.5, 0.5],
              "indices": [0], "outcomes": [{"m": 0, "p": 0.5, "after": [h, h, 0, 0]}, {"m": 1, "p": 0.5, "after": [0, 0, h, h]}], "round_to": 1e-10})
cases.append({"id": "B6.test_measure_probs", "ref": "qip/src/state_ops/measurement_ops.rs:328-335", "check": "measure_probs", "n": 2, "state": [0.5, 0.5, 0.5, 0.5],
              "indices": [1], "probs": [0.5, 0.5]})
# B7: README.md:26-63 (CSWAP example; the amplitudes follow from the circuit: SURVEY.md §3.3) — 7 qubits, |ra> = 000, |rb> = 001
cases.append({"id": "B7.readme_cswap", "ref": "README.md:26-63", "check": "cswap_state", "n": 7, "initial_index": 4, "pipeline_entries": 191,
              "amplitudes": {"4": 0.5, "32": 0.5, "68": 0.5, "96": -0.5}, "tolerance": 1e-12})

out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_vectors.json")
with open(out, "w") as f:  # one case per line
    f.write('{"source": "Renmusxd/RustQIP (qip 1.5.0, qip-iterators): constants of its own tests, transcribed; see make_vectors.py",\n "cases": [\n')
    f.write(",\n".join("  " + json.dumps(c) for c in cases))
    f.write("\n ]}\n")
print(out, len(cases), "cases")
