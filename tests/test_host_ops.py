"""Host logic of the drop-in boundary — CPU only: constructors and their error behaviour
(qip/src/state_ops/matrix_ops.rs:12-122), the C ABI's validator / byte accounting, symbol
export, and that compute entry points fail loudly without a GPU."""
import os
import re
import subprocess

import numpy as np
import pytest

import rustqip_amd as q
from rustqip_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    hdr = open(os.path.join(ROOT, "include", header)).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return set(re.findall(r"\b(qip_hip_[a-z0-9_]+)\s*\(", hdr))


def test_library_exports_every_declared_symbol():
    names = _declared("qip_hip.h")          # the binding contract
    debug = _declared("qip_hip_debug.h")    # host-only test hooks: exported, not part of the contract
    assert 25 <= len(names) <= 66 and len(debug) == 8 and not (names & debug)
    for name in sorted(names | debug):
        assert hasattr(_ffi.lib, name), f"libqip_hip.so does not export {name}"
    assert set(_ffi.SIGNATURES) == names and set(_ffi.DEBUG_SIGNATURES) == debug
    assert _ffi.lib.qip_hip_abi_version() == 8
    # nothing else is exported, and the test hooks sit under their own version-script node
    out = subprocess.run(["nm", "-D", "--defined-only", "--with-symbol-versions", _ffi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {}
    for line in out.splitlines():
        parts = line.split()
        if len(parts) >= 3 and parts[1] in "TW" and parts[2].startswith("qip_hip_"):
            sym, _, node = parts[2].partition("@@")
            exported[sym] = node
    assert set(exported) == names | debug
    assert all(exported[s] == "QIP_HIP_DEBUG" for s in debug) and all(exported[s] == "QIP_HIP" for s in names)
    # the Rust binding a maintainer would compile (bindings/rust/qip-hip/src/sys.rs; no rustc in this image) declares exactly
    # the contract: it cannot be type-checked here, so at least it cannot fall behind the header (67 of 68)
    sys_rs = open(os.path.join(ROOT, "bindings", "rust", "qip-hip", "src", "sys.rs")).read()
    assert set(re.findall(r"pub fn (qip_hip_[a-z0-9_]+)\s*\(", sys_rs)) == names


def test_make_matrix_op_errors():
    with pytest.raises(q.CircuitError, match="at least one op index"):
        q.make_matrix_op([], [1])
    with pytest.raises(q.CircuitError, match="entries versus expected"):
        q.make_matrix_op([0], [1, 0, 0])
    op = q.make_matrix_op([0, 1, 2], np.eye(8).ravel())
    assert op.num_indices() == 3 and [op.get_index(i) for i in range(3)] == [0, 1, 2]


def test_get_index_condition_and_swap():
    # qip/src/state_ops/matrix_ops.rs:276-304
    mop = q.MatrixOp.new_matrix([2, 3], [])
    op = q.make_control_op([0, 1], mop)
    assert op.num_indices() == 4 and [op.get_index(i) for i in range(4)] == [0, 1, 2, 3]
    sw = q.MatrixOp.new_swap([0, 1], [2, 3])
    assert sw.num_indices() == 4 and [sw.get_index(i) for i in range(4)] == [0, 1, 2, 3]


def test_make_sparse_endianness_b5():
    # qip/src/state_ops/matrix_ops.rs:346-377
    one = 1 + 0j
    expected = [[(1, one)], [(0, one)], [(3, one)], [(2, one)]]
    op1 = q.make_sparse_matrix_op([0, 1], expected, q.Representation.BigEndian)
    op2 = q.make_sparse_matrix_op([0, 1], [[(2, one)], [(3, one)], [(0, one)], [(1, one)]],
                                  q.Representation.LittleEndian)
    assert op1.rows == expected
    assert op2.rows == expected


def test_make_sparse_errors():
    with pytest.raises(q.CircuitError, match="at least one op index"):
        q.make_sparse_matrix_op([], [])
    with pytest.raises(q.CircuitError, match="rows versus expected"):
        q.make_sparse_matrix_op([0], [[(0, 1)]])
    with pytest.raises(q.CircuitError, match="must have data"):
        q.make_sparse_matrix_op([0], [[(0, 1)], []])


def test_make_swap_and_control_errors():
    with pytest.raises(q.CircuitError, match="at least 1 swap index"):
        q.make_swap_op([], [1])
    with pytest.raises(q.CircuitError, match="equal length"):
        q.make_swap_op([0, 1], [2])
    with pytest.raises(q.CircuitError, match="at least one control index"):
        q.make_control_op([], q.make_matrix_op([0], [0, 1, 1, 0]))
    inner = q.make_control_op([1], q.make_matrix_op([2], [0, 1, 1, 0]))
    outer = q.make_control_op([0], inner)  # collapse (:112-115)
    assert outer.n_controls == 2 and outer.indices == [0, 1, 2] and outer.inner.kind == "Matrix"


def test_flip_bits_doctest():
    assert q.flip_bits(3, 0b100) == 0b001 and q.flip_bits(3, 0b010) == 0b010 and q.flip_bits(4, 0b1010) == 0b0101


def test_c_validator_matches_constructors_and_panics():
    x = q.make_matrix_op([1], [0, 1, 1, 0])
    q.validate_op(3, x)
    with pytest.raises(q.CircuitError, match="out of range"):
        q.validate_op(1, x)  # the reference would underflow n-1-index and panic
    with pytest.raises(q.CircuitError, match="must have data"):
        q.validate_op(2, q.MatrixOp.new_sparse([0], [[(0, 1)], []]))
    with pytest.raises(q.CircuitError, match="out of range"):
        q.validate_op(2, q.MatrixOp.new_sparse([0], [[(0, 1)], [(2, 1)]]))
    with pytest.raises(q.CircuitError, match="equal length"):
        q.validate_op(4, q.MatrixOp("Swap", [0, 1, 2], half=1))
    with pytest.raises(q.CircuitError, match="differ"):
        bad = q.MatrixOp.new_control([0], [1, 2], q.MatrixOp.new_control([1], [2, 3], x))
        q.validate_op(4, bad)
    q.validate_op(4, q.MatrixOp.new_control([0], [1, 2], q.MatrixOp.new_control([1], [2], x)))


def test_algorithmic_bytes_table():
    """BASELINE.md §3 / SURVEY.md §8(d)."""
    n = 28
    full = 32.0 * 2**n
    H = [2**-0.5, 2**-0.5, 2**-0.5, -(2**-0.5)]
    assert q.algorithmic_bytes(n, q.make_matrix_op([5], H)) == full
    assert q.algorithmic_bytes(n, q.make_matrix_op([5], [0, 1, 1, 0])) == full
    assert q.algorithmic_bytes(n, q.make_matrix_op([5], [np.exp(-0.3j), 0, 0, np.exp(0.3j)])) == full
    assert q.algorithmic_bytes(n, q.make_swap_op([1], [9])) == full
    assert q.algorithmic_bytes(n, q.make_matrix_op([1, 9], np.ones(16))) == full
    cnot = q.make_control_op([3], q.make_matrix_op([7], [0, 1, 1, 0]))
    assert q.algorithmic_bytes(n, cnot) == full / 2
    for d in ([1, 0, 0, -1], [1, 0, 0, 1j]):  # Z, S: one non-unit diagonal entry of two
        assert q.algorithmic_bytes(n, q.make_matrix_op([4], d)) == full / 2
    cphase = q.make_control_op([3], q.make_matrix_op([7], [1, 0, 0, np.exp(0.1j)]))
    assert q.algorithmic_bytes(n, cphase) == full / 4
    assert q.algorithmic_bytes(n, q.make_matrix_op([4], [1, 0, 0, 1])) == 0  # identity: nothing can change
    assert q.algorithmic_bytes(n, q.make_matrix_op([4], H), _ffi.QIP_C32) == full / 2


def test_lowering_table_matrices():
    """builder.rs:436-498"""
    from rustqip_amd.builder import PipelineEntry, lower_to_matrix_op

    h = lower_to_matrix_op(PipelineEntry([0], "H")).data
    s = np.sqrt(0.5)
    assert np.array_equal(h, np.array([s, s, s, -s], dtype=np.complex128))
    assert np.signbit(h[3].imag)  # -nl = (-s, -0.0)
    t = lower_to_matrix_op(PipelineEntry([0], "T")).data
    assert t[3] == complex(np.cos(np.pi / 4), np.sin(np.pi / 4))
    rz = lower_to_matrix_op(PipelineEntry([0], "Rz", 0.5)).data
    assert rz[0] == complex(np.cos(-0.25), np.sin(-0.25)) and rz[3] == complex(np.cos(0.25), np.sin(0.25))
    cn = lower_to_matrix_op(PipelineEntry([2, 5], "CNOT"))
    assert cn.kind == "Control" and cn.n_controls == 1 and cn.indices == [2, 5]
    sw = lower_to_matrix_op(PipelineEntry([0, 1, 2, 3], "SWAP"))
    assert sw.kind == "Swap" and sw.half == 2


def test_builder_broadcast_and_init_index():
    b = q.HipBuilder()
    r = b.register(3)
    b.h(r)
    assert [(e.indices, e.kind) for e in b.pipeline] == [([0], "H"), ([1], "H"), ([2], "H")]  # builder.rs:382-387
    r2 = b.register(2)
    assert b.initial_index([(r, 0b101), (r2, 0b10)]) == (1 << 4) | (1 << 2) | (1 << 0)
    with pytest.raises(q.CircuitError, match="same size"):
        b.swap(r, r2)
    with pytest.raises(q.CircuitError, match="single control"):
        b.cnot(r, r2)


def test_no_gpu_fails_loudly():
    if q.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(q.QipHipError, match="no CPU fallback"):
        q.HipState(3)
    inp = np.zeros(2, dtype=np.complex128)
    out = np.zeros(2, dtype=np.complex128)
    with pytest.raises(q.QipHipError, match="no CPU fallback"):
        q.apply_op(1, q.make_matrix_op([0], [0, 1, 1, 0]), inp, out)
    # the real / integer element types of the slice-level calls: the same answer, for host and for device slices
    for dt in (np.float64, np.float32, np.int64, np.int32):
        with pytest.raises(q.QipHipError, match="no CPU fallback"):
            q.apply_op(1, q.MatrixOp.new_matrix([0], [0, 1, 1, 0]), np.ones(2, dtype=dt), np.zeros(2, dtype=dt))
        with pytest.raises(q.QipHipError, match="no CPU fallback"):
            q.apply_op_device(1, q.MatrixOp.new_matrix([0], [0, 1, 1, 0]), q.DeviceSlice(4096, 2, dt), q.DeviceSlice(8192, 2, dt))
    # ... and argument errors are reported before any device work (a state is Complex<P>; payloads of a real P are real)
    with pytest.raises(q.CircuitError, match="out of range"):
        q.apply_op(1, q.MatrixOp.new_matrix([3], [0, 1, 1, 0]), np.ones(2), np.zeros(2))
    with pytest.raises(q.CircuitError, match="imaginary"):
        q.apply_op(1, q.MatrixOp.new_matrix([0], [0, 1j, 1, 0]), np.ones(2), np.zeros(2))
    with pytest.raises(q.CircuitError, match="unsupported"):
        q.apply_op(1, q.MatrixOp.new_matrix([0], [0, 1, 1, 0]), np.ones(2, dtype=np.int16), np.zeros(2, dtype=np.int16))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "rustqip_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                hit = re.search(r"(import|from)\s+oracle|oracle[/.]|qip_oracle|libqip_oracle", text)
                assert hit is None, f"{f} references the CPU oracle: {hit.group(0)!r}"


def test_tile_schedule_invariants():
    """Host-side scheduling of the LDS-resident multi-gate sweeps (qip_hip_plan_tiles): every op lands in exactly
    one step; an op overtakes only ops it commutes with (on every shared qubit both gates only test it: controls,
    diagonal targets), in circuit-order mode only when one of the two is rounding-free; a segment claims at most 6
    free bits for its exchanging gates."""
    from rustqip_amd import circuits
    from rustqip_amd.ops import plan_tiles
    from rustqip_amd.ops import flatten

    n = 20
    rng = np.random.default_rng(0)
    ops = circuits.h_layer(n) + circuits.c2_random_circuit(n, 300, seed=3) + circuits.c3_qft(n)[:120]
    ops.insert(50, q.make_matrix_op([3, 9], np.eye(4).ravel() * (1 + 0j) + 0.1))      # dense k = 2: rides in a segment
    ops.insert(120, q.make_matrix_op([2, 7, 11], np.eye(8).ravel() * (1 + 0j) + 0.1))  # dense k = 3: a pass of its own three bits
    ops.insert(160, q.make_matrix_op([1, 5, 12, 17], np.eye(16).ravel() * (1 + 0j) + 0.1))  # dense k = 4: not tileable
    ops.insert(200, q.make_control_op(list(range(12)), q.make_matrix_op([15], [1, 0, 0, -1])))  # many controls: fine
    qubits = [set(flatten(o)[0]) | set(flatten(o)[2]) for o in ops]

    def exchange_bits(o):
        ctrl, inner, tgt = flatten(o)
        if inner.kind == "Swap":
            return {n - 1 - t for t in tgt}
        d = np.asarray(inner.data).reshape(2 ** len(tgt), -1)
        return set() if np.count_nonzero(d - np.diag(np.diagonal(d))) == 0 else {n - 1 - t for t in tgt}

    def exact(o):  # entries in {0, +-1, +-i}, one per row: the gate never rounds
        ctrl, inner, tgt = flatten(o)
        if inner.kind == "Swap":
            return True
        d = np.asarray(inner.data).reshape(2 ** len(tgt), -1)
        if len(tgt) != 1:
            return False
        ok = all((z == 0) or (abs(z) == 1 and (z.real == 0 or z.imag == 0)) for z in d.ravel())
        return ok and all(np.count_nonzero(row) <= 1 for row in d)

    def roles(o):  # qubit -> True when the op only TESTS the qubit (control / diagonal target), False when it exchanges
        ctrl, inner, tgt = flatten(o)
        if len(tgt) >= 4:  # not tileable: every qubit counts as exchanged
            return {t: False for t in list(ctrl) + list(tgt)}
        r = {c: True for c in ctrl}
        diag = inner.kind == "Matrix" and len(tgt) == 1 and not exchange_bits(o)
        r.update({t: diag for t in tgt})
        return r

    role = [roles(o) for o in ops]

    def commute(a, b):
        return all(role[a][x] and role[b][x] for x in role[a].keys() & role[b].keys())

    overtakes = {1: 0, 2: 0}
    shared = 0
    for mode in (1, 2):
        steps = plan_tiles(n, ops, mode)
        flat = [i for st in steps for i in st]
        assert sorted(flat) == list(range(len(ops)))
        pos = {i: k for k, i in enumerate(flat)}
        for a in range(len(ops)):
            for b in range(a + 1, len(ops)):
                if pos[b] < pos[a]:  # b overtook a
                    overtakes[mode] += 1
                    assert commute(a, b), (a, b)  # on every shared qubit both only test it
                    shared += bool(qubits[a] & qubits[b])
                    if mode == 1:  # only rounding-free commutations keep the result IEEE-equal
                        assert exact(ops[a]) or exact(ops[b]), (a, b)
    assert 0 < overtakes[1] < overtakes[2] and shared > 0
    # with the qubits relabelled (mode bits 2 + 3) the same ordering rules hold among the ops that are still ops (an
    # uncontrolled Swap becomes a label exchange and appears in no step); swaps are put in here to see some absorbed
    ops_sw = list(ops)
    for at, (a, b) in ((70, (2, 17)), (150, (0, 9)), (260, (5, 6))):
        ops_sw.insert(at, q.make_swap_op([a], [b]))
    role_sw = [roles(o) for o in ops_sw]
    for mode in (1 | 4 | 8, 2 | 4 | 8):
        steps = plan_tiles(n, ops_sw, mode)
        flat = [i for st in steps for i in st]
        assert len(set(flat)) == len(flat)
        gone = set(range(len(ops_sw))) - set(flat)
        assert gone and all(ops_sw[i].kind == "Swap" for i in gone)
        pos = {i: k for k, i in enumerate(flat)}
        for a in flat:
            for b in flat:
                if a < b and pos[b] < pos[a]:
                    assert all(role_sw[a][x] and role_sw[b][x] for x in role_sw[a].keys() & role_sw[b].keys()), (a, b)
                    if mode & 3 == 1:
                        assert exact(ops_sw[a]) or exact(ops_sw[b]), (a, b)
    for mode in (1, 2):
        steps = plan_tiles(n, ops, mode)
        for st in steps:
            assert all(st[k] < st[k + 1] for k in range(len(st) - 1))  # circuit order inside a step
            if len(st) > 1:
                free = set()
                for i in st:
                    free |= {p for p in exchange_bits(ops[i]) if p not in (0, 1, 2, 3, 4, 11 if n >= 12 else 5)}  # (the rows: qip_tile.h tile_p5)
                assert len(free) <= 6 and len(st) <= 256
                assert all(len(flatten(ops[i])[2]) <= 3 for i in st)  # 1-qubit gates, swaps, dense 2- and 3-qubit gates
        assert [160] in steps  # the dense 4-qubit gate is launched on its own
        assert not [50] in steps and not [120] in steps  # the dense 2- and 3-qubit gates share a sweep
    assert len(plan_tiles(n, ops, 2)) <= len(plan_tiles(n, ops, 1)) < len(ops) / 4
    print('steps', len(plan_tiles(n, ops, 1)), len(plan_tiles(n, ops, 2)), 'of', len(ops))
    with pytest.raises(q.CircuitError):
        plan_tiles(8, ops[:3], 1)  # n below the tile size


@pytest.mark.parametrize("dtype_name", ["c64", "c32"])
def test_tile_pass_layout_is_a_bijection_and_bank_conflict_free(dtype_name):
    """The LDS layout of the tile sweeps, checked against the banking model of MI355X_MICROARCH.md (section LDS)
    without a GPU: for every choice of three pass bits the lane-bit assignment is a bijection onto the other
    nine tile bits, and with the XOR-swizzled slot function the pass's reads (ds_read_b128: four 16-lane groups
    over 16 slots of 16 B; 8-byte amplitudes: ds_read_b64, two 32-lane groups over 32 slots) and writes
    (ds_write_b128: eight 8-lane groups over 8 slots; ds_write_b64: four 16-lane groups over 16 slots) are
    conflict-free unless the pass holds both bits of a pair (j, j + S) — then exactly 2-way."""
    import itertools

    from rustqip_amd import _ffi
    from rustqip_amd.ops import TILE_BITS, TILE_LANE_BITS, tile_lane_assignment

    if dtype_name == "c64":
        dtype, S = _ffi.QIP_C64, 4
        half = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
                [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
        read_groups = half + [[lane + 32 for lane in g] for g in half]
        write_groups = [list(range(8 * k, 8 * k + 8)) for k in range(8)]
        read_slots, write_slots = 16, 8
    else:
        dtype, S = _ffi.QIP_C32, 5
        read_groups = [list(range(32)), list(range(32, 64))]
        write_groups = [list(range(16 * k, 16 * k + 16)) for k in range(4)]
        read_slots, write_slots = 32, 16

    def slot(t):
        return t ^ ((t >> S) & ((1 << S) - 1))

    def worst(groups, nslots, pos, P):
        w = 1
        for wave in range(1 << (TILE_LANE_BITS - 6)):
            for i in range(8):
                ibits = sum(((i >> j) & 1) << P[j] for j in range(3))
                for g in groups:
                    hit = {}
                    for lane in g:
                        tid = wave * 64 + lane
                        t = ibits | sum(((tid >> k) & 1) << pos[k] for k in range(TILE_LANE_BITS))
                        hit.setdefault(slot(t) % nslots, set()).add(slot(t))
                    w = max(w, max(len(v) for v in hit.values()))
        return w

    for P in itertools.combinations(range(TILE_BITS), 3):
        pos = tile_lane_assignment(P, dtype)
        assert sorted(pos + list(P)) == list(range(TILE_BITS)), (P, pos)
        has_pair = any(b + S in P for b in P)
        r, w = worst(read_groups, read_slots, pos, P), worst(write_groups, write_slots, pos, P)
        if has_pair:
            assert r <= 2 and w <= 2, (P, r, w)
        else:
            assert (r, w) == (1, 1), (P, r, w)
    with pytest.raises(q.CircuitError):
        tile_lane_assignment((3, 3, 5), dtype)


def test_header_is_plain_c11(tmp_path):
    """include/qip_hip.h is the drop-in boundary: it must compile as C11 (no C++-isms) so cgo / bindgen / a C host can
    consume it, and a C program must be able to fill the descriptor and transport structs it declares."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "use_header.c"
    src.write_text("""
#include "qip_hip.h"
static int a2a(void* ctx, const void* send, void* recv, uint64_t chunk_bytes, void* stream) { (void)ctx; (void)send; (void)recv; (void)chunk_bytes; (void)stream; return 0; }
static int ars(void* ctx, double* v, uint64_t count) { (void)ctx; (void)v; (void)count; return 0; }
int main(void) {
  static const uint64_t idx[2] = {0, 1};
  static const qip_c64 x[4] = {{0, 0}, {1, 0}, {1, 0}, {0, 0}};
  qip_op inner = {QIP_OP_MATRIX, 1, idx + 1, 0, x, 0, 0, 0, 0};
  qip_op cnot = {QIP_OP_CONTROL, 2, idx, 1, 0, 0, 0, 0, &inner};
  qip_hip_transport t = {0, a2a, ars};
  qip_hip_dist_stats st = {0, 0, 0, 0.0, 0.0, 0, -1, 0, 0, 0, 0, 0, 0, 0};
  qip_hip_jit_counters jc = {0, 0, 0, 0, 0, 0, 0.0, 0.0, 0, 0, 0, 0};
  qip_hip_all_to_all_slice_fn slice = 0;
  (void)jc; (void)slice;
  char id[QIP_HIP_UNIQUE_ID_BYTES];
  (void)id; (void)st; (void)t;
  return qip_hip_validate_op(2, &cnot) == QIP_OK && qip_hip_abi_version() >= 1 ? 0 : 1;
}
""")
    exe = tmp_path / "use_header"
    lib_dir = os.path.join(root, "rustqip_amd", "lib")
    subprocess.run(["gcc", "-std=c11", "-pedantic-errors", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(root, "include"),
                    str(src), "-o", str(exe), "-L", lib_dir, "-lqip_hip", "-Wl,-rpath," + lib_dir], check=True)
    env = dict(os.environ, LD_LIBRARY_PATH=lib_dir + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    assert subprocess.run([str(exe)], env=env).returncode == 0  # validate_op is host code: runs without a GPU


PRODUCT_GLOBAL_OPTIONS = ("force_generic", "jit_cache_cap", "jit_disk_cap_mb", "jit_disk_cache", "jit_procs", "tile_sched", "single_via_tile",
                          "dist_fold_pack", "dist_plan_cost", "collective_timeout_s")
TUNING_ONLY_OPTIONS = ("line_bits", "tile_pad_from", "soft_measure_one_pass", "tile_wide_dense3_inline", "tile_wide_pin", "sparse_tile",
                       "debug_slice_sweeps", "tile_diag_runs", "jit_threads", "tile_row_split_f32", "tile_row_split", "tile_wave_rule", "tile_remap",
                       "k4_direct", "single_via_tile_f32")


def test_option_surface_without_a_gpu():
    """r6: the product build knows the options include/qip_hip.h documents and nothing else — the measured alternatives of
    earlier rounds exist only in a -DQIP_HIP_TUNING build (conftest.has_tuning_options)."""
    names = [_ffi.lib.qip_hip_kernel_class_name(i).decode() for i in range(_ffi.lib.qip_hip_kernel_class_count())]
    assert names[-2:] == ["tile_sweep_parts", "k_dense_small"] and names.index("k_tile_passes") == 8  # (existing indices unchanged)
    for key, good, bad in (("jit_procs", 2, 65), ("jit_disk_cache", 0, None), ("collective_timeout_s", 30, -1), ("jit_disk_cap_mb", 64, None)):
        q.set_global_option(key, good)
        if bad is not None:
            with pytest.raises(q.CircuitError):
                q.set_global_option(key, bad)
    q.set_global_option("jit_procs", 0)
    q.set_global_option("jit_disk_cache", 1)
    q.set_global_option("collective_timeout_s", 120)
    q.set_global_option("jit_disk_cap_mb", -1)
    from conftest import has_tuning_options

    if not has_tuning_options():
        for key in TUNING_ONLY_OPTIONS:
            with pytest.raises(q.CircuitError, match="unknown global option"):
                q.set_global_option(key, 0)
    # every option the header documents is accepted, every accepted one is documented
    hdr = open(os.path.join(ROOT, "include", "qip_hip.h")).read()
    for key in PRODUCT_GLOBAL_OPTIONS:
        assert '"%s"' % key in hdr, key
    assert len(set(re.findall(r'^ \*   "([a-z0-9_]+)"', hdr, flags=re.M))) <= 25
    c = _ffi.jit_counters()
    assert c["procs"] >= 1 and c["disk_cache"] in (0, 1)
