"""ctypes binding of the C ABI declared in include/qip_hip.h.

The shared library is the product: if it is missing or fails to load, importing this
module raises — there is no Python / CPU fallback for any compute entry point.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (QIP_HIP_LIB: another build of the same library — how tools/ A/B two kernel variants inside one GPU call)
LIB_PATH = os.environ.get("QIP_HIP_LIB") or os.path.join(_HERE, "lib", "libqip_hip.so")

QIP_C64, QIP_C32 = 0, 1
QIP_F64, QIP_F32, QIP_I64, QIP_I32 = 2, 3, 4, 5  # real / integer P of the slice-level calls (include/qip_hip.h, enum qip_dtype)
QIP_OP_MATRIX, QIP_OP_SPARSE, QIP_OP_SWAP, QIP_OP_CONTROL = 0, 1, 2, 3
QIP_OK, QIP_ERR_INVALID, QIP_ERR_DEVICE, QIP_ERR_NO_DEVICE, QIP_ERR_UNSUPPORTED = 0, 1, 2, 3, 4


class QipOp(C.Structure):
    """struct qip_op (include/qip_hip.h): flat mirror of MatrixOp<P> (ops.rs:11-20)."""


QipOp._fields_ = [
    ("kind", C.c_int32),
    ("n_indices", C.c_uint32),
    ("indices", C.POINTER(C.c_uint64)),
    ("n_controls", C.c_uint32),
    ("dense", C.c_void_p),
    ("sparse_rowptr", C.POINTER(C.c_uint64)),
    ("sparse_cols", C.POINTER(C.c_uint64)),
    ("sparse_vals", C.c_void_p),
    ("inner", C.POINTER(QipOp)),
]

QIP_HIP_UNIQUE_ID_BYTES = 128

# struct qip_hip_transport: the two callbacks a caller-supplied transport provides
A2A_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)
ARS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_uint64)
A2AS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p)  # qip_hip_all_to_all_slice_fn


class QipTransport(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("all_to_all", A2A_FN), ("all_reduce_sum", ARS_FN)]


class QipDistStats(C.Structure):
    _fields_ = [("remaps", C.c_uint64), ("pack_sweeps", C.c_uint64), ("bytes_sent", C.c_uint64),
                ("exchange_ms", C.c_double), ("pack_ms", C.c_double),
                ("rccl_ranks", C.c_int32), ("rccl_rank", C.c_int32), ("pieces_sent", C.c_uint64), ("piece_bytes", C.c_uint64),
                ("packs_via_permute", C.c_uint64), ("packs_folded", C.c_uint64),
                ("remaps_overlapped", C.c_uint64), ("remaps_overlapped_after", C.c_uint64), ("slices_overlapped", C.c_uint64)]


# name -> (restype, argtypes); every symbol include/qip_hip.h declares
_u32, _u64, _i64, _int, _dbl = C.c_uint32, C.c_uint64, C.c_int64, C.c_int, C.c_double
_vp, _cp = C.c_void_p, C.c_char_p
_opp = C.POINTER(QipOp)
_u64p = C.POINTER(C.c_uint64)
_dblp = C.POINTER(C.c_double)
_statep = C.c_void_p

class QipJitCounters(C.Structure):
    """struct qip_hip_jit_counters (include/qip_hip.h)"""
    _fields_ = [("kernels_resident_total", C.c_uint64), ("compiled", C.c_uint64), ("compiled_by_helpers", C.c_uint64),
                ("helper_processes", C.c_uint64), ("disk_hits", C.c_uint64), ("disk_stores", C.c_uint64),
                ("compile_ms", C.c_double), ("disk_load_ms", C.c_double), ("procs", C.c_int32), ("disk_cache", C.c_int32),
                ("background_segments", C.c_uint64), ("disk_trimmed", C.c_uint64)]


def jit_counters() -> dict:
    """Where the run-time-compiled tile segments of this process came from (qip_hip_jit_stats2)."""
    c = QipJitCounters()
    if lib.qip_hip_jit_stats2(C.byref(c)) != 0:
        raise RuntimeError(last_error())
    return {name: getattr(c, name) for name, _ in QipJitCounters._fields_}


SIGNATURES = {
    "qip_hip_last_error": (_cp, []),
    "qip_hip_device_count": (_int, []),
    "qip_hip_abi_version": (_int, []),
    "qip_hip_set_global_option": (_int, [_cp, _i64]),
    "qip_hip_validate_op": (_int, [_u32, _opp]),
    "qip_hip_op_algorithmic_bytes": (_int, [_int, _u32, _opp, _dblp]),
    "qip_hip_apply_op_host": (_int, [_int, _u32, _opp, _vp, _u64, _vp, _u64, _u64, _u64, _int]),
    "qip_hip_apply_op_row_host": (_int, [_int, _u32, _opp, _vp, _u64, _u64, _u64, _u64, _vp]),
    "qip_hip_apply_op_device": (_int, [_int, _int, _vp, _u32, _opp, _vp, _u64, _vp, _u64, _u64, _u64, _int]),
    "qip_hip_measure_probs_host": (_int, [_int, _u32, _u64p, _u32, _vp, _u64, _u64, _dblp]),
    "qip_hip_measure_prob_host": (_int, [_int, _u32, _u64, _u64p, _u32, _vp, _u64, _u64, _dblp]),
    "qip_hip_state_create": (_int, [_u32, _int, _int, C.POINTER(_statep)]),
    "qip_hip_state_wrap": (_int, [_u32, _int, _int, _vp, _vp, _vp, C.POINTER(_statep)]),
    "qip_hip_state_destroy": (_int, [_statep]),
    "qip_hip_state_init_basis": (_int, [_statep, _u64]),
    "qip_hip_state_upload": (_int, [_statep, _vp, _u64, _u64]),
    "qip_hip_state_download": (_int, [_statep, _vp, _u64, _u64]),
    "qip_hip_state_device_ptr": (_int, [_statep, C.POINTER(_vp)]),
    "qip_hip_state_scratch_ptr": (_int, [_statep, C.POINTER(_vp)]),
    "qip_hip_state_swap_buffers": (_int, [_statep]),
    "qip_hip_state_sync": (_int, [_statep]),
    "qip_hip_state_permute_bits": (_int, [_statep, C.POINTER(C.c_uint32)]),
    "qip_hip_state_apply_op": (_int, [_statep, _opp]),
    "qip_hip_state_apply_ops": (_int, [_statep, _opp, _u64]),
    "qip_hip_program_create": (_int, [_statep, _opp, _u64, C.POINTER(C.c_void_p)]),
    "qip_hip_program_run": (_int, [C.c_void_p]),
    "qip_hip_program_is_graph": (_int, [C.c_void_p]),
    "qip_hip_program_destroy": (_int, [C.c_void_p]),
    "qip_hip_plan_tiles": (_int, [_int, _u32, _opp, _u64, _int, C.POINTER(C.c_int64), _u64p]),
    "qip_hip_tile_bits": (_int, []),
    "qip_hip_jit_cache_info": (_int, [_u64p, _u64p, _u64p]),
    "qip_hip_jit_stats2": (_int, [C.POINTER(QipJitCounters)]),
    "qip_hip_jit_set_cache_dir": (_int, [_cp]),
    "qip_hip_jit_cache_dir": (_cp, []),
    "qip_hip_jit_compile_file": (_int, [_cp, _int, _cp]),
    "qip_hip_state_set_option": (_int, [_statep, _cp, _i64]),
    "qip_hip_kernel_class_count": (_int, []),
    "qip_hip_kernel_class_name": (_cp, [_int]),
    "qip_hip_state_profile_get": (_int, [_statep, _int, _u64p, _dblp, _dblp]),
    "qip_hip_state_profile_reset": (_int, [_statep]),
    "qip_hip_state_copy_from": (_int, [_statep, _statep]),
    "qip_hip_state_max_abs_diff": (_int, [_statep, _statep, _dblp, _u64p]),
    "qip_hip_state_download_indices": (_int, [_statep, _u64p, _u64, _vp]),
    "qip_hip_state_norm_sqr": (_int, [_statep, _dblp]),
    "qip_hip_state_measure_probs": (_int, [_statep, _u64p, _u32, _dblp]),
    "qip_hip_state_measure_prob": (_int, [_statep, _u64, _u64p, _u32, _dblp]),
    "qip_hip_state_soft_measure": (_int, [_statep, _u64p, _u32, _dbl, _u64p]),
    "qip_hip_state_measure": (_int, [_statep, _u64p, _u32, _i64, _dbl, _u64p, _dblp]),
    "qip_hip_state_measure_state": (_int, [_statep, _u64p, _u32, _u64, _dbl]),
    "qip_hip_dist_unique_id": (_int, [_vp]),
    "qip_hip_dist_create": (_int, [_u32, _int, _int, _int, _int, _vp, C.POINTER(QipTransport), C.POINTER(_vp)]),
    "qip_hip_dist_destroy": (_int, [_vp]),
    "qip_hip_dist_set_slice_transport": (_int, [_vp, A2AS_FN]),
    "qip_hip_dist_init_basis": (_int, [_vp, _u64]),
    "qip_hip_dist_apply_op": (_int, [_vp, _opp]),
    "qip_hip_dist_apply_ops": (_int, [_vp, _opp, _u64]),
    "qip_hip_dist_sync": (_int, [_vp]),
    "qip_hip_dist_set_option": (_int, [_vp, _cp, _i64]),
    "qip_hip_dist_norm_sqr": (_int, [_vp, _dblp]),
    "qip_hip_dist_measure_probs": (_int, [_vp, _u64p, _u32, _dblp]),
    "qip_hip_dist_measure": (_int, [_vp, _u64p, _u32, _i64, _dbl, _u64p, _dblp]),
    "qip_hip_dist_local_state": (_int, [_vp, C.POINTER(_vp)]),
    "qip_hip_dist_layout": (_int, [_vp, C.POINTER(C.c_uint32)]),
    "qip_hip_dist_rank_flip": (_int, [_vp, C.POINTER(C.c_uint32)]),
    "qip_hip_dist_soft_measure": (_int, [_vp, _u64p, _u32, _dbl, _u64p]),
    "qip_hip_dist_take_stats": (_int, [_vp, C.POINTER(QipDistStats)]),
}

# the host-only test hooks of include/qip_hip_debug.h (not part of the binding contract: bindings do not mirror them)
DEBUG_SIGNATURES = {
    "qip_hip_debug_permute_plan": (_cp, [_u32, C.POINTER(C.c_uint32), _u32, _u32]),
    "qip_hip_tile_lane_assignment": (_int, [_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]),
    "qip_hip_debug_tile_plan": (_cp, [_int, _u32, _opp, _u64, _int]),
    "qip_hip_debug_sparse_tile": (_cp, [_int, _u32, _opp]),
    "qip_hip_debug_tile_jit": (_int, [_int, _u32, _opp, _u64, _int, _u64p, _u64p, _u64p, C.POINTER(C.c_char_p)]),
    "qip_hip_dist_debug_plan": (_cp, [_u32, _int, _int, _int, _opp, _u64]),
    "qip_hip_dist_debug_overlap": (_cp, [_u32, _int, _int, _int, _opp, _u64, _int, _int]),
    "qip_hip_dist_debug_pieces": (_i64, [_int, _int, _u64, _u64, _u64, C.POINTER(C.c_int32), _u64p, _u64p]),
}


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). rustqip_amd has no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in list(SIGNATURES.items()) + list(DEBUG_SIGNATURES.items()):
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def last_error() -> str:
    return lib.qip_hip_last_error().decode("utf-8", "replace")
