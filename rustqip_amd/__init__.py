"""rustqip_amd — MI355X-native backend for RustQIP's gate-application hot path.

Public surface (mirrors the reference names for this path):
  ops      MatrixOp, make_matrix_op, make_sparse_matrix_op, make_swap_op, make_control_op, CircuitError
  state    apply_op, apply_op_overwrite, apply_op_row (host-pointer twins; any element type P: complex, real, integer),
           apply_op_device (the same on device slices), HipState (device-resident Complex<P> state), make_op_matrix
  builder  HipBuilder (LocalBuilder's run loop on the GPU)
  circuits workload generators of BASELINE.json's configs
Importing the package loads rustqip_amd/lib/libqip_hip.so and fails loudly if it is missing.
"""
from . import _ffi  # noqa: F401  (raises ImportError when the HIP library is absent)
from .ops import (CircuitError, MatrixOp, Representation, algorithmic_bytes, flip_bits, make_control_op,
                  make_matrix_op, make_sparse_matrix_op, make_swap_op, validate_op)
from .state import (HipState, QipHipError, apply_op, apply_op_device, apply_op_overwrite, apply_op_row, device_count,
                    make_op_matrix, set_global_option)
from .builder import Conditioned, HipBuilder, Measurements, Register, lower_to_matrix_op
from .state import DeviceSlice, HipProgram
from . import circuits  # noqa: F401  (workload generators: rustqip_amd.circuits)
from . import replay  # noqa: F401  (flat circuit-replay format)

__all__ = [
    "CircuitError", "MatrixOp", "Representation", "algorithmic_bytes", "flip_bits", "make_control_op",
    "make_matrix_op", "make_sparse_matrix_op", "make_swap_op", "validate_op", "HipState", "QipHipError",
    "apply_op", "apply_op_device", "apply_op_overwrite", "apply_op_row", "device_count", "make_op_matrix", "set_global_option", "Conditioned",
    "HipBuilder", "Measurements", "Register", "lower_to_matrix_op", "HipProgram", "DeviceSlice", "circuits", "replay",
]
