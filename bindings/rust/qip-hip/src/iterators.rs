//! `qip_iterators::matrix_ops::{apply_op, apply_op_overwrite, apply_op_row}` on the GPU for ANY element type the library
//! is built for — the reference functions are generic over `P` (`qip-iterators/src/matrix_ops.rs:38-59,98-152`; its unit
//! tests run them on `i32`, its benches on `f64`).  Same argument order as the reference; host slices (the call uploads,
//! runs the HIP kernels, downloads) or device pointers ([`apply_op_device`]).  Results are bit-equal to the CPU fold for
//! real and integer `P` (one lane folds one row in the reference's order; integers wrap as in a release build).
//!
//! Status: **uncompiled**, like the rest of this crate (see `lib.rs`).
use crate::state::{check, HipError};
use crate::sys;
use num_complex::Complex;
use qip_iterators::iterators::MatrixOp;
use std::marker::PhantomData;
use std::os::raw::{c_int, c_void};

/// An element type of `enum qip_dtype` (`include/qip_hip.h`).  `Copy` + `#[repr(C)]`-compatible layout: payloads and
/// vectors are passed by pointer.
pub trait HipElement: Copy {
    const DTYPE: c_int;
}
impl HipElement for Complex<f64> {
    const DTYPE: c_int = sys::QIP_C64;
}
impl HipElement for Complex<f32> {
    const DTYPE: c_int = sys::QIP_C32;
}
impl HipElement for f64 {
    const DTYPE: c_int = sys::QIP_F64;
}
impl HipElement for f32 {
    const DTYPE: c_int = sys::QIP_F32;
}
impl HipElement for i64 {
    const DTYPE: c_int = sys::QIP_I64;
}
impl HipElement for i32 {
    const DTYPE: c_int = sys::QIP_I32;
}

/// `struct qip_op` tree of a `MatrixOp<E>`: owns the widened indices and the CSR image, borrows dense data from the op.
pub struct ElemOp<'a, E: HipElement> {
    raw: Box<sys::qip_op>,
    _idx: Vec<u64>,
    _rowptr: Vec<u64>,
    _cols: Vec<u64>,
    _vals: Vec<E>,
    _inner: Option<Box<ElemOp<'a, E>>>,
    _borrow: PhantomData<&'a MatrixOp<E>>,
}

impl<'a, E: HipElement> ElemOp<'a, E> {
    pub fn as_ptr(&self) -> *const sys::qip_op {
        &*self.raw
    }
}

/// As `op::marshal`, for any element type.
pub fn marshal<'a, E: HipElement>(op: &'a MatrixOp<E>) -> ElemOp<'a, E> {
    let idx: Vec<u64> = op.indices().iter().map(|&i| i as u64).collect();
    let mut raw = sys::qip_op {
        kind: sys::QIP_OP_MATRIX,
        n_indices: idx.len() as u32,
        indices: std::ptr::null(),
        n_controls: 0,
        dense: std::ptr::null(),
        sparse_rowptr: std::ptr::null(),
        sparse_cols: std::ptr::null(),
        sparse_vals: std::ptr::null(),
        inner: std::ptr::null(),
    };
    let (mut rowptr, mut cols, mut vals, mut inner) = (Vec::new(), Vec::new(), Vec::new(), None);
    match op {
        MatrixOp::Matrix(_, data) => raw.dense = data.as_ptr() as *const _,
        MatrixOp::SparseMatrix(_, rows) => {
            raw.kind = sys::QIP_OP_SPARSE;
            rowptr.push(0u64);
            for row in rows {
                for (col, v) in row {
                    cols.push(*col as u64);
                    vals.push(*v);
                }
                rowptr.push(cols.len() as u64);
            }
        }
        MatrixOp::Swap(_, _) => raw.kind = sys::QIP_OP_SWAP,
        MatrixOp::Control(nc, _, boxed) => {
            raw.kind = sys::QIP_OP_CONTROL;
            raw.n_controls = *nc as u32;
            inner = Some(Box::new(marshal(boxed)));
        }
    }
    raw.indices = idx.as_ptr();
    if raw.kind == sys::QIP_OP_SPARSE {
        raw.sparse_rowptr = rowptr.as_ptr();
        raw.sparse_cols = cols.as_ptr();
        raw.sparse_vals = vals.as_ptr() as *const _;
    }
    if let Some(i) = &inner {
        raw.inner = i.as_ptr();
    }
    ElemOp { raw: Box::new(raw), _idx: idx, _rowptr: rowptr, _cols: cols, _vals: vals, _inner: inner, _borrow: PhantomData }
}

fn apply_host<E: HipElement>(
    n: usize, op: &MatrixOp<E>, input: &[E], output: &mut [E], input_offset: usize, output_offset: usize, accumulate: c_int,
) -> Result<(), HipError> {
    let c = marshal(op);
    check(unsafe {
        sys::qip_hip_apply_op_host(
            E::DTYPE, n as u32, c.as_ptr(), input.as_ptr() as *const c_void, input.len() as u64,
            output.as_mut_ptr() as *mut c_void, output.len() as u64, input_offset as u64, output_offset as u64, accumulate,
        )
    })
}

/// `qip_iterators::matrix_ops::apply_op` (matrix_ops.rs:98-123): `output[r] += (op · input)[r]`, windows included.
pub fn apply_op<E: HipElement>(
    n: usize, op: &MatrixOp<E>, input: &[E], output: &mut [E], input_offset: usize, output_offset: usize,
) -> Result<(), HipError> {
    apply_host(n, op, input, output, input_offset, output_offset, 1)
}

/// `apply_op_overwrite` (matrix_ops.rs:127-152): `output[r] = (op · input)[r]`.
pub fn apply_op_overwrite<E: HipElement>(
    n: usize, op: &MatrixOp<E>, input: &[E], output: &mut [E], input_offset: usize, output_offset: usize,
) -> Result<(), HipError> {
    apply_host(n, op, input, output, input_offset, output_offset, 0)
}

/// `apply_op_row` (matrix_ops.rs:38-59): the value of row `output_offset + outputrow`.
pub fn apply_op_row<E: HipElement + Default>(
    n: usize, op: &MatrixOp<E>, input: &[E], outputrow: usize, input_offset: usize, output_offset: usize,
) -> Result<E, HipError> {
    let c = marshal(op);
    let mut value = E::default();
    check(unsafe {
        sys::qip_hip_apply_op_row_host(
            E::DTYPE, n as u32, c.as_ptr(), input.as_ptr() as *const c_void, input.len() as u64, outputrow as u64,
            input_offset as u64, output_offset as u64, &mut value as *mut E as *mut c_void,
        )
    })?;
    Ok(value)
}

/// The same on device slices: `d_in` / `d_out` are device pointers to `in_len` / `out_len` elements on `device`, `stream`
/// a `hipStream_t` (null = the null stream).  A dense op on <= 4 qubits or a `Swap` is one asynchronous launch.
///
/// # Safety
/// The pointers must be valid device allocations of the stated lengths that do not alias.
pub unsafe fn apply_op_device<E: HipElement>(
    n: usize, op: &MatrixOp<E>, d_in: *const E, in_len: usize, d_out: *mut E, out_len: usize, input_offset: usize,
    output_offset: usize, accumulate: bool, device: i32, stream: *mut c_void,
) -> Result<(), HipError> {
    let c = marshal(op);
    check(sys::qip_hip_apply_op_device(
        E::DTYPE, device, stream, n as u32, c.as_ptr(), d_in as *const c_void, in_len as u64, d_out as *mut c_void,
        out_len as u64, input_offset as u64, output_offset as u64, accumulate as c_int,
    ))
}
