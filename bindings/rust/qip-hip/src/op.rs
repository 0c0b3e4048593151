//! `MatrixOp<Complex<P>>` -> `struct qip_op`, for `P = f64` (`QIP_C64`) and `P = f32` (`QIP_C32`).
//!
//! `Complex<P>` is `#[repr(C)] { re, im }` in `num-complex`, so dense data and sparse values are passed by
//! pointer without copying; only the `usize` indices (widened to `u64`) and the CSR image of
//! `Vec<Vec<(usize, P)>>` are materialised.  A `COp` owns those and borrows the rest from the `MatrixOp`.
use crate::sys;
use num_complex::Complex;
use qip::Precision;
use qip_iterators::iterators::MatrixOp;
use std::marker::PhantomData;
use std::os::raw::c_int;

/// The two precisions the library is built for (`qip/src/types.rs:6-13`): selects the `dtype` argument of the C ABI.
pub trait HipPrecision: Precision {
    const DTYPE: c_int;
    fn to_f64(self) -> f64;
    fn from_f64(v: f64) -> Self;
}
impl HipPrecision for f64 {
    const DTYPE: c_int = sys::QIP_C64;
    fn to_f64(self) -> f64 {
        self
    }
    fn from_f64(v: f64) -> Self {
        v
    }
}
impl HipPrecision for f32 {
    const DTYPE: c_int = sys::QIP_C32;
    fn to_f64(self) -> f64 {
        self as f64
    }
    fn from_f64(v: f64) -> Self {
        v as f32
    }
}

pub struct COp<'a, P: HipPrecision> {
    raw: Box<sys::qip_op>,
    _idx: Vec<u64>,
    _rowptr: Vec<u64>,
    _cols: Vec<u64>,
    _vals: Vec<Complex<P>>,
    _inner: Option<Box<COp<'a, P>>>,
    _borrow: PhantomData<&'a MatrixOp<Complex<P>>>,
}

impl<'a, P: HipPrecision> COp<'a, P> {
    pub fn as_ptr(&self) -> *const sys::qip_op {
        &*self.raw
    }
    pub fn raw(&self) -> &sys::qip_op {
        &self.raw
    }
}

/// The index list the reference's `MatrixOp::indices()` yields (ops.rs:39-46): controls first for `Control`,
/// A half then B half for `Swap`.
pub fn marshal<'a, P: HipPrecision>(op: &'a MatrixOp<Complex<P>>) -> COp<'a, P> {
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
        MatrixOp::Matrix(_, data) => {
            raw.kind = sys::QIP_OP_MATRIX;
            raw.dense = data.as_ptr() as *const _;
        }
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
    // pointers are taken after the vectors have reached their final size; moving a Vec does not move its heap
    // buffer, so they stay valid inside the returned COp
    raw.indices = idx.as_ptr();
    if raw.kind == sys::QIP_OP_SPARSE {
        raw.sparse_rowptr = rowptr.as_ptr();
        raw.sparse_cols = cols.as_ptr();
        raw.sparse_vals = vals.as_ptr() as *const _;
    }
    if let Some(i) = &inner {
        raw.inner = i.as_ptr();
    }
    COp { raw: Box::new(raw), _idx: idx, _rowptr: rowptr, _cols: cols, _vals: vals, _inner: inner, _borrow: PhantomData }
}
