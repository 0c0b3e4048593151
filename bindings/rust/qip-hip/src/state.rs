//! `HipState`: the device-resident `2^n` amplitude vector (`state` + `arena` of `builder.rs:406-407`).
use crate::op::{marshal, HipPrecision};
use crate::sys;
use num_complex::Complex;
use qip_iterators::iterators::MatrixOp;
use std::ffi::{CStr, CString};
use std::marker::PhantomData;

/// A non-zero status of the C ABI with the library's message (`qip_hip_last_error`).
#[derive(Debug, Clone)]
pub struct HipError {
    pub code: i32,
    pub message: String,
}
impl std::fmt::Display for HipError {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "qip_hip error {}: {}", self.code, self.message)
    }
}
impl std::error::Error for HipError {}

pub(crate) fn check(rc: i32) -> Result<(), HipError> {
    if rc == sys::QIP_OK {
        return Ok(());
    }
    let message = unsafe { CStr::from_ptr(sys::qip_hip_last_error()) }.to_string_lossy().into_owned();
    Err(HipError { code: rc, message })
}

pub struct HipState<P: HipPrecision = f64> {
    n: usize,
    h: *mut sys::qip_hip_state,
    _p: PhantomData<P>,
}

// One host thread drives a handle at a time (the reference's run loop is sequential, builder.rs:423-517);
// moving it to another thread is fine, sharing it is not.
unsafe impl<P: HipPrecision> Send for HipState<P> {}

impl<P: HipPrecision> HipState<P> {
    /// `2^n` `Complex<P>` amplitudes (+ as much scratch) in the HBM of `device`.  Fails when no gfx950 device
    /// is visible: there is no CPU fallback.
    pub fn new(n: usize, device: i32) -> Result<Self, HipError> {
        let mut h = std::ptr::null_mut();
        check(unsafe { sys::qip_hip_state_create(n as u32, P::DTYPE, device, &mut h) })?;
        Ok(Self { n, h, _p: PhantomData })
    }
    pub fn n(&self) -> usize {
        self.n
    }
    pub fn set_option(&mut self, key: &str, value: i64) -> Result<(), HipError> {
        let k = CString::new(key).expect("option key without NUL");
        check(unsafe { sys::qip_hip_state_set_option(self.h, k.as_ptr(), value) })
    }
    /// new[j] = old[src(j)], bit pi[d] of src(j) = bit d of j: any permutation of the index bits in one sweep
    /// (what a run of `Swap` ops composes to, qip-iterators/src/iterators/qubit_iterators.rs:208-218).
    pub fn permute_bits(&mut self, pi: &[u32]) -> Result<(), HipError> {
        assert_eq!(pi.len(), self.n, "the permutation must list all n index bits");
        check(unsafe { sys::qip_hip_state_permute_bits(self.h, pi.as_ptr()) })
    }
    /// |index> (builder.rs:409-421).
    pub fn init_basis(&mut self, index: usize) -> Result<(), HipError> {
        check(unsafe { sys::qip_hip_state_init_basis(self.h, index as u64) })
    }
    pub fn upload(&mut self, amps: &[Complex<P>], offset: usize) -> Result<(), HipError> {
        check(unsafe { sys::qip_hip_state_upload(self.h, amps.as_ptr() as *const _, offset as u64, amps.len() as u64) })
    }
    pub fn download(&self) -> Result<Vec<Complex<P>>, HipError> {
        let mut out = vec![Complex::new(P::zero(), P::zero()); 1usize << self.n];
        check(unsafe { sys::qip_hip_state_download(self.h, out.as_mut_ptr() as *mut _, 0, out.len() as u64) })?;
        Ok(out)
    }
    /// `apply_op_overwrite` + buffer swap of the reference run loop (builder.rs:499,514), in place on the device.
    pub fn apply_op(&mut self, op: &MatrixOp<Complex<P>>) -> Result<(), HipError> {
        let c = marshal(op);
        check(unsafe { sys::qip_hip_state_apply_op(self.h, c.as_ptr()) })
    }
    /// A run of gates in one call, so the library may schedule them (options "tile", "fuse").
    pub fn apply_ops(&mut self, ops: &[MatrixOp<Complex<P>>]) -> Result<(), HipError> {
        let keep: Vec<_> = ops.iter().map(marshal).collect();
        let flat: Vec<sys::qip_op> = keep.iter().map(|c| unsafe { std::ptr::read(c.as_ptr()) }).collect();
        let rc = unsafe { sys::qip_hip_state_apply_ops(self.h, flat.as_ptr(), flat.len() as u64) };
        std::mem::forget(flat); // bitwise copies of descriptors that `keep` owns
        check(rc)
    }
    /// `self <- other`, device to device (same n, precision and device).
    pub fn copy_from(&mut self, other: &mut HipState<P>) -> Result<(), HipError> {
        check(unsafe { sys::qip_hip_state_copy_from(self.h, other.h) })
    }
    /// `(max_i |self_i - other_i|, number of amplitudes that are not IEEE-equal)` over the whole vector, on the device:
    /// what the parity checks use to hold a state against a reference copy.
    pub fn max_abs_diff(&mut self, other: &mut HipState<P>) -> Result<(f64, u64), HipError> {
        let (mut worst, mut differ) = (0.0f64, 0u64);
        check(unsafe { sys::qip_hip_state_max_abs_diff(self.h, other.h, &mut worst, &mut differ) })?;
        Ok((worst, differ))
    }
    /// Amplitudes at an explicit list of indices (one gather kernel): a logical window of a sharded / relabelled state.
    pub fn download_indices(&mut self, indices: &[u64]) -> Result<Vec<Complex<P>>, HipError> {
        let mut out = vec![Complex::new(P::zero(), P::zero()); indices.len()];
        check(unsafe { sys::qip_hip_state_download_indices(self.h, indices.as_ptr(), indices.len() as u64, out.as_mut_ptr() as *mut _) })?;
        Ok(out)
    }
    pub fn norm_sqr(&self) -> Result<f64, HipError> {
        let mut v = 0.0;
        check(unsafe { sys::qip_hip_state_norm_sqr(self.h, &mut v) })?;
        Ok(v)
    }
    /// `measure_probs` (measurement_ops.rs:115-127).
    pub fn measure_probs(&self, indices: &[usize]) -> Result<Vec<f64>, HipError> {
        let idx: Vec<u64> = indices.iter().map(|&i| i as u64).collect();
        let mut out = vec![0.0; 1usize << idx.len()];
        check(unsafe { sys::qip_hip_state_measure_probs(self.h, idx.as_ptr(), idx.len() as u32, out.as_mut_ptr()) })?;
        Ok(out)
    }
    /// `measure` (measurement_ops.rs:190-214): sample with `rand_u01` (or force `measured`), collapse, renormalise.
    pub fn measure(&mut self, indices: &[usize], forced: Option<usize>, rand_u01: f64) -> Result<(usize, f64), HipError> {
        let idx: Vec<u64> = indices.iter().map(|&i| i as u64).collect();
        let (mut m, mut p) = (0u64, 0f64);
        let f = forced.map(|v| v as i64).unwrap_or(-1);
        check(unsafe { sys::qip_hip_state_measure(self.h, idx.as_ptr(), idx.len() as u32, f, rand_u01, &mut m, &mut p) })?;
        Ok((m as usize, p))
    }
    pub fn raw(&mut self) -> *mut sys::qip_hip_state {
        self.h
    }
}

impl<P: HipPrecision> Drop for HipState<P> {
    fn drop(&mut self) {
        unsafe { sys::qip_hip_state_destroy(self.h) };
    }
}
