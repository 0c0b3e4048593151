//! `HipBuilder`: circuits are built with the reference's own `LocalBuilder<f64>`; only the state calculation
//! (`LocalBuilder::calculate_state_with_init`, `qip/src/builder.rs:400-519`) runs on the GPU.
//!
//! Usage: build the circuit on `hb.local()` exactly as with a `LocalBuilder` (every trait of the
//! `CircuitBuilder` family is available there: `builder.rs:325,522,531,576,591,599,620,638,663,828,844,863,969`),
//! then call `hb.calculate_state_with_init(..)` instead of the local one.  The pipeline is read back through
//! `Subcircuitable::make_subcircuit` (`builder.rs:831-833`), lowered with the table of `builder.rs:436-498`
//! (`lower`, below) and replayed through the C ABI.
use crate::state::{HipError, HipState};
use num_complex::Complex;
use num_traits::ToPrimitive;
use qip::builder::{
    BuilderCircuitObject, BuilderCircuitObjectType, LocalBuilder, MeasurementObject, Qudit, RotationObject,
    UnitaryMatrixObject,
};
use qip::builder_traits::{CircuitBuilder, QubitRegister, Subcircuitable};
use qip::errors::CircuitResult;
use qip::state_ops::matrix_ops::{make_control_op, make_matrix_op, make_swap_op};
use qip_iterators::iterators::MatrixOp;

type C = Complex<f64>;

/// What `Measurements<P>` is for `LocalBuilder` (`builder.rs:304-323`); that struct has no public constructor,
/// so the GPU run returns its own.
#[derive(Debug, Clone)]
pub enum HipMeasurement {
    /// `MeasurementResults::Single(value, probability)`
    Single(usize, f64),
    /// `MeasurementResults::Stochastic(probabilities)`
    Stochastic(Vec<f64>),
}
#[derive(Debug, Clone, Default)]
pub struct HipMeasurements {
    pub results: Vec<HipMeasurement>,
}
impl HipMeasurements {
    /// `id` = position among the circuit's measurement stages, as in `MeasurementHandle` (`builder.rs:595-616`).
    pub fn get_measurement(&self, id: usize) -> (usize, f64) {
        match &self.results[id] {
            HipMeasurement::Single(v, p) => (*v, *p),
            HipMeasurement::Stochastic(_) => unreachable!("stage {id} is a stochastic measurement"),
        }
    }
    pub fn get_stochastic_measurement(&self, id: usize) -> &[f64] {
        match &self.results[id] {
            HipMeasurement::Stochastic(p) => p,
            HipMeasurement::Single(..) => unreachable!("stage {id} is a collapse measurement"),
        }
    }
}

pub struct HipBuilder {
    local: LocalBuilder<f64>,
    device: i32,
    /// option "tile" of the library: 1 = several gates per sweep, IEEE-equal to one sweep per gate (default)
    pub tile: i64,
}

impl Default for HipBuilder {
    fn default() -> Self {
        Self { local: LocalBuilder::default(), device: 0, tile: 1 }
    }
}

/// One pipeline entry as the matrix-level op the reference's run loop applies (the table at
/// `builder.rs:436-498`).  `None` for a global phase, which the run loop records but never applies (:431-432).
pub fn lower(indices: &[usize], obj: &UnitaryMatrixObject<f64>) -> CircuitResult<Option<MatrixOp<C>>> {
    let re = |x: f64| C::new(x, 0.0);
    let (zero, one, i) = (re(0.0), re(1.0), C::new(0.0, 1.0));
    let on_all = |m: [C; 4]| make_matrix_op(indices.to_vec(), m.to_vec());
    let diag = |d0: C, d1: C| on_all([d0, zero, zero, d1]);
    let op = match obj {
        UnitaryMatrixObject::GlobalPhase(_) => return Ok(None),
        UnitaryMatrixObject::X => on_all([zero, one, one, zero]),
        UnitaryMatrixObject::Y => on_all([zero, -i, i, zero]),
        UnitaryMatrixObject::Z => diag(one, -one),
        UnitaryMatrixObject::H => {
            let s = one * std::f64::consts::FRAC_1_SQRT_2; // the reference's `1 * FRAC_1_SQRT_2` (:448-450)
            on_all([s, s, s, -s])
        }
        UnitaryMatrixObject::S => diag(one, i),
        UnitaryMatrixObject::T => diag(one, C::from_polar(1.0, std::f64::consts::FRAC_PI_4)),
        UnitaryMatrixObject::Rz(rot) => {
            // a PiRational is converted WITHOUT the factor pi, like the reference does (:487-489)
            let theta = match rot {
                RotationObject::Floating(t) => *t,
                RotationObject::PiRational(r) => r.to_f64().expect("ratio fits f64"),
            };
            let half = theta * 0.5;
            diag(C::from_polar(1.0, -half), C::from_polar(1.0, half))
        }
        UnitaryMatrixObject::MAT(data) => make_matrix_op(indices.to_vec(), data.clone()),
        UnitaryMatrixObject::CNOT => {
            let not = make_matrix_op(indices[1..].to_vec(), vec![zero, one, one, zero])?;
            make_control_op(vec![indices[0]], not)
        }
        UnitaryMatrixObject::SWAP => {
            assert_eq!(indices.len() % 2, 0, "swap over an odd number of qubits");
            let (a, b) = indices.split_at(indices.len() / 2);
            make_swap_op(a.to_vec(), b.to_vec())
        }
    }?;
    Ok(Some(op))
}

/// Basis index of the initial state (`builder.rs:409-421`): bit `k` of a register's value belongs to the
/// register's `k`-th qubit `q = indices[k]`, which is bit `n-1-q` of the state index.
pub fn initial_index<'a, It>(n: usize, it: It) -> usize
where
    It: IntoIterator<Item = (&'a Qudit, usize)>,
{
    let mut index = 0usize;
    for (reg, value) in it {
        for (k, &q) in reg.indices().iter().enumerate() {
            index |= ((value >> k) & 1) << (n - 1 - q);
        }
    }
    index
}

impl HipBuilder {
    pub fn new(device: i32) -> Self {
        Self { device, ..Self::default() }
    }
    /// The circuit under construction: use it exactly like a `LocalBuilder<f64>`.
    pub fn local(&mut self) -> &mut LocalBuilder<f64> {
        &mut self.local
    }
    pub fn n(&self) -> usize {
        self.local.n()
    }

    /// `calculate_state_with_init` on the GPU.  Errors of the C ABI (no device, out of memory, a descriptor the
    /// library rejects) come back as `HipError`; the reference's own version is infallible because it `.unwrap()`s
    /// (`builder.rs:517`) — call `.expect(..)` for the same behaviour.
    pub fn calculate_state_with_init<'a, It>(&mut self, it: It) -> Result<(Vec<C>, HipMeasurements), HipError>
    where
        It: IntoIterator<Item = (&'a Qudit, usize)>,
    {
        let n = self.local.n();
        let mut st = HipState::new(n, self.device)?;
        st.set_option("tile", self.tile)?;
        st.init_basis(initial_index(n, it))?;
        let pipeline: Vec<(Vec<usize>, BuilderCircuitObject<f64>)> =
            self.local.make_subcircuit().expect("LocalBuilder::make_subcircuit is infallible");
        let mut measurements = HipMeasurements::default();
        let mut run: Vec<MatrixOp<C>> = Vec::new(); // gates since the last measurement: one apply_ops call
        for (indices, obj) in &pipeline {
            match obj.object() {
                BuilderCircuitObjectType::Unitary(u) => {
                    let lowered = lower(indices, u).map_err(|e| HipError { code: 1, message: format!("{e:?}") })?;
                    if let Some(op) = lowered {
                        run.push(op);
                    }
                }
                BuilderCircuitObjectType::Measurement(kind) => {
                    if !run.is_empty() {
                        st.apply_ops(&run)?;
                        run.clear();
                    }
                    match kind {
                        // builder.rs:502-506; the uniform sample stays on the Rust side (`rand`)
                        MeasurementObject::Measurement => {
                            let (m, p) = st.measure(indices, None, rand::random::<f64>())?;
                            measurements.results.push(HipMeasurement::Single(m, p));
                        }
                        // builder.rs:507-510
                        MeasurementObject::StochasticMeasurement => {
                            measurements.results.push(HipMeasurement::Stochastic(st.measure_probs(indices)?));
                        }
                    }
                }
            }
        }
        if !run.is_empty() {
            st.apply_ops(&run)?;
        }
        Ok((st.download()?, measurements))
    }

    pub fn calculate_state(&mut self) -> Result<(Vec<C>, HipMeasurements), HipError> {
        self.calculate_state_with_init(std::iter::empty())
    }
}
