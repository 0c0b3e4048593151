//! `HipBuilder<P>`: a `CircuitBuilder` (and every extension trait `LocalBuilder<P>` implements) whose state calculation
//! runs on an MI355X.
//!
//! All circuit bookkeeping is the reference's own: the builder owns a `LocalBuilder<P>` and every trait method below
//! forwards to it (`qip/src/builder.rs:325,522,531,576,591,599,620,638,663,828,844,863,969`).  Only
//! `calculate_state_with_init` (`builder.rs:400-519`) is replaced: the pipeline is read back through
//! `Subcircuitable::make_subcircuit` (`builder.rs:831-833`), lowered with the table of `builder.rs:436-498`
//! ([`lower`]) and replayed through the C ABI (`include/qip_hip.h`).
//!
//! So code written against the trait family runs unchanged:
//! ```ignore
//! let mut b = HipBuilder::<f64>::default();          // was: LocalBuilder::<f64>::default()
//! let q = b.qubit();  let r = b.register(NonZeroUsize::new(3).unwrap());
//! let q = b.h(q);     let (q, r) = b.cnot(q, r)?;     // CliffordTBuilder, AdvancedCircuitBuilder, ...
//! let (r, handle) = b.measure(r);
//! let (state, measured) = b.calculate_state();        // Vec<Complex<f64>>, Measurements-shaped results
//! let (value, prob) = measured.get_measurement(handle);
//! ```
//! The one visible difference: `qip::builder::Measurements` and `MeasurementHandle` have private fields and no public
//! constructor (`builder.rs:304-323,595-616`), so a crate outside `qip` cannot produce them.  `StateCalculation` is
//! therefore `(Vec<Complex<P>>, HipMeasurements<P>)` with handle types of the same shape and the same two accessor
//! methods; with type inference the calling code above does not change.  (The two-line upstream patch that would let
//! this crate return the reference's own types is shown in `INTEGRATION.md` §4.)
use crate::op::HipPrecision;
use crate::state::{HipError, HipState};
use num_complex::Complex;
use num_traits::ToPrimitive;
use qip::builder::{
    BuilderCircuitObject, BuilderCircuitObjectType, LocalBuilder, MeasurementObject, Qudit, RotationObject,
    UnitaryMatrixObject,
};
use qip::builder_traits::{
    AdvancedCircuitBuilder, CircuitBuilder, CliffordTBuilder, MeasurementBuilder, QubitRegister, RotationsBuilder,
    SplitResult, StochasticMeasurementBuilder, Subcircuitable, TemporaryRegisterBuilder, UnitaryBuilder,
};
use qip::conditioning::{Conditionable, ConditionableSubcircuit};
use qip::errors::{CircuitError, CircuitResult};
use qip::inverter::{Invertable, RecursiveCircuitBuilder};
use qip::state_ops::matrix_ops::{make_control_op, make_matrix_op, make_swap_op};
use qip_iterators::iterators::MatrixOp;
use std::num::NonZeroUsize;

/// `MeasurementResults<P>` (`builder.rs:292-301`).
#[derive(Debug, Clone)]
pub enum HipMeasurementResults<P: HipPrecision> {
    /// the measured value and the likelihood of that measurement
    Single(usize, P),
    /// the probability of each outcome, indexed by the outcome
    Stochastic(Vec<P>),
}

/// `MeasurementHandle` (`builder.rs:593-597`): position among the circuit's measurement stages.
#[derive(Debug, Clone, Copy)]
pub struct HipMeasurementHandle {
    id: usize,
}
/// `StochasticMeasurementHandle` (`builder.rs:614-618`).
#[derive(Debug, Clone, Copy)]
pub struct HipStochasticMeasurementHandle {
    id: usize,
}

/// `Measurements<P>` (`builder.rs:303-323`): same two accessors.
#[derive(Debug, Default)]
pub struct HipMeasurements<P: HipPrecision> {
    measurements: Vec<HipMeasurementResults<P>>,
}
impl<P: HipPrecision> HipMeasurements<P> {
    pub fn get_measurement(&self, handle: HipMeasurementHandle) -> (usize, P) {
        match &self.measurements[handle.id] {
            HipMeasurementResults::Single(val, prob) => (*val, *prob),
            HipMeasurementResults::Stochastic(_) => unreachable!(),
        }
    }
    pub fn get_stochastic_measurement(&self, handle: HipStochasticMeasurementHandle) -> &[P] {
        match &self.measurements[handle.id] {
            HipMeasurementResults::Single(_, _) => unreachable!(),
            HipMeasurementResults::Stochastic(probs) => probs.as_slice(),
        }
    }
}

#[derive(Debug)]
pub struct HipBuilder<P: HipPrecision = f64> {
    local: LocalBuilder<P>,
    /// measurement stages issued so far = id of the next handle (the inner builder counts the same way, `builder.rs:609-611`)
    measurements: usize,
    /// HIP device ordinal
    pub device: i32,
    /// option "tile" of the library: 1 = several gates per sweep, IEEE-equal to one sweep per gate (default)
    pub tile: i64,
    /// option "tile_relabel": 1 = the scheduler may relabel the qubits when that shortens the plan (bit-identical for `tile` = 1;
    /// uses the second buffer, as the reference's run loop does with `arena`, `builder.rs:406-407`) (default)
    pub tile_relabel: i64,
}

impl<P: HipPrecision> Default for HipBuilder<P> {
    fn default() -> Self {
        Self { local: LocalBuilder::default(), measurements: 0, device: 0, tile: 1, tile_relabel: 1 }
    }
}

/// One pipeline entry as the matrix-level op the reference's run loop applies (the table at `builder.rs:436-498`).
/// `None` for a global phase, which the run loop records but never applies (:431-432).
pub fn lower<P: HipPrecision>(indices: &[usize], obj: &UnitaryMatrixObject<P>) -> CircuitResult<Option<MatrixOp<Complex<P>>>> {
    let re = |x: f64| Complex::new(P::from_f64(x), P::zero());
    let (zero, one, i) = (re(0.0), re(1.0), Complex::new(P::zero(), P::one()));
    let on_all = |m: [Complex<P>; 4]| make_matrix_op(indices.to_vec(), m.to_vec());
    let diag = |d0: Complex<P>, d1: Complex<P>| on_all([d0, zero, zero, d1]);
    let op = match obj {
        UnitaryMatrixObject::GlobalPhase(_) => return Ok(None),
        UnitaryMatrixObject::X => on_all([zero, one, one, zero]),
        UnitaryMatrixObject::Y => on_all([zero, -i, i, zero]),
        UnitaryMatrixObject::Z => diag(one, -one),
        UnitaryMatrixObject::H => {
            let s = one * P::from_f64(std::f64::consts::FRAC_1_SQRT_2); // the reference's `1 * FRAC_1_SQRT_2` (:448-450)
            on_all([s, s, s, -s])
        }
        UnitaryMatrixObject::S => diag(one, i),
        UnitaryMatrixObject::T => diag(one, Complex::from_polar(P::one(), P::from_f64(std::f64::consts::FRAC_PI_4))),
        UnitaryMatrixObject::Rz(rot) => {
            // a PiRational is converted WITHOUT the factor pi, like the reference does (:480-489)
            let theta = match rot {
                RotationObject::Floating(t) => *t,
                RotationObject::PiRational(r) => P::from_f64(r.to_f64().expect("ratio fits f64")),
            };
            let half = theta * P::from_f64(0.5);
            diag(Complex::from_polar(P::one(), -half), Complex::from_polar(P::one(), half))
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

/// Basis index of the initial state (`builder.rs:409-421`): bit `k` of a register's value belongs to the register's
/// `k`-th qubit `q = indices[k]`, which is bit `n-1-q` of the state index.
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

impl<P: HipPrecision> HipBuilder<P> {
    pub fn new(device: i32) -> Self {
        Self { device, ..Self::default() }
    }
    /// The inner builder (read-only: e.g. `pipeline_depth`, `ToOpenQasm`).
    pub fn local(&self) -> &LocalBuilder<P> {
        &self.local
    }

    /// `calculate_state_with_init` with the C ABI's status as a `Result` instead of the panic the trait method turns
    /// it into (no device, out of memory, a descriptor the library rejects).
    pub fn try_calculate_state_with_init<'a, It>(&mut self, it: It) -> Result<(Vec<Complex<P>>, HipMeasurements<P>), HipError>
    where
        It: IntoIterator<Item = (&'a Qudit, usize)>,
    {
        let n = self.local.n();
        let mut st = HipState::<P>::new(n, self.device)?;
        st.set_option("tile", self.tile)?;
        st.set_option("tile_relabel", if self.tile >= 1 { self.tile_relabel } else { 0 })?;
        st.init_basis(initial_index(n, it))?;
        let pipeline = self.local.make_subcircuit().expect("LocalBuilder::make_subcircuit is infallible");
        let mut measurements = HipMeasurements { measurements: Vec::new() };
        let mut run: Vec<MatrixOp<Complex<P>>> = Vec::new(); // gates since the last measurement: one apply_ops call
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
                            let (m, p) = st.measure(indices, None, qip::rand::random::<f64>())?;
                            measurements.measurements.push(HipMeasurementResults::Single(m, P::from_f64(p)));
                        }
                        // builder.rs:507-510
                        MeasurementObject::StochasticMeasurement => {
                            let probs = st.measure_probs(indices)?.into_iter().map(P::from_f64).collect();
                            measurements.measurements.push(HipMeasurementResults::Stochastic(probs));
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
}

// ---- the trait family, by delegation (qip/src/builder_traits.rs:61-222 and the extension traits) --------------------

impl<P: HipPrecision> CircuitBuilder for HipBuilder<P> {
    type Register = Qudit;
    type CircuitObject = BuilderCircuitObject<P>;
    type StateCalculation = (Vec<Complex<P>>, HipMeasurements<P>);

    fn n(&self) -> usize {
        self.local.n()
    }
    fn register(&mut self, n: NonZeroUsize) -> Self::Register {
        self.local.register(n)
    }
    fn merge_two_registers(&mut self, r1: Self::Register, r2: Self::Register) -> Self::Register {
        self.local.merge_two_registers(r1, r2)
    }
    fn split_register_relative<It>(&mut self, r: Self::Register, indices: It) -> SplitResult<Self::Register>
    where
        It: IntoIterator<Item = usize>,
    {
        self.local.split_register_relative(r, indices)
    }
    fn apply_circuit_object(&mut self, r: Self::Register, c: Self::CircuitObject) -> CircuitResult<Self::Register> {
        self.local.apply_circuit_object(r, c)
    }
    /// `builder.rs:400-519` on the GPU.  Like the reference (which `.unwrap()`s, :517) this panics on failure; use
    /// [`HipBuilder::try_calculate_state_with_init`] to get the status instead.
    fn calculate_state_with_init<'a, It>(&mut self, it: It) -> Self::StateCalculation
    where
        Self::Register: 'a,
        It: IntoIterator<Item = (&'a Self::Register, usize)>,
    {
        self.try_calculate_state_with_init(it).expect("qip_hip state calculation failed")
    }
}

impl<P: HipPrecision> UnitaryBuilder<P> for HipBuilder<P> {
    fn vec_matrix_to_circuitobject(n: usize, data: Vec<Complex<P>>) -> Self::CircuitObject {
        LocalBuilder::<P>::vec_matrix_to_circuitobject(n, data)
    }
}

impl<P: HipPrecision> CliffordTBuilder<P> for HipBuilder<P> {
    fn make_x(&self) -> Self::CircuitObject {
        self.local.make_x()
    }
    fn make_y(&self) -> Self::CircuitObject {
        self.local.make_y()
    }
    fn make_z(&self) -> Self::CircuitObject {
        self.local.make_z()
    }
    fn make_h(&self) -> Self::CircuitObject {
        self.local.make_h()
    }
    fn make_s(&self) -> Self::CircuitObject {
        self.local.make_s()
    }
    fn make_t(&self) -> Self::CircuitObject {
        self.local.make_t()
    }
    fn make_cnot(&self) -> Self::CircuitObject {
        self.local.make_cnot()
    }
}

impl<P: HipPrecision> TemporaryRegisterBuilder for HipBuilder<P> {
    fn make_zeroed_temp_qubit(&mut self) -> Self::Register {
        self.local.make_zeroed_temp_qubit()
    }
    fn return_zeroed_temp_register(&mut self, r: Self::Register) {
        self.local.return_zeroed_temp_register(r)
    }
}

/// `basic_toffoli` / `toffoli` are provided methods built from the traits above (`builder_traits.rs:501-568`).
impl<P: HipPrecision> AdvancedCircuitBuilder<P> for HipBuilder<P> {}

impl<P: HipPrecision> MeasurementBuilder for HipBuilder<P> {
    type MeasurementHandle = HipMeasurementHandle;

    fn measure(&mut self, r: Self::Register) -> (Self::Register, Self::MeasurementHandle) {
        let (r, _inner_handle) = self.local.measure(r); // same id as ours: both count measure* calls (:609-611)
        let id = self.measurements;
        self.measurements += 1;
        (r, HipMeasurementHandle { id })
    }
}

impl<P: HipPrecision> StochasticMeasurementBuilder for HipBuilder<P> {
    type StochasticMeasurementHandle = HipStochasticMeasurementHandle;

    fn measure_stochastic(&mut self, r: Self::Register) -> (Self::Register, Self::StochasticMeasurementHandle) {
        let (r, _inner_handle) = self.local.measure_stochastic(r);
        let id = self.measurements;
        self.measurements += 1;
        (r, HipStochasticMeasurementHandle { id })
    }
}

impl<P: HipPrecision> RotationsBuilder<P> for HipBuilder<P> {
    fn rz(&mut self, r: Self::Register, theta: P) -> Self::Register {
        self.local.rz(r, theta)
    }
    fn rz_pi_by(&mut self, r: Self::Register, m: i64) -> CircuitResult<Self::Register> {
        self.local.rz_pi_by(r, m)
    }
}

impl<P: HipPrecision> Conditionable for HipBuilder<P> {
    fn try_apply_with_condition(
        &mut self,
        cr: Self::Register,
        r: Self::Register,
        co: Self::CircuitObject,
    ) -> Result<(Self::Register, Self::Register), CircuitError> {
        self.local.try_apply_with_condition(cr, r, co)
    }
}

impl<P: HipPrecision> Subcircuitable for HipBuilder<P> {
    type Subcircuit = Vec<(Vec<usize>, BuilderCircuitObject<P>)>;

    fn make_subcircuit(&self) -> CircuitResult<Self::Subcircuit> {
        self.local.make_subcircuit()
    }
    fn apply_subcircuit(&mut self, sc: Self::Subcircuit, r: Self::Register) -> CircuitResult<Self::Register> {
        self.local.apply_subcircuit(sc, r)
    }
}

impl<P: HipPrecision> Invertable for HipBuilder<P> {
    type SimilarBuilder = Self;

    fn new_similar(&self) -> Self {
        Self { device: self.device, tile: self.tile, tile_relabel: self.tile_relabel, ..Self::default() }
    }
    fn invert_subcircuit(sc: Self::Subcircuit) -> CircuitResult<Self::Subcircuit> {
        LocalBuilder::<P>::invert_subcircuit(sc)
    }
}

impl<P: HipPrecision> ConditionableSubcircuit for HipBuilder<P> {
    fn apply_conditioned_subcircuit(
        &mut self,
        sc: Self::Subcircuit,
        cr: Self::Register,
        r: Self::Register,
    ) -> Result<(Self::Register, Self::Register), CircuitError> {
        self.local.apply_conditioned_subcircuit(sc, cr, r)
    }
}

impl<P: HipPrecision> RecursiveCircuitBuilder<P> for HipBuilder<P> {
    type RecursiveSimilarBuilder = Self::SimilarBuilder;
}
