//! Writer for the flat circuit-replay format "qipc 1" (specified in `rustqip_amd/replay.py`; read by that module
//! and by `rustqip_amd/host/qip_replay.hpp` / `tools/qip_replay`).  For setups where the program that builds the
//! circuit cannot link `libqip_hip.so`: dump the lowered pipeline, replay it where the GPU is.
//!
//! This file needs only `qip`, not the C ABI: it can be copied into any RustQIP program.
use crate::builder::lower;
use num_complex::Complex;
use qip::builder::{BuilderCircuitObject, BuilderCircuitObjectType, MeasurementObject};
use qip_iterators::iterators::MatrixOp;
use std::io::{self, Write};

fn num(w: &mut impl Write, z: &Complex<f64>) -> io::Result<()> {
    // `{:?}` of an f64 is the shortest decimal that parses back to the same value
    write!(w, " {:?} {:?}", z.re, z.im)
}

/// One `MatrixOp` as statement tokens (`matrix ..`, `sparse ..`, `swap ..`, `control .. <inner>`).
pub fn write_op(w: &mut impl Write, op: &MatrixOp<Complex<f64>>) -> io::Result<()> {
    match op {
        MatrixOp::Matrix(indices, data) => {
            write!(w, "matrix {}", indices.len())?;
            for i in indices {
                write!(w, " {i}")?;
            }
            for z in data {
                num(w, z)?;
            }
        }
        MatrixOp::SparseMatrix(indices, rows) => {
            write!(w, "sparse {}", indices.len())?;
            for i in indices {
                write!(w, " {i}")?;
            }
            for row in rows {
                write!(w, " {}", row.len())?;
                for (col, z) in row {
                    write!(w, " {col}")?;
                    num(w, z)?;
                }
            }
        }
        MatrixOp::Swap(h, indices) => {
            write!(w, "swap {h}")?;
            for i in indices {
                write!(w, " {i}")?;
            }
        }
        MatrixOp::Control(nc, indices, inner) => {
            write!(w, "control {nc}")?;
            for i in &indices[..*nc] {
                write!(w, " {i}")?;
            }
            write!(w, " ")?;
            write_op(w, inner)?;
        }
    }
    Ok(())
}

/// The pipeline of a `LocalBuilder<f64>` (`make_subcircuit()`, `builder.rs:831-833`) as a replay file.
/// `rand_u01` supplies the uniform sample of each collapse measurement, in pipeline order.
pub fn write_pipeline(
    w: &mut impl Write,
    n: usize,
    initial_index: usize,
    pipeline: &[(Vec<usize>, BuilderCircuitObject<f64>)],
    mut rand_u01: impl FnMut() -> f64,
) -> io::Result<()> {
    writeln!(w, "qipc 1")?;
    writeln!(w, "n {n}")?;
    if initial_index != 0 {
        writeln!(w, "init {initial_index}")?;
    }
    for (indices, obj) in pipeline {
        match obj.object() {
            BuilderCircuitObjectType::Unitary(u) => {
                let lowered = lower(indices, u).map_err(|e| io::Error::new(io::ErrorKind::InvalidData, format!("{e:?}")))?;
                if let Some(op) = lowered {
                    write_op(w, &op)?;
                    writeln!(w)?;
                }
            }
            BuilderCircuitObjectType::Measurement(kind) => {
                let word = match kind {
                    MeasurementObject::Measurement => "measure",
                    MeasurementObject::StochasticMeasurement => "probs",
                };
                write!(w, "{word} {}", indices.len())?;
                for i in indices {
                    write!(w, " {i}")?;
                }
                if matches!(kind, MeasurementObject::Measurement) {
                    write!(w, " {:?}", rand_u01())?;
                }
                writeln!(w)?;
            }
        }
    }
    Ok(())
}
