/*
 * qip_oracle.c — CPU ORACLE.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C restatement of the gate-application hot path of Renmusxd/RustQIP
 * (qip 1.5.0 / qip-iterators), used only as the checker by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing under
 * rustqip_amd/ may import, link or call it.
 *
 * The reference is Rust; no rustc/cargo exists in this image, so the reference
 * itself cannot be built (there is no oracle/_ref).  Parity is pinned instead by
 * replaying every golden vector of the reference's own tests for this path
 * (tests/test_oracle_golden.py; list in SURVEY.md Appendix B):
 *   qip-iterators/src/matrix_ops.rs:271-374, iterators/qubit_iterators.rs:289-379,
 *   qip/src/state_ops/matrix_ops.rs:306-377, state_ops/measurement_ops.rs:24-43,
 *   136-152, 290-335, qip-iterators/src/utils.rs doctests, qip/src/utils.rs doctests.
 * For general complex values the reference's tests pin nothing; there the oracle
 * rests on the published num-complex arithmetic definitions (see qip_oracle_impl.h).
 *
 * Build: see oracle/Makefile (gcc -O3 -fopenmp -ffp-contract=off).
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/qip_hip.h" /* type definitions only (qip_op, qip_c64, qip_c32) */

/* ---- qip-iterators/src/utils.rs ---------------------------------------- */

/* get_flat_index (utils.rs:5-8) */
uint64_t qip_oracle_get_flat_index(uint32_t nindices, uint64_t i, uint64_t j) {
  uint64_t mat_side = (uint64_t)1 << nindices;
  return (i * mat_side) + j;
}

/* flip_bits (utils.rs:22-25): reverse the low n bits */
uint64_t qip_oracle_flip_bits(uint32_t n, uint64_t num) {
  uint64_t rev = 0;
  for (int b = 0; b < 64; ++b)
    if ((num >> b) & 1) rev |= (uint64_t)1 << (63 - b);
  uint32_t leading_zeros = 64 - n;
  return leading_zeros >= 64 ? 0 : rev >> leading_zeros;
}

/* set_bit (utils.rs:37-44) */
uint64_t qip_oracle_set_bit(uint64_t num, uint32_t bit_index, int value) {
  uint64_t v = (uint64_t)1 << bit_index;
  return value ? (num | v) : (num & ~v);
}

/* get_bit (utils.rs:55-57) */
int qip_oracle_get_bit(uint64_t num, uint32_t bit_index) { return ((num >> bit_index) & 1) != 0; }

/* ---- qip/src/utils.rs --------------------------------------------------- */

/* entwine_bits (qip/src/utils.rs:21-43) */
uint64_t qip_oracle_entwine_bits(uint32_t n, uint64_t selector, uint64_t off_bits, uint64_t on_bits) {
  uint64_t result = 0;
  for (uint32_t i = 0; i < n; ++i) {
    if ((selector & 1) == 0) {
      result |= (off_bits & 1) << i;
      off_bits >>= 1;
    } else {
      result |= (on_bits & 1) << i;
      on_bits >>= 1;
    }
    selector >>= 1;
  }
  return result;
}

/* extract_bits (qip/src/utils.rs:54-60) */
uint64_t qip_oracle_extract_bits(uint64_t num, const uint64_t* indices, uint32_t k) {
  uint64_t acc = 0;
  for (uint32_t i = 0; i < k; ++i) acc |= ((num >> indices[i]) & 1) << i;
  return acc;
}

/* ---- qip-iterators/src/matrix_ops.rs index maps ------------------------- */

/* full_to_sub (matrix_ops.rs:12-21): indices[0] lands in the MSB of the sub-index */
uint64_t qip_oracle_full_to_sub(uint32_t n, const uint64_t* mat_indices, uint32_t nindices,
                                uint64_t full_index) {
  uint64_t acc = 0;
  for (uint32_t j = 0; j < nindices; ++j) {
    int bit = qip_oracle_get_bit(full_index, (uint32_t)(n - 1 - mat_indices[j]));
    acc = qip_oracle_set_bit(acc, nindices - 1 - j, bit);
  }
  return acc;
}

/* sub_to_full (matrix_ops.rs:24-30) */
uint64_t qip_oracle_sub_to_full(uint32_t n, const uint64_t* mat_indices, uint32_t nindices,
                                uint64_t sub_index, uint64_t base) {
  uint64_t acc = base;
  for (uint32_t j = 0; j < nindices; ++j) {
    int bit = qip_oracle_get_bit(sub_index, nindices - 1 - j);
    acc = qip_oracle_set_bit(acc, (uint32_t)(n - 1 - mat_indices[j]), bit);
  }
  return acc;
}

/* ---- precision-generic body --------------------------------------------- */

#define REAL double
#define CPLX qip_c64
#define FN(x) x##_c64
#define SQRT sqrt
#include "qip_oracle_impl.h"
#undef REAL
#undef CPLX
#undef FN
#undef SQRT

#define REAL float
#define CPLX qip_c32
#define FN(x) x##_c32
#define SQRT sqrtf
#include "qip_oracle_impl.h"
#undef REAL
#undef CPLX
#undef FN
#undef SQRT

/* ---- real / integer P (the generic `P` of qip-iterators' apply_op) --------- */

#define RT double
#define FN(x) x##_f64
#include "qip_oracle_real_impl.h"
#undef RT
#undef FN
#define RT float
#define FN(x) x##_f32
#include "qip_oracle_real_impl.h"
#undef RT
#undef FN
#define RT uint64_t
#define FN(x) x##_i64
#include "qip_oracle_real_impl.h"
#undef RT
#undef FN
#define RT uint32_t
#define FN(x) x##_i32
#include "qip_oracle_real_impl.h"
#undef RT
#undef FN

int qip_oracle_max_threads(void) { return omp_get_max_threads(); }
/* the CPU-baseline leg times the same loop with 1 thread and with all of them (bench.py) */
void qip_oracle_set_num_threads(int n) { omp_set_num_threads(n > 0 ? n : 1); }
