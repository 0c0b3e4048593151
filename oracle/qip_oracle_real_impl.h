/*
 * TEST INFRASTRUCTURE — the CPU oracle's row kernel for a REAL or INTEGER `P`, instantiated once per element type by
 * qip_oracle.c (RT = double / float / uint64_t / uint32_t, FN(x) = x##_f64 / _f32 / _i64 / _i32).
 *
 * qip-iterators' kernel is generic over P (matrix_ops.rs:98-107: `P: Sum + AddAssign + Clone + One + Zero + Mul + Send +
 * Sync`); its own unit tests run it on i32 (matrix_ops.rs:271-374), its benches on f64 (benches/matmul_bench.rs).  This file
 * restates the same functions as qip_oracle_impl.h — apply_op_row_indices, sum_for_op_cols, the four row iterators, apply_op /
 * apply_op_overwrite — with P's own arithmetic: Mul and Add of the scalar, Sum = fold(P::zero(), +), is_zero = (v == 0).
 * Integers are computed in the unsigned type of their width: the bits of wrapping two's-complement arithmetic (Rust in release
 * builds; a debug build panics on overflow, which no test vector reaches).  -ffp-contract=off as everywhere in the oracle.
 */

/* the closure `f` of apply_op_row_indices (matrix_ops.rs:78-90) */
static inline RT FN(rterm)(uint32_t n, const uint64_t* idx, uint32_t k, uint64_t row, uint64_t col, RT val, const RT* input,
                           uint64_t in_len, uint64_t in_off) {
  uint64_t colbits = qip_oracle_sub_to_full(n, idx, k, col, row); /* :79 */
  if (colbits < in_off) return (RT)0;                              /* :80-81 */
  uint64_t vecrow = colbits - in_off;                              /* :83 */
  if (vecrow >= in_len) return (RT)0;                              /* :84-85 */
  return val * input[vecrow];                                      /* :87 */
}

/* the columns of an inner (non-Control) op for `row`, each shifted by `shift` (qubit_iterators.rs:155-158), folded from zero */
static RT FN(rsum_inner)(const qip_op* op, uint32_t k_op, uint64_t row, uint64_t shift, uint32_t n, const uint64_t* idx,
                         uint32_t k_all, uint64_t full_row, const RT* input, uint64_t in_len, uint64_t in_off) {
  RT acc = (RT)0;
  switch (op->kind) {
    case QIP_OP_MATRIX: { /* MatrixOpIterator (qubit_iterators.rs:23-55): ascending columns, zero entries skipped (:49) */
      const RT* rowdata = (const RT*)op->dense + qip_oracle_get_flat_index(k_op, row, 0);
      uint64_t side = (uint64_t)1 << k_op;
      for (uint64_t col = 0; col < side; ++col) {
        RT v = rowdata[col];
        if (!(v == (RT)0)) acc = acc + FN(rterm)(n, idx, k_all, full_row, col + shift, v, input, in_len, in_off);
      }
      break;
    }
    case QIP_OP_SPARSE: { /* SparseMatrixOpIterator (:73-101): stored order, nothing filtered */
      const RT* vals = (const RT*)op->sparse_vals;
      for (uint64_t p = op->sparse_rowptr[row]; p < op->sparse_rowptr[row + 1]; ++p)
        acc = acc + FN(rterm)(n, idx, k_all, full_row, op->sparse_cols[p] + shift, vals[p], input, in_len, in_off);
      break;
    }
    case QIP_OP_SWAP: { /* SwapOpIterator (:195-218) */
      uint32_t half_n = k_op >> 1;
      uint64_t lower_mask = ~(~(uint64_t)0 << half_n);
      uint64_t col = ((row & lower_mask) << half_n) + (row >> half_n);
      acc = acc + FN(rterm)(n, idx, k_all, full_row, col + shift, (RT)1, input, in_len, in_off);
      break;
    }
    default:
      break;
  }
  return acc;
}

/* apply_op_row_indices (matrix_ops.rs:62-94) + sum_for_op_cols / sum_for_control_iterator (ops.rs:100-156) +
 * ControlledOpIterator (qubit_iterators.rs:124-171) */
RT FN(qip_oracle_apply_op_row)(uint32_t n, const qip_op* op, const RT* input, uint64_t in_len, uint64_t outputrow,
                               uint64_t in_off, uint64_t out_off) {
  const uint64_t* idx = op->indices;
  uint32_t k = op->n_indices;
  uint64_t row = out_off + outputrow;                       /* :74 */
  uint64_t matrow = qip_oracle_full_to_sub(n, idx, k, row); /* :75 */
  if (op->kind != QIP_OP_CONTROL) return FN(rsum_inner)(op, k, matrow, 0, n, idx, k, row, input, in_len, in_off);
  uint32_t n_control = op->n_controls;
  uint32_t n_op = op->n_indices - op->n_controls;
  const qip_op* inner = op->inner;
  while (inner->kind == QIP_OP_CONTROL) { /* ops.rs:150-154 */
    n_control = n_control + inner->n_controls;
    n_op = inner->n_indices - inner->n_controls;
    inner = inner->inner;
  }
  uint64_t index_threshold = ((uint64_t)1 << (n_control + n_op)) - ((uint64_t)1 << n_op); /* qubit_iterators.rs:130-131 */
  if (matrow >= index_threshold)                                                          /* :132 */
    return FN(rsum_inner)(inner, n_op, matrow - index_threshold, index_threshold, n, idx, k, row, input, in_len, in_off);
  RT acc = (RT)0; /* :160-169: exactly one (row, 1) */
  acc = acc + FN(rterm)(n, idx, k, row, matrow, (RT)1, input, in_len, in_off);
  return acc;
}

/* apply_op (matrix_ops.rs:98-123) / apply_op_overwrite (:127-152) */
void FN(qip_oracle_apply_op)(uint32_t n, const qip_op* op, const RT* input, uint64_t in_len, RT* output, uint64_t out_len,
                             uint64_t in_off, uint64_t out_off, int accumulate, int nthreads) {
  int64_t rows = (int64_t)out_len;
  if (nthreads <= 0) nthreads = rows < (1 << 20) ? 1 : omp_get_max_threads();
#pragma omp parallel for schedule(static) num_threads(nthreads)
  for (int64_t r = 0; r < rows; ++r) {
    RT y = FN(qip_oracle_apply_op_row)(n, op, input, in_len, (uint64_t)r, in_off, out_off);
    output[r] = accumulate ? (RT)(output[r] + y) : y; /* :110 / :139 */
  }
}
