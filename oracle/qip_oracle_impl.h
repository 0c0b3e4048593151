/*
 * TEST INFRASTRUCTURE — body of the CPU oracle, instantiated once per precision
 * by qip_oracle.c (REAL = double / float, CPLX = qip_c64 / qip_c32, FN(x) = x##_c64 / _c32).
 *
 * It restates, literally and without re-optimising, the algorithm of
 *   qip-iterators/src/matrix_ops.rs            (row kernel, apply_op, apply_op_overwrite)
 *   qip-iterators/src/iterators/ops.rs         (per-variant dispatch, nested Control)
 *   qip-iterators/src/iterators/qubit_iterators.rs (the four row iterators)
 *   qip/src/state_ops/measurement_ops.rs       (measurement)
 * Scalar arithmetic follows num-complex ^0.4 (not vendored in /root/reference, version
 * unpinned by a lockfile): Mul = (ac - bd, ad + bc), Add componentwise, Sum = fold(0, +),
 * is_zero = re == 0 && im == 0, norm_sqr = re*re + im*im, Complex * real = (re*p, im*p).
 * Built with -ffp-contract=off so no product is fused into an add (rustc never contracts).
 */

/* num-complex Mul<Complex<T>> for Complex<T> */
static inline CPLX FN(cmul)(CPLX a, CPLX b) {
  CPLX r;
  r.re = a.re * b.re - a.im * b.im;
  r.im = a.re * b.im + a.im * b.re;
  return r;
}
static inline CPLX FN(cadd)(CPLX a, CPLX b) {
  CPLX r;
  r.re = a.re + b.re;
  r.im = a.im + b.im;
  return r;
}
static inline int FN(cis_zero)(CPLX a) { return a.re == (REAL)0 && a.im == (REAL)0; }

/* The closure `f` of apply_op_row_indices (matrix_ops.rs:78-90): column `col` of the
 * op matrix with value `val` -> val * input[...] or zero when outside the window. */
static inline CPLX FN(term)(uint32_t n, const uint64_t* idx, uint32_t k, uint64_t row,
                            uint64_t col, CPLX val, const CPLX* input, uint64_t in_len,
                            uint64_t in_off) {
  CPLX zero = {(REAL)0, (REAL)0};
  uint64_t colbits = qip_oracle_sub_to_full(n, idx, k, col, row); /* :79 */
  if (colbits < in_off) return zero;                               /* :80-81 */
  uint64_t vecrow = colbits - in_off;                              /* :83 */
  if (vecrow >= in_len) return zero;                               /* :84-85 */
  return FN(cmul)(val, input[vecrow]);                             /* :87 */
}

/*
 * Sum over the columns an inner (non-Control) op yields for `row`, with every column
 * shifted by `shift` (0 for a bare op, index_threshold under a Control:
 * qubit_iterators.rs:155-158).  `k_op` is the number of indices the inner iterator is
 * built with (ops.rs:104-110 bare; ops.rs:129-148 under Control).
 * Returns fold(0, +) in iterator order (std::iter::Sum).
 */
static CPLX FN(sum_inner)(const qip_op* op, uint32_t k_op, uint64_t row, uint64_t shift,
                          uint32_t n, const uint64_t* idx, uint32_t k_all, uint64_t full_row,
                          const CPLX* input, uint64_t in_len, uint64_t in_off) {
  CPLX acc = {(REAL)0, (REAL)0};
  CPLX one = {(REAL)1, (REAL)0};
  switch (op->kind) {
    case QIP_OP_MATRIX: {
      /* MatrixOpIterator (qubit_iterators.rs:23-55): the row slice
       * data[row*2^k .. (row+1)*2^k], ascending columns, entries equal to zero skipped (:49) */
      const CPLX* data = (const CPLX*)op->dense;
      uint64_t side = (uint64_t)1 << k_op;
      const CPLX* rowdata = data + qip_oracle_get_flat_index(k_op, row, 0);
      for (uint64_t col = 0; col < side; ++col) {
        CPLX v = rowdata[col];
        if (!FN(cis_zero)(v))
          acc = FN(cadd)(acc, FN(term)(n, idx, k_all, full_row, col + shift, v, input, in_len, in_off));
      }
      break;
    }
    case QIP_OP_SPARSE: {
      /* SparseMatrixOpIterator (:73-101): data[row] verbatim, stored order, nothing filtered */
      const CPLX* vals = (const CPLX*)op->sparse_vals;
      for (uint64_t p = op->sparse_rowptr[row]; p < op->sparse_rowptr[row + 1]; ++p)
        acc = FN(cadd)(acc, FN(term)(n, idx, k_all, full_row, op->sparse_cols[p] + shift, vals[p],
                                     input, in_len, in_off));
      break;
    }
    case QIP_OP_SWAP: {
      /* SwapOpIterator (:195-218): one column, the two halves of the sub-index exchanged */
      uint32_t half_n = k_op >> 1;                                  /* :198 */
      uint64_t lower_mask = ~(~(uint64_t)0 << half_n);              /* :210 */
      uint64_t lower = row & lower_mask;
      uint64_t upper = row >> half_n;
      uint64_t col = (lower << half_n) + upper;                     /* :213 */
      acc = FN(cadd)(acc, FN(term)(n, idx, k_all, full_row, col + shift, one, input, in_len, in_off));
      break;
    }
    default:
      break; /* Control handled by the caller */
  }
  return acc;
}

/* apply_op_row_indices (matrix_ops.rs:62-94) + MatrixOp::sum_for_op_cols (ops.rs:100-116)
 * + sum_for_control_iterator (ops.rs:118-156) + ControlledOpIterator (qubit_iterators.rs:124-171) */
CPLX FN(qip_oracle_apply_op_row)(uint32_t n, const qip_op* op, const CPLX* input, uint64_t in_len,
                                 uint64_t outputrow, uint64_t in_off, uint64_t out_off) {
  const uint64_t* idx = op->indices; /* op.indices(): the OUTER list only (matrix_ops.rs:108) */
  uint32_t k = op->n_indices;
  uint64_t row = out_off + outputrow;                       /* :74 */
  uint64_t matrow = qip_oracle_full_to_sub(n, idx, k, row); /* :75 */

  if (op->kind != QIP_OP_CONTROL)
    return FN(sum_inner)(op, k, matrow, 0, n, idx, k, row, input, in_len, in_off);

  /* ops.rs:111-114 */
  uint32_t n_control = op->n_controls;
  uint32_t n_op = op->n_indices - op->n_controls;
  const qip_op* inner = op->inner;
  /* ops.rs:150-154: nested controls accumulate, op count comes from the nested list */
  while (inner->kind == QIP_OP_CONTROL) {
    n_control = n_control + inner->n_controls;
    n_op = inner->n_indices - inner->n_controls;
    inner = inner->inner;
  }
  uint32_t n_indices = n_control + n_op;                                        /* qubit_iterators.rs:130 */
  uint64_t index_threshold = ((uint64_t)1 << n_indices) - ((uint64_t)1 << n_op); /* :131 */
  if (matrow >= index_threshold) {                                              /* :132 */
    return FN(sum_inner)(inner, n_op, matrow - index_threshold, index_threshold, n, idx, k, row,
                         input, in_len, in_off);
  } else {
    /* :160-169: exactly one (row, 1) */
    CPLX acc = {(REAL)0, (REAL)0};
    CPLX one = {(REAL)1, (REAL)0};
    acc = FN(cadd)(acc, FN(term)(n, idx, k, row, matrow, one, input, in_len, in_off));
    return acc;
  }
}

/* apply_op (matrix_ops.rs:98-123, accumulate) / apply_op_overwrite (:127-152).
 * The rayon par_iter_mut over rows (:122,:151) becomes a static OpenMP split. */
void FN(qip_oracle_apply_op)(uint32_t n, const qip_op* op, const CPLX* input, uint64_t in_len,
                             CPLX* output, uint64_t out_len, uint64_t in_off, uint64_t out_off,
                             int accumulate, int nthreads) {
  int64_t rows = (int64_t)out_len;
  /* tiny vectors: forking a team on a many-core host costs far more than the loop */
  if (nthreads <= 0) nthreads = rows < (1 << 20) ? 1 : omp_get_max_threads();
#pragma omp parallel for schedule(static) num_threads(nthreads)
  for (int64_t r = 0; r < rows; ++r) {
    CPLX y = FN(qip_oracle_apply_op_row)(n, op, input, in_len, (uint64_t)r, in_off, out_off);
    if (accumulate) {
      output[r] = FN(cadd)(output[r], y); /* *outputloc += ...  (:110) */
    } else {
      output[r] = y;                      /* *outputloc = ...   (:139) */
    }
  }
}

/* ---------------- measurement (qip/src/state_ops/measurement_ops.rs) ---------------- */

/* prob_magnitude (:11-13) */
REAL FN(qip_oracle_prob_magnitude)(const CPLX* input, uint64_t len) {
  REAL s = (REAL)0;
  for (uint64_t i = 0; i < len; ++i) s += input[i].re * input[i].re + input[i].im * input[i].im;
  return s;
}

/* measure_prob (:44-58) via measure_prob_fn (:65-112).  Sequential sum over the
 * remaining-index space in increasing order (rayon's order is unspecified). */
REAL FN(qip_oracle_measure_prob)(uint32_t n, uint64_t measured, const uint64_t* indices, uint32_t k,
                                 const CPLX* input, uint64_t in_len, uint64_t in_off) {
  uint64_t templ = 0; /* :72-79 */
  for (uint32_t i = 0; i < k; ++i) {
    uint64_t sel_bit = (measured >> i) & 1;
    templ |= sel_bit << (n - 1 - indices[i]);
  }
  uint64_t remaining[64]; /* :80 */
  uint32_t nrem = 0;
  for (uint64_t q = 0; q < n; ++q) {
    int found = 0;
    for (uint32_t i = 0; i < k; ++i)
      if (indices[i] == q) found = 1;
    if (!found) remaining[nrem++] = q;
  }
  REAL sum = (REAL)0;
  uint64_t count = (uint64_t)1 << nrem; /* :109 */
  for (uint64_t rbits = 0; rbits < count; ++rbits) {
    uint64_t tmp_index = 0; /* :83-91 */
    for (uint32_t i = 0; i < nrem; ++i) {
      uint64_t sel_bit = (rbits >> i) & 1;
      tmp_index |= sel_bit << (n - 1 - remaining[i]);
    }
    uint64_t index = tmp_index + templ; /* :92 */
    if (index < in_off) continue;       /* :93-94 */
    index -= in_off;
    CPLX amp = {(REAL)0, (REAL)0};      /* :50-56 */
    if (index < in_len) amp = input[index];
    if (amp.re == (REAL)0 && amp.im == (REAL)0) continue; /* :98-99 */
    sum += amp.re * amp.re + amp.im * amp.im;             /* :101 */
  }
  return sum;
}

/* measure_probs (:115-127) */
void FN(qip_oracle_measure_probs)(uint32_t n, const uint64_t* indices, uint32_t k, const CPLX* input,
                                  uint64_t in_len, uint64_t in_off, REAL* out) {
  uint64_t count = (uint64_t)1 << k;
  for (uint64_t m = 0; m < count; ++m)
    out[m] = FN(qip_oracle_measure_prob)(n, m, indices, k, input, in_len, in_off);
}

/* soft_measure (:153-176), with rand::random::<f64>() (:160) supplied as `rand_u01`. */
uint64_t FN(qip_oracle_soft_measure)(uint32_t n, const uint64_t* indices, uint32_t k,
                                     const CPLX* input, uint64_t in_len, uint64_t in_off,
                                     double rand_u01) {
  REAL r = (REAL)rand_u01; /* :160 */
  if (in_len < ((uint64_t)1 << n)) r = r * FN(qip_oracle_prob_magnitude)(input, in_len); /* :161-165 */
  uint64_t measured_indx = 0;
  for (uint64_t i = 0; i < in_len; ++i) { /* :167-173 */
    r -= input[i].re * input[i].re + input[i].im * input[i].im;
    if (r <= (REAL)0) {
      measured_indx = i + in_off;
      break;
    }
  }
  uint64_t pos[64]; /* :174 */
  for (uint32_t i = 0; i < k; ++i) pos[i] = n - 1 - indices[i];
  return qip_oracle_extract_bits(measured_indx, pos, k); /* :175 */
}

/* measure_state (:220-269).  Returns 1 if output was written, 0 for the p == 0 no-op (:230). */
int FN(qip_oracle_measure_state)(uint32_t n, const uint64_t* indices, uint32_t k, uint64_t measured,
                                 REAL measured_prob, const CPLX* input, uint64_t in_len,
                                 CPLX* output, uint64_t out_len, uint64_t in_off, uint64_t out_off) {
  if (measured_prob == (REAL)0) return 0;            /* :230 */
  REAL p_mult = (REAL)1 / SQRT(measured_prob);       /* :231 */
  uint64_t row_mask = 0, measured_mask = 0;          /* :233-241 */
  for (uint32_t i = 0; i < k; ++i) {
    row_mask += (uint64_t)1 << (n - 1 - indices[i]);
    measured_mask += ((measured >> i) & 1) << (n - 1 - indices[i]);
  }
  uint64_t lower = in_off > out_off ? in_off : out_off; /* :243-248 */
  uint64_t upper_in = in_off + in_len, upper_out = out_off + out_len;
  uint64_t upper = upper_in < upper_out ? upper_in : upper_out;
  if (upper < lower) return 1;
  const CPLX* in = input + (lower - in_off);
  CPLX* out = output + (lower - out_off);
  for (uint64_t i = 0; i < upper - lower; ++i) { /* :250-263 */
    uint64_t row = i + lower;
    if (((row & row_mask) ^ measured_mask) != 0) {
      out[i].re = (REAL)0;
      out[i].im = (REAL)0;
    } else {
      out[i].re = in[i].re * p_mult;
      out[i].im = in[i].im * p_mult;
    }
  }
  return 1;
}

/* measure (:190-214): forced >= 0 plays MeasuredCondition{measured, prob: None}. */
void FN(qip_oracle_measure)(uint32_t n, const uint64_t* indices, uint32_t k, const CPLX* input,
                            uint64_t in_len, CPLX* output, uint64_t out_len, int64_t forced,
                            double rand_u01, uint64_t* measured, REAL* prob) {
  uint64_t m = forced >= 0 ? (uint64_t)forced
                           : FN(qip_oracle_soft_measure)(n, indices, k, input, in_len, 0, rand_u01);
  REAL p = FN(qip_oracle_measure_prob)(n, m, indices, k, input, in_len, 0);
  FN(qip_oracle_measure_state)(n, indices, k, m, p, input, in_len, output, out_len, 0, 0);
  *measured = m;
  *prob = p;
}
