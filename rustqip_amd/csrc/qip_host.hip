// qip_host.hip — host-pointer twins of the reference functions (parity tests call these).
#include "qip_internal.h"

// ---------------------------------------------------------------------------------------
// host-pointer twin of apply_op / apply_op_overwrite
// ---------------------------------------------------------------------------------------
template <typename T>
static int apply_op_host_t(int dtype, uint32_t n, const qip_op* op, const void* in, uint64_t in_len,
                           void* out, uint64_t out_len, uint64_t in_off, uint64_t out_off,
                           int accumulate) {
  const uint64_t N = 1ull << n;
  const size_t ab = sizeof(amp_t<T>);
  FlatOp f;
  QCHK(flatten_op(n, op, false, &f));
  if (out_len == 0) return QIP_OK;
  qip_hip_state* s = nullptr;
  QCHK(qip_hip_state_create(n, dtype, 0, &s));
  int rc = QIP_OK;
  auto body = [&]() -> int {
    const bool full = in_off == 0 && out_off == 0 && in_len == N && out_len == N;
    if (full) {
      QCHK(qip_hip_state_upload(s, in, 0, N));
      QCHK(apply_op_t<T>(s, op));
      if (accumulate) {
        QCHK(ensure_alt(s));
        HIPCHK(hipMemcpyAsync(s->alt, out, N * ab, hipMemcpyHostToDevice, s->stream));
        hipLaunchKernelGGL((k_add_into<T>), dim3(grid_stride(N)), dim3(kBlock), 0, s->stream,
                           (amp_t<T>*)s->alt, (const amp_t<T>*)s->cur, N);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(out, s->alt, N * ab, hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
      } else {
        QCHK(qip_hip_state_download(s, out, 0, N));
      }
      return QIP_OK;
    }
    // windowed: literal gather kernel on dedicated buffers
    void *d_in = nullptr, *d_out = nullptr;
    HIPCHK(hipMalloc(&d_in, std::max<size_t>(in_len * ab, 16)));
    hipError_t e = hipMalloc(&d_out, out_len * ab);
    if (e != hipSuccess) {
      (void)hipFree(d_in);
      return fail(QIP_ERR_DEVICE, "hipMalloc failed: %s", hipGetErrorString(e));
    }
    auto inner = [&]() -> int {
      if (in_len) HIPCHK(hipMemcpyAsync(d_in, in, in_len * ab, hipMemcpyHostToDevice, s->stream));
      if (accumulate) HIPCHK(hipMemcpyAsync(d_out, out, out_len * ab, hipMemcpyHostToDevice, s->stream));
      QCHK(launch_gather<T>(s, f, (const amp_t<T>*)d_in, in_len, (amp_t<T>*)d_out, out_len, in_off,
                            out_off, accumulate));
      HIPCHK(hipMemcpyAsync(out, d_out, out_len * ab, hipMemcpyDeviceToHost, s->stream));
      HIPCHK(hipStreamSynchronize(s->stream));
      return QIP_OK;
    };
    int r2 = inner();
    (void)hipStreamSynchronize(s->stream);
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    return r2;
  };
  rc = body();
  std::string keep = g_last_error;
  qip_hip_state_destroy(s);
  if (rc != QIP_OK) g_last_error = keep;
  return rc;
}

extern "C" int qip_hip_apply_op_host(int dtype, uint32_t n, const qip_op* op, const void* in,
                                     uint64_t in_len, void* out, uint64_t out_len, uint64_t in_off,
                                     uint64_t out_off, int accumulate) try {
  if (dtype != QIP_C64 && dtype != QIP_C32) return fail(QIP_ERR_INVALID, "bad dtype %d", dtype);
  if ((in_len && !in) || (out_len && !out)) return fail(QIP_ERR_INVALID, "null buffer");
  if (n == 0 || n > 40) return fail(QIP_ERR_INVALID, "n = %u out of range [1, 40]", n);
  return dtype == QIP_C64
             ? apply_op_host_t<double>(dtype, n, op, in, in_len, out, out_len, in_off, out_off, accumulate)
             : apply_op_host_t<float>(dtype, n, op, in, in_len, out, out_len, in_off, out_off, accumulate);
} QIP_CATCH_ALL

// ---------------------------------------------------------------------------------------
// host-pointer twins of apply_op_row and of the windowed measurement functions
// ---------------------------------------------------------------------------------------
extern "C" int qip_hip_apply_op_row_host(int dtype, uint32_t n, const qip_op* op, const void* in, uint64_t in_len,
                                         uint64_t outputrow, uint64_t in_off, uint64_t out_off, void* out_value) try {
  if (!out_value) return fail(QIP_ERR_INVALID, "null output");
  // apply_op_row (matrix_ops.rs:38-59): the value of row out_off + outputrow = a one-row output window there
  return qip_hip_apply_op_host(dtype, n, op, in, in_len, out_value, 1, in_off, out_off + outputrow, 0);
} QIP_CATCH_ALL

template <typename T>
static int measure_probs_host_t(uint32_t n, const uint64_t* indices, uint32_t k, const void* in, uint64_t in_len,
                                uint64_t in_off, double* out) {
  if (k == 0 || k > n || k > 26 || !indices) return fail(QIP_ERR_INVALID, "bad measurement index list");
  MeasDesc md;
  memset(&md, 0, sizeof md);
  md.k = k;
  uint64_t seen = 0;
  for (uint32_t i = 0; i < k; ++i) {
    if (indices[i] >= n) return fail(QIP_ERR_INVALID, "measured qubit index out of range");
    if (seen & (1ull << indices[i])) return fail(QIP_ERR_INVALID, "repeated measured qubit index");
    seen |= 1ull << indices[i];
    md.mpos[i] = (uint32_t)(n - 1 - indices[i]);
  }
  const uint64_t outcomes = 1ull << k;
  for (uint64_t m = 0; m < outcomes; ++m) out[m] = 0.0;
  if (in_len == 0) return QIP_OK;
  if (qip_hip_device_count() <= 0) return fail(QIP_ERR_NO_DEVICE, "no HIP device visible: qip_hip has no CPU fallback");
  HIPCHK(hipSetDevice(0));
  void *d_in = nullptr, *d_out = nullptr;
  HIPCHK(hipMalloc(&d_in, in_len * sizeof(amp_t<T>)));
  hipError_t e = hipMalloc(&d_out, outcomes * sizeof(double));
  if (e == hipSuccess) e = hipMemcpy(d_in, in, in_len * sizeof(amp_t<T>), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemset(d_out, 0, outcomes * sizeof(double));
  if (e == hipSuccess) {
    hipLaunchKernelGGL((k_measure_probs_scatter<T>), dim3(grid_stride(in_len)), dim3(kBlock), 0, nullptr, (const amp_t<T>*)d_in,
                       in_len, md, in_off, (double*)d_out);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(out, d_out, outcomes * sizeof(double), hipMemcpyDeviceToHost);
  (void)hipFree(d_in);
  if (d_out) (void)hipFree(d_out);
  if (e != hipSuccess) return fail(QIP_ERR_DEVICE, "windowed measure_probs failed: %s", hipGetErrorString(e));
  return QIP_OK;
}

extern "C" int qip_hip_measure_probs_host(int dtype, uint32_t n, const uint64_t* indices, uint32_t k, const void* in,
                                          uint64_t in_len, uint64_t in_off, double* out) try {
  if (dtype != QIP_C64 && dtype != QIP_C32) return fail(QIP_ERR_INVALID, "bad dtype %d", dtype);
  if ((in_len && !in) || !out) return fail(QIP_ERR_INVALID, "null buffer");
  if (n == 0 || n > 62) return fail(QIP_ERR_INVALID, "n = %u out of range [1, 62]", n);
  if (in_off > (1ull << n) || in_len > (1ull << n) - in_off) return fail(QIP_ERR_INVALID, "window outside the 2^n vector");
  return dtype == QIP_C64 ? measure_probs_host_t<double>(n, indices, k, in, in_len, in_off, out)
                          : measure_probs_host_t<float>(n, indices, k, in, in_len, in_off, out);
} QIP_CATCH_ALL

extern "C" int qip_hip_measure_prob_host(int dtype, uint32_t n, uint64_t measured, const uint64_t* indices, uint32_t k,
                                         const void* in, uint64_t in_len, uint64_t in_off, double* out) try {
  if (!out) return fail(QIP_ERR_INVALID, "null output");
  if (k > 26) return fail(QIP_ERR_UNSUPPORTED, "windowed measure_prob over %u qubits", k);
  if (k > 0 && (measured >> k) != 0) return fail(QIP_ERR_INVALID, "measured value has more than k bits");
  std::vector<double> probs(1ull << k);
  QCHK(qip_hip_measure_probs_host(dtype, n, indices, k, in, in_len, in_off, probs.data()));
  *out = probs[measured];
  return QIP_OK;
} QIP_CATCH_ALL

