// qip_hip.hpp — C++17 host mirror of the reference's interface for the gate-application path,
// over the C ABI in include/qip_hip.h.  Header-only; link with -lqip_hip.
//
// The reference is Rust and no Rust toolchain exists in the build image, so the compiled-language
// host side is written in C++ with the reference's names, argument order and error behaviour:
//   qip::MatrixOp<P>                          qip-iterators/src/iterators/ops.rs:11-91
//   qip::make_matrix_op / make_sparse_matrix_op / make_swap_op / make_control_op
//                                             qip/src/state_ops/matrix_ops.rs:12-122
//   qip::apply_op / apply_op_overwrite        qip-iterators/src/matrix_ops.rs:98-152
//   qip::CircuitError                         qip/src/errors.rs:6-22
//   qip::HipBuilder<P>                        LocalBuilder<P>'s recording + run loop, qip/src/builder.rs:325-519
// Nothing in this file touches amplitudes: every compute call goes to libqip_hip.so (HIP kernels).
#pragma once

#include <cmath>
#include <complex>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "qip_hip.h"

namespace qip {

/// qip::errors::CircuitError::Generic(String)
class CircuitError : public std::runtime_error {
 public:
  explicit CircuitError(const std::string& msg) : std::runtime_error(msg) {}
};

/// qip/src/types.rs:16-22
enum class Representation { BigEndian, LittleEndian };

template <typename P> struct dtype_of;
template <> struct dtype_of<double> { static constexpr int value = QIP_C64; };
template <> struct dtype_of<float> { static constexpr int value = QIP_C32; };

inline void check(int rc) {
  if (rc == QIP_OK) return;
  throw CircuitError(std::string(qip_hip_last_error()));
}

/// qip-iterators/src/utils.rs:22-25
inline size_t flip_bits(size_t n, size_t num) {
  size_t out = 0;
  for (size_t b = 0; b < n; ++b)
    if ((num >> b) & 1) out |= size_t(1) << (n - 1 - b);
  return out;
}

/// MatrixOp<E> of qip-iterators (ops.rs:11-20), E the ELEMENT type of the vectors it applies to: the reference's enum is
/// generic (its kernel runs on i32 in its tests, on f64 in its benches).  std::complex<P> is layout-compatible with
/// qip_c64 / qip_c32 and with num_complex::Complex<P>; a real / integer E is passed as is (enum qip_dtype).
template <typename E> class BasicMatrixOp {
 public:
  using C = E;
  using SparseRows = std::vector<std::vector<std::pair<size_t, C>>>;
  enum class Kind { Matrix, SparseMatrix, Swap, Control };

  static BasicMatrixOp new_matrix(std::vector<size_t> indices, std::vector<C> data) {
    BasicMatrixOp op(Kind::Matrix, std::move(indices));
    op.data_ = std::move(data);
    return op;
  }
  static BasicMatrixOp new_sparse(std::vector<size_t> indices, SparseRows rows) {
    BasicMatrixOp op(Kind::SparseMatrix, std::move(indices));
    op.rows_ = std::move(rows);
    return op;
  }
  static BasicMatrixOp new_swap(std::vector<size_t> a, const std::vector<size_t>& b) {
    const size_t h = a.size();
    a.insert(a.end(), b.begin(), b.end());
    BasicMatrixOp op(Kind::Swap, std::move(a));
    op.half_ = h;
    return op;
  }
  static BasicMatrixOp new_control(std::vector<size_t> c, const std::vector<size_t>& r, BasicMatrixOp inner) {
    const size_t nc = c.size();
    c.insert(c.end(), r.begin(), r.end());
    BasicMatrixOp op(Kind::Control, std::move(c));
    op.n_controls_ = nc;
    op.inner_ = std::make_shared<BasicMatrixOp>(std::move(inner));
    return op;
  }

  Kind kind() const { return kind_; }
  size_t num_indices() const { return kind_ == Kind::Swap ? 2 * half_ : indices_.size(); }  // ops.rs:24-36
  const std::vector<size_t>& indices() const { return indices_; }                           // ops.rs:39-46
  size_t n_controls() const { return n_controls_; }
  const std::vector<C>& data() const { return data_; }
  const SparseRows& rows() const { return rows_; }
  const BasicMatrixOp* inner() const { return inner_.get(); }
  std::shared_ptr<BasicMatrixOp> inner_ptr() const { return inner_; }

  /// The `struct qip_op` tree for the C ABI.  It owns every buffer it points into (including a copy
  /// of the matrix data), so it stays valid after the BasicMatrixOp it was made from is gone.
  struct CView {
    qip_op op{};
    std::vector<uint64_t> idx, rowptr, cols;
    std::vector<C> vals, dense;
    std::unique_ptr<CView> inner;
  };
  std::unique_ptr<CView> to_c() const {
    auto v = std::make_unique<CView>();
    v->idx.assign(indices_.begin(), indices_.end());
    v->op.n_indices = (uint32_t)v->idx.size();
    v->op.indices = v->idx.data();
    switch (kind_) {
      case Kind::Matrix:
        v->op.kind = QIP_OP_MATRIX;
        if (data_.size() != (size_t(1) << (2 * indices_.size())))
          throw CircuitError("Matrix data has " + std::to_string(data_.size()) + " entries versus expected 2^2*" +
                             std::to_string(indices_.size()));
        v->dense = data_;
        v->op.dense = v->dense.data();
        break;
      case Kind::SparseMatrix:
        v->op.kind = QIP_OP_SPARSE;
        if (rows_.size() != (size_t(1) << indices_.size()))
          throw CircuitError("Sparse matrix has " + std::to_string(rows_.size()) + " rows versus expected 2^" +
                             std::to_string(indices_.size()));
        v->rowptr.push_back(0);
        for (const auto& row : rows_) {
          for (const auto& e : row) {
            v->cols.push_back(e.first);
            v->vals.push_back(e.second);
          }
          v->rowptr.push_back(v->cols.size());
        }
        v->op.sparse_rowptr = v->rowptr.data();
        v->op.sparse_cols = v->cols.data();
        v->op.sparse_vals = v->vals.data();
        break;
      case Kind::Swap:
        v->op.kind = QIP_OP_SWAP;
        break;
      case Kind::Control:
        v->op.kind = QIP_OP_CONTROL;
        v->op.n_controls = (uint32_t)n_controls_;
        if (!inner_) throw CircuitError("Control op without inner op");
        v->inner = inner_->to_c();
        v->op.inner = &v->inner->op;
        break;
    }
    return v;
  }

 private:
  BasicMatrixOp(Kind k, std::vector<size_t> idx) : kind_(k), indices_(std::move(idx)) {}
  Kind kind_;
  std::vector<size_t> indices_;
  std::vector<C> data_;
  SparseRows rows_;
  size_t half_ = 0, n_controls_ = 0;
  std::shared_ptr<BasicMatrixOp> inner_;
};

/// The `qip` crate's MatrixOp<Complex<P>> (what LocalBuilder lowers gates to): P is the real precision of the amplitudes.
template <typename P> using MatrixOp = BasicMatrixOp<std::complex<P>>;

/// enum qip_dtype of an element type
template <typename E> struct element_dtype;
template <> struct element_dtype<std::complex<double>> { static constexpr int value = QIP_C64; };
template <> struct element_dtype<std::complex<float>> { static constexpr int value = QIP_C32; };
template <> struct element_dtype<double> { static constexpr int value = QIP_F64; };
template <> struct element_dtype<float> { static constexpr int value = QIP_F32; };
template <> struct element_dtype<int64_t> { static constexpr int value = QIP_I64; };
template <> struct element_dtype<int32_t> { static constexpr int value = QIP_I32; };

/// get_index (matrix_ops.rs:33-35)
template <typename P> size_t get_index(const MatrixOp<P>& op, size_t i) { return op.indices()[i]; }

// ---- validated constructors (qip/src/state_ops/matrix_ops.rs:12-122) -------------------------------
template <typename P>
MatrixOp<P> make_matrix_op(std::vector<size_t> indices, std::vector<std::complex<P>> dat) {
  const size_t n = indices.size();
  if (indices.empty()) throw CircuitError("Must supply at least one op index");
  if (dat.size() != (size_t(1) << (2 * n)))
    throw CircuitError("Matrix data has " + std::to_string(dat.size()) + " entries versus expected 2^2*" +
                       std::to_string(n));
  return MatrixOp<P>::new_matrix(std::move(indices), std::move(dat));
}

template <typename P>
MatrixOp<P> make_sparse_matrix_op(std::vector<size_t> indices, typename MatrixOp<P>::SparseRows dat,
                                  Representation order = Representation::BigEndian) {
  const size_t n = indices.size();
  if (indices.empty()) throw CircuitError("Must supply at least one op index");
  if (dat.size() != (size_t(1) << n))
    throw CircuitError("Sparse matrix has " + std::to_string(dat.size()) + " rows versus expected 2^" +
                       std::to_string(n));
  for (size_t r = 0; r < dat.size(); ++r)
    if (dat[r].empty())
      throw CircuitError("All rows of sparse matrix must have data (" + std::to_string(r) + " is empty)");
  if (order == Representation::LittleEndian) {  // :62-77
    typename MatrixOp<P>::SparseRows out(dat.size());
    for (size_t r = 0; r < dat.size(); ++r) {
      for (auto& e : dat[r]) e.first = flip_bits(n, e.first);
      out[flip_bits(n, r)] = std::move(dat[r]);
    }
    dat = std::move(out);
  }
  return MatrixOp<P>::new_sparse(std::move(indices), std::move(dat));
}

template <typename P>
MatrixOp<P> make_swap_op(std::vector<size_t> a_indices, const std::vector<size_t>& b_indices) {
  if (a_indices.empty() || b_indices.empty()) throw CircuitError("Need at least 1 swap index for a and b");
  if (a_indices.size() != b_indices.size())
    throw CircuitError("Swap must be performed on two sets of indices of equal length, found " +
                       std::to_string(a_indices.size()) + " vs " + std::to_string(b_indices.size()));
  return MatrixOp<P>::new_swap(std::move(a_indices), b_indices);
}

template <typename P> MatrixOp<P> make_control_op(std::vector<size_t> c_indices, MatrixOp<P> op) {
  if (c_indices.empty()) throw CircuitError("Must supply at least one control index");
  if (op.kind() == MatrixOp<P>::Kind::Control) {  // collapse (:112-115)
    const size_t nc = c_indices.size() + op.n_controls();
    std::vector<size_t> all = std::move(c_indices);
    all.insert(all.end(), op.indices().begin(), op.indices().end());
    std::vector<size_t> ctrl(all.begin(), all.begin() + nc), rest(all.begin() + nc, all.end());
    return MatrixOp<P>::new_control(std::move(ctrl), rest, *op.inner());
  }
  std::vector<size_t> r = op.indices();
  return MatrixOp<P>::new_control(std::move(c_indices), r, std::move(op));
}

// ---- inner seam: apply_op / apply_op_overwrite (matrix_ops.rs:98-152) --------------------------------
template <typename P>
void apply_op(size_t n, const MatrixOp<P>& op, const std::vector<std::complex<P>>& input,
              std::vector<std::complex<P>>& output, size_t input_offset, size_t output_offset) {
  auto c = op.to_c();
  check(qip_hip_apply_op_host(dtype_of<P>::value, (uint32_t)n, &c->op, input.data(), input.size(), output.data(),
                              output.size(), input_offset, output_offset, 1));
}
template <typename P>
void apply_op_overwrite(size_t n, const MatrixOp<P>& op, const std::vector<std::complex<P>>& input,
                        std::vector<std::complex<P>>& output, size_t input_offset, size_t output_offset) {
  auto c = op.to_c();
  check(qip_hip_apply_op_host(dtype_of<P>::value, (uint32_t)n, &c->op, input.data(), input.size(), output.data(),
                              output.size(), input_offset, output_offset, 0));
}

// ---- qip_iterators::matrix_ops for ANY element type (matrix_ops.rs:38-59,98-152 are generic over P) ---------------------
// `qip::iterators::apply_op(n, op, input, output, input_offset, output_offset)` is the reference's signature with
// &[P] / &mut [P] as std::vector<E>: E = std::complex<double / float>, double, float, int64_t, int32_t.  Integer
// arithmetic wraps.  Host vectors (the call uploads, runs the HIP kernels, downloads); `apply_op_device` takes device
// pointers and a hipStream_t and copies nothing.
namespace iterators {
template <typename E>
void apply_op(size_t n, const BasicMatrixOp<E>& op, const std::vector<E>& input, std::vector<E>& output, size_t input_offset,
              size_t output_offset) {
  auto c = op.to_c();
  check(qip_hip_apply_op_host(element_dtype<E>::value, (uint32_t)n, &c->op, input.data(), input.size(), output.data(),
                              output.size(), input_offset, output_offset, 1));
}
template <typename E>
void apply_op_overwrite(size_t n, const BasicMatrixOp<E>& op, const std::vector<E>& input, std::vector<E>& output,
                        size_t input_offset, size_t output_offset) {
  auto c = op.to_c();
  check(qip_hip_apply_op_host(element_dtype<E>::value, (uint32_t)n, &c->op, input.data(), input.size(), output.data(),
                              output.size(), input_offset, output_offset, 0));
}
template <typename E>
E apply_op_row(size_t n, const BasicMatrixOp<E>& op, const std::vector<E>& input, size_t outputrow, size_t input_offset,
               size_t output_offset) {
  auto c = op.to_c();
  E value{};
  check(qip_hip_apply_op_row_host(element_dtype<E>::value, (uint32_t)n, &c->op, input.data(), input.size(), outputrow,
                                  input_offset, output_offset, &value));
  return value;
}
/// device slices: d_in / d_out point at in_len / out_len elements on `device`; asynchronous on `stream` for a dense op on
/// <= 4 qubits or a Swap (include/qip_hip.h: qip_hip_apply_op_device)
template <typename E>
void apply_op_device(size_t n, const BasicMatrixOp<E>& op, const E* d_in, size_t in_len, E* d_out, size_t out_len,
                     size_t input_offset, size_t output_offset, bool accumulate = true, int device = 0, void* stream = nullptr) {
  auto c = op.to_c();
  check(qip_hip_apply_op_device(element_dtype<E>::value, device, stream, (uint32_t)n, &c->op, d_in, in_len, d_out, out_len,
                                input_offset, output_offset, accumulate ? 1 : 0));
}
}  // namespace iterators

// ---- outer seam: device-resident state ---------------------------------------------------------------
template <typename P> class HipState {
 public:
  using C = std::complex<P>;
  explicit HipState(size_t n, int device = 0) : n_(n) {
    check(qip_hip_state_create((uint32_t)n, dtype_of<P>::value, device, &h_));
  }
  ~HipState() { qip_hip_state_destroy(h_); }
  HipState(const HipState&) = delete;
  HipState& operator=(const HipState&) = delete;

  size_t n() const { return n_; }
  void init_basis(size_t index) { check(qip_hip_state_init_basis(h_, index)); }
  void upload(const std::vector<C>& v, size_t offset = 0) { check(qip_hip_state_upload(h_, v.data(), offset, v.size())); }
  std::vector<C> download() const {
    std::vector<C> out(size_t(1) << n_);
    check(qip_hip_state_download(h_, out.data(), 0, out.size()));
    return out;
  }
  void apply_op(const MatrixOp<P>& op) {
    auto c = op.to_c();
    check(qip_hip_state_apply_op(h_, &c->op));
  }
  /// A run of gates in one call, so the library may schedule them (options "tile", "fuse").
  void apply_ops(const std::vector<MatrixOp<P>>& ops) {
    std::vector<std::unique_ptr<typename MatrixOp<P>::CView>> views;
    std::vector<qip_op> flat;
    views.reserve(ops.size());
    flat.reserve(ops.size());
    for (const auto& op : ops) {
      views.push_back(op.to_c());
      flat.push_back(views.back()->op);
    }
    check(qip_hip_state_apply_ops(h_, flat.data(), flat.size()));
  }
  void set_option(const char* key, int64_t value) { check(qip_hip_state_set_option(h_, key, value)); }
  // new[j] = old[src(j)], bit pi[d] of src(j) = bit d of j: any permutation of the index bits in one sweep (what a run of
  // Swap ops composes to, qubit_iterators.rs:208-218)
  void permute_bits(const std::vector<uint32_t>& pi) {
    if (pi.size() != n_) throw CircuitError("the permutation must list all n index bits");
    check(qip_hip_state_permute_bits(h_, pi.data()));
  }
  void sync() { check(qip_hip_state_sync(h_)); }
  /// two states side by side (validation support): device copy, and max |a_i - b_i| + the number of amplitudes that are not
  /// IEEE-equal over the whole vector
  void copy_from(HipState& other) { check(qip_hip_state_copy_from(h_, other.h_)); }
  std::pair<double, uint64_t> max_abs_diff(HipState& other) {
    double worst = 0;
    uint64_t differ = 0;
    check(qip_hip_state_max_abs_diff(h_, other.h_, &worst, &differ));
    return {worst, differ};
  }
  /// amplitudes at an explicit list of indices (one gather kernel: a logical window of a sharded / relabelled state)
  std::vector<std::complex<P>> download_indices(const std::vector<uint64_t>& idx) {
    std::vector<std::complex<P>> out(idx.size());
    check(qip_hip_state_download_indices(h_, idx.data(), idx.size(), out.data()));
    return out;
  }
  double norm_sqr() const {
    double v = 0;
    check(qip_hip_state_norm_sqr(h_, &v));
    return v;
  }
  /// measure_prob (measurement_ops.rs:44-112): probability of outcome `measured` on `indices`.
  double measure_prob(size_t measured, const std::vector<size_t>& indices) const {
    std::vector<uint64_t> idx(indices.begin(), indices.end());
    double v = 0;
    check(qip_hip_state_measure_prob(h_, measured, idx.data(), (uint32_t)idx.size(), &v));
    return v;
  }
  /// soft_measure (measurement_ops.rs:153-176) with the caller's uniform sample.
  size_t soft_measure(const std::vector<size_t>& indices, double rand_u01) const {
    std::vector<uint64_t> idx(indices.begin(), indices.end());
    uint64_t m = 0;
    check(qip_hip_state_soft_measure(h_, idx.data(), (uint32_t)idx.size(), rand_u01, &m));
    return size_t(m);
  }
  std::vector<double> measure_probs(const std::vector<size_t>& indices) const {
    std::vector<uint64_t> idx(indices.begin(), indices.end());
    std::vector<double> out(size_t(1) << idx.size());
    check(qip_hip_state_measure_probs(h_, idx.data(), (uint32_t)idx.size(), out.data()));
    return out;
  }
  /// measure (measurement_ops.rs:190-214); forced < 0 samples with rand_u01.
  std::pair<size_t, double> measure(const std::vector<size_t>& indices, int64_t forced, double rand_u01) {
    std::vector<uint64_t> idx(indices.begin(), indices.end());
    uint64_t m = 0;
    double p = 0;
    check(qip_hip_state_measure(h_, idx.data(), (uint32_t)idx.size(), forced, rand_u01, &m, &p));
    return {size_t(m), p};
  }
  qip_hip_state* handle() { return h_; }

 private:
  size_t n_;
  qip_hip_state* h_ = nullptr;
};

// ---- the state sharded over several GPUs: one process per GPU (include/qip_hip.h, qip_hip_dist_*) --------------
// Rank 0 calls unique_id() and hands the bytes to the other ranks over whatever the launcher offers; every rank then
// constructs DistState with the same id and issues the same calls in the same order.
template <typename P> class DistState {
 public:
  using C = std::complex<P>;
  static std::vector<unsigned char> unique_id() {
    std::vector<unsigned char> id(QIP_HIP_UNIQUE_ID_BYTES);
    check(qip_hip_dist_unique_id(id.data()));
    return id;
  }
  /// Built-in RCCL transport (`id` from unique_id()).
  DistState(size_t n, int device, int rank, int world, const std::vector<unsigned char>& id) : n_(n), rank_(rank), world_(world) {
    check(qip_hip_dist_create((uint32_t)n, dtype_of<P>::value, device, rank, world, id.data(), nullptr, &h_));
  }
  /// Caller-supplied transport (tests; hosts that already own a communicator).
  DistState(size_t n, int device, int rank, int world, const qip_hip_transport& t) : n_(n), rank_(rank), world_(world) {
    check(qip_hip_dist_create((uint32_t)n, dtype_of<P>::value, device, rank, world, nullptr, &t, &h_));
  }
  ~DistState() { qip_hip_dist_destroy(h_); }
  DistState(const DistState&) = delete;
  DistState& operator=(const DistState&) = delete;

  void init_basis(size_t logical_index) { check(qip_hip_dist_init_basis(h_, logical_index)); }
  void apply_op(const MatrixOp<P>& op) {
    auto c = op.to_c();
    check(qip_hip_dist_apply_op(h_, &c->op));
  }
  void apply_ops(const std::vector<MatrixOp<P>>& ops) {
    std::vector<std::unique_ptr<typename MatrixOp<P>::CView>> views;
    std::vector<qip_op> flat;
    for (const auto& op : ops) {
      views.push_back(op.to_c());
      flat.push_back(views.back()->op);
    }
    check(qip_hip_dist_apply_ops(h_, flat.data(), flat.size()));
  }
  void set_option(const char* key, int64_t value) { check(qip_hip_dist_set_option(h_, key, value)); }
  void sync() { check(qip_hip_dist_sync(h_)); }
  double norm_sqr() {
    double v = 0;
    check(qip_hip_dist_norm_sqr(h_, &v));
    return v;
  }
  std::vector<double> measure_probs(const std::vector<size_t>& indices) {
    std::vector<uint64_t> idx(indices.begin(), indices.end());
    std::vector<double> out(size_t(1) << idx.size());
    check(qip_hip_dist_measure_probs(h_, idx.data(), (uint32_t)idx.size(), out.data()));
    return out;
  }
  std::pair<size_t, double> measure(const std::vector<size_t>& indices, int64_t forced, double rand_u01) {
    std::vector<uint64_t> idx(indices.begin(), indices.end());
    uint64_t m = 0;
    double p = 0;
    check(qip_hip_dist_measure(h_, idx.data(), (uint32_t)idx.size(), forced, rand_u01, &m, &p));
    return {size_t(m), p};
  }
  /// phys[p] = physical bit position of logical bit position p (= n-1-qubit)
  std::vector<uint32_t> layout() {
    std::vector<uint32_t> phys(n_);
    check(qip_hip_dist_layout(h_, phys.data()));
    return phys;
  }
  /// pending rank renamings: this rank holds the amplitudes whose rank bits read rank ^ rank_flip()
  uint32_t rank_flip() {
    uint32_t m = 0;
    check(qip_hip_dist_rank_flip(h_, &m));
    return m;
  }
  /// this rank's amplitudes, in local (physical) order
  std::vector<C> download_shard(size_t n_local) {
    qip_hip_state* sh = nullptr;
    check(qip_hip_dist_local_state(h_, &sh));
    std::vector<C> out(size_t(1) << n_local);
    check(qip_hip_state_download(sh, out.data(), 0, out.size()));
    return out;
  }
  /// logical (reference-order) index of every amplitude of this rank's shard, in the shard's local order: what a host
  /// needs to gather a vector.  The layout changes with every exchange AND with every uncontrolled Swap (a relabelling),
  /// also at world = 1.
  std::vector<uint64_t> shard_logical_indices() {
    uint32_t g = 0;
    while ((1 << g) < world_) ++g;
    const size_t L = n_ - g;
    const std::vector<uint32_t> phys = layout();
    const uint64_t top = (uint64_t)((uint32_t)rank_ ^ rank_flip()) << L;
    std::vector<uint64_t> out(size_t(1) << L);
    for (uint64_t loc = 0; loc < out.size(); ++loc) {
      const uint64_t P64 = loc | top;
      uint64_t idx = 0;
      for (size_t p = 0; p < n_; ++p) idx |= ((P64 >> phys[p]) & 1ull) << p;
      out[loc] = idx;
    }
    return out;
  }
  /// soft_measure (measurement_ops.rs:153-176) of the sharded state: rank 0's sample decides; no collapse
  size_t soft_measure(const std::vector<size_t>& indices, double rand_u01) {
    std::vector<uint64_t> idx(indices.begin(), indices.end());
    uint64_t m = 0;
    check(qip_hip_dist_soft_measure(h_, idx.data(), (uint32_t)idx.size(), rand_u01, &m));
    return (size_t)m;
  }
  qip_hip_dist_stats take_stats() {
    qip_hip_dist_stats st;
    check(qip_hip_dist_take_stats(h_, &st));
    return st;
  }

 private:
  size_t n_;
  int rank_ = 0, world_ = 1;
  qip_hip_dist* h_ = nullptr;
};

// ---- LocalBuilder<P>'s recording and run loop (qip/src/builder.rs:325-519) ----------------------------
struct Register {
  std::vector<size_t> indices;
  size_t n() const { return indices.size(); }
};

struct MeasurementResult {
  bool stochastic = false;
  size_t measured = 0;
  double prob = 0;
  std::vector<double> probs;
};

template <typename P> class HipBuilder {
 public:
  using C = std::complex<P>;
  enum class Obj { X, Y, Z, H, S, T, CNOT, SWAP, Rz, MAT, GlobalPhase, Measurement, StochasticMeasurement };
  struct Entry {
    std::vector<size_t> indices;
    Obj obj;
    P theta = 0;
    std::vector<C> mat;
  };

  size_t n() const { return n_; }
  Register register_(size_t n) {
    if (n == 0) throw CircuitError("register size must be non-zero");
    Register r;
    for (size_t i = 0; i < n; ++i) r.indices.push_back(n_ + i);
    n_ += n;
    return r;
  }
  Register qubit() { return register_(1); }

  // CliffordTBuilder (a 1-qubit object on a wider register is broadcast, builder.rs:382-387)
  Register x(Register r) { return apply1(std::move(r), Obj::X); }
  Register y(Register r) { return apply1(std::move(r), Obj::Y); }
  Register z(Register r) { return apply1(std::move(r), Obj::Z); }
  Register h(Register r) { return apply1(std::move(r), Obj::H); }
  Register s(Register r) { return apply1(std::move(r), Obj::S); }
  Register t(Register r) { return apply1(std::move(r), Obj::T); }
  Register s_dagger(Register r) { return s(z(std::move(r))); }         // builder_traits.rs:419-422
  Register t_dagger(Register r) { return t(s_dagger(std::move(r))); }  // :408-411
  Register rz(Register r, P theta) {
    for (size_t q : r.indices) pipeline_.push_back({{q}, Obj::Rz, theta, {}});
    return r;
  }
  std::pair<Register, Register> cnot(Register cr, Register r) {  // :425-451
    if (cr.n() > 1) throw CircuitError("Clifford CNOT can only have a single control qubit.");
    for (size_t q : r.indices) pipeline_.push_back({{cr.indices[0], q}, Obj::CNOT, 0, {}});
    return {std::move(cr), std::move(r)};
  }
  std::pair<Register, Register> swap(Register ra, Register rb) {  // :454-482: three CNOTs per pair
    if (ra.n() != rb.n()) throw CircuitError("Swap must be between registers of the same size.");
    for (size_t i = 0; i < ra.n(); ++i) {
      Register a{{ra.indices[i]}}, b{{rb.indices[i]}};
      cnot(a, b);
      cnot(b, a);
      cnot(a, b);
    }
    return {std::move(ra), std::move(rb)};
  }
  std::pair<Register, Register> swap_op(Register ra, Register rb) {  // UnitaryMatrixObject::SWAP
    if (ra.n() != rb.n()) throw CircuitError("Swap must be between registers of the same size.");
    std::vector<size_t> idx = ra.indices;
    idx.insert(idx.end(), rb.indices.begin(), rb.indices.end());
    pipeline_.push_back({idx, Obj::SWAP, 0, {}});
    return {std::move(ra), std::move(rb)};
  }
  Register apply_matrix(Register r, std::vector<C> data) {  // UnitaryBuilder::apply_vec_matrix
    if (data.size() != (size_t(1) << (2 * r.n()))) throw CircuitError("Matrix has incorrect N and cannot be broadcast");
    pipeline_.push_back({r.indices, Obj::MAT, 0, std::move(data)});
    return r;
  }
  std::pair<Register, size_t> measure(Register r) {
    pipeline_.push_back({r.indices, Obj::Measurement, 0, {}});
    return {std::move(r), n_measurements_++};
  }
  std::pair<Register, size_t> measure_stochastic(Register r) {
    pipeline_.push_back({r.indices, Obj::StochasticMeasurement, 0, {}});
    return {std::move(r), n_measurements_++};
  }
  const std::vector<Entry>& pipeline() const { return pipeline_; }

  /// The lowering table of the reference run loop (builder.rs:436-498).
  static MatrixOp<P> lower(const Entry& e) {
    const C l(1, 0), o(0, 0), i(0, 1);
    switch (e.obj) {
      case Obj::X: return make_matrix_op<P>(e.indices, {o, l, l, o});
      case Obj::Y: return make_matrix_op<P>(e.indices, {o, -i, i, o});
      case Obj::Z: return make_matrix_op<P>(e.indices, {l, o, o, -l});
      case Obj::H: {
        const C nl = C(1, 0) * P(0.70710678118654752440);  // Complex::one() * FRAC_1_SQRT_2
        return make_matrix_op<P>(e.indices, {nl, nl, nl, -nl});
      }
      case Obj::S: return make_matrix_op<P>(e.indices, {l, o, o, i});
      case Obj::T: return make_matrix_op<P>(e.indices, {l, o, o, std::polar(P(1), P(0.78539816339744830962))});
      case Obj::CNOT: {
        std::vector<size_t> rest(e.indices.begin() + 1, e.indices.end());
        return make_control_op<P>({e.indices[0]}, make_matrix_op<P>(rest, {o, l, l, o}));
      }
      case Obj::MAT: return make_matrix_op<P>(e.indices, e.mat);
      case Obj::SWAP: {
        const size_t x = e.indices.size() / 2;
        std::vector<size_t> a(e.indices.begin(), e.indices.begin() + x), b(e.indices.begin() + x, e.indices.end());
        return make_swap_op<P>(a, b);
      }
      case Obj::Rz: {
        const P h_theta = e.theta * P(0.5);
        return make_matrix_op<P>(e.indices, {std::polar(P(1), -h_theta), o, o, std::polar(P(1), h_theta)});
      }
      default: throw CircuitError("cannot lower this pipeline entry");
    }
  }

  /// calculate_state_with_init (builder.rs:400-519).  `rand_u01` supplies the uniform samples for the
  /// collapsing measurements in order (the reference draws rand::random::<f64>() itself, :160);
  /// `forced` (if non-empty) plays MeasuredCondition instead.  The stale-buffer quirk of the reference
  ///  is deliberately not reproduced.
  std::pair<std::vector<C>, std::vector<MeasurementResult>> calculate_state_with_init(
      const std::vector<std::pair<const Register*, size_t>>& init, std::vector<double> rand_u01 = {},
      std::vector<int64_t> forced = {}) {
    if (n_ == 0) throw CircuitError("empty circuit");
    size_t index = 0;
    for (const auto& rx : init)  // :409-421: bit k of x sets qubit r.indices[k]
      for (size_t k = 0; k < rx.first->indices.size(); ++k)
        index |= ((rx.second >> k) & 1) << (n_ - 1 - rx.first->indices[k]);
    HipState<P> st(n_);
    st.init_basis(index);
    std::vector<MeasurementResult> results;
    size_t mi = 0;
    for (const auto& e : pipeline_) {
      if (e.obj == Obj::GlobalPhase) continue;
      if (e.obj == Obj::Measurement) {
        const int64_t f = mi < forced.size() ? forced[mi] : -1;
        const double r = mi < rand_u01.size() ? rand_u01[mi] : 0.5;
        ++mi;
        auto mp = st.measure(e.indices, f, r);
        MeasurementResult res;
        res.measured = mp.first;
        res.prob = mp.second;
        results.push_back(res);
      } else if (e.obj == Obj::StochasticMeasurement) {
        MeasurementResult res;
        res.stochastic = true;
        res.probs = st.measure_probs(e.indices);
        results.push_back(res);
      } else {
        st.apply_op(lower(e));
      }
    }
    return {st.download(), results};
  }

 private:
  Register apply1(Register r, Obj o) {
    for (size_t q : r.indices) pipeline_.push_back({{q}, o, 0, {}});
    return r;
  }
  size_t n_ = 0, n_measurements_ = 0;
  std::vector<Entry> pipeline_;
};

}  // namespace qip
