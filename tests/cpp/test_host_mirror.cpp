// C++ host mirror test (rustqip_amd/host/qip_hip.hpp).  Reads like the reference's own unit tests:
//   qip/src/state_ops/matrix_ops.rs:276-377 (constructors, apply_op on Complex<f64>)
//   qip-iterators/src/matrix_ops.rs:350-374 (index order)
// Usage: test_host_mirror cpu   -> host-only checks (constructors, errors, marshalling, validator)
//        test_host_mirror gpu   -> also runs the HIP path and compares with the CPU oracle (linked
//                                  in as the checker: oracle/libqip_oracle.so)
#include <climits>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>

#include "qip_hip.hpp"

extern "C" void qip_oracle_apply_op_c64(uint32_t n, const qip_op* op, const qip_c64* input, uint64_t in_len,
                                        qip_c64* output, uint64_t out_len, uint64_t in_off, uint64_t out_off,
                                        int accumulate, int nthreads);

extern "C" void qip_oracle_apply_op_f64(uint32_t n, const qip_op* op, const double* input, uint64_t in_len, double* output,
                                        uint64_t out_len, uint64_t in_off, uint64_t out_off, int accumulate, int nthreads);
extern "C" void qip_oracle_apply_op_i32(uint32_t n, const qip_op* op, const int32_t* input, uint64_t in_len, int32_t* output,
                                        uint64_t out_len, uint64_t in_off, uint64_t out_off, int accumulate, int nthreads);

using qip::CircuitError;
using C = std::complex<double>;
using Op = qip::MatrixOp<double>;

static int failures = 0;
#define EXPECT(cond)                                                   \
  do {                                                                 \
    if (!(cond)) {                                                     \
      std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond);      \
      ++failures;                                                      \
    }                                                                  \
  } while (0)

template <typename F> static bool throws(F&& f, const char* needle) {
  try {
    f();
  } catch (const CircuitError& e) {
    return std::strstr(e.what(), needle) != nullptr;
  }
  return false;
}

static std::vector<C> from_reals(std::initializer_list<double> v) {
  std::vector<C> out;
  for (double x : v) out.emplace_back(x, 0.0);
  return out;
}

static_assert(qip::element_dtype<std::complex<double>>::value == QIP_C64 && qip::element_dtype<double>::value == QIP_F64 &&
              qip::element_dtype<float>::value == QIP_F32 && qip::element_dtype<int64_t>::value == QIP_I64 &&
              qip::element_dtype<int32_t>::value == QIP_I32, "enum qip_dtype");

static void host_checks() {
  // test_get_index_simple / _condition / _swap
  auto op = Op::new_matrix({0, 1, 2}, {});
  EXPECT(op.num_indices() == 3 && qip::get_index(op, 0) == 0 && qip::get_index(op, 2) == 2);
  auto cop = qip::make_control_op<double>({0, 1}, Op::new_matrix({2, 3}, {}));
  EXPECT(cop.num_indices() == 4 && qip::get_index(cop, 3) == 3 && cop.n_controls() == 2);
  auto sw = Op::new_swap({0, 1}, {2, 3});
  EXPECT(sw.num_indices() == 4 && qip::get_index(sw, 2) == 2);
  // constructor errors
  EXPECT(throws([] { qip::make_matrix_op<double>({}, {}); }, "at least one op index"));
  EXPECT(throws([] { qip::make_matrix_op<double>({0}, from_reals({1, 0, 0})); }, "entries versus expected"));
  EXPECT(throws([] { qip::make_swap_op<double>({}, {1}); }, "at least 1 swap index"));
  EXPECT(throws([] { qip::make_swap_op<double>({0, 1}, {2}); }, "equal length"));
  EXPECT(throws([] { qip::make_control_op<double>({}, Op::new_matrix({0}, from_reals({0, 1, 1, 0}))); },
                "at least one control index"));
  EXPECT(throws([] { qip::make_sparse_matrix_op<double>({0}, {{{0, C(1)}}, {}}); }, "must have data"));
  // nested control collapse (matrix_ops.rs:112-115)
  auto inner = qip::make_control_op<double>({1}, qip::make_matrix_op<double>({2}, from_reals({0, 1, 1, 0})));
  auto outer = qip::make_control_op<double>({0}, inner);
  EXPECT(outer.n_controls() == 2 && outer.indices().size() == 3 && outer.inner()->kind() == Op::Kind::Matrix);
  // test_make_sparse_mat (matrix_ops.rs:346-377)
  Op::SparseRows expected = {{{1, C(1)}}, {{0, C(1)}}, {{3, C(1)}}, {{2, C(1)}}};
  auto s1 = qip::make_sparse_matrix_op<double>({0, 1}, expected, qip::Representation::BigEndian);
  auto s2 = qip::make_sparse_matrix_op<double>({0, 1}, {{{2, C(1)}}, {{3, C(1)}}, {{0, C(1)}}, {{1, C(1)}}},
                                               qip::Representation::LittleEndian);
  EXPECT(s1.rows() == expected && s2.rows() == expected);
  // marshalling + the C validator (no GPU needed)
  auto cv = outer.to_c();
  EXPECT(cv->op.kind == QIP_OP_CONTROL && cv->op.n_controls == 2 && cv->op.inner->kind == QIP_OP_MATRIX);
  EXPECT(qip_hip_validate_op(3, &cv->op) == QIP_OK);
  EXPECT(qip_hip_validate_op(2, &cv->op) != QIP_OK);  // index 2 out of range for n = 2
  // lowering table (builder.rs:436-498)
  qip::HipBuilder<double>::Entry e{{0}, qip::HipBuilder<double>::Obj::H, 0, {}};
  auto hop = qip::HipBuilder<double>::lower(e);
  EXPECT(hop.data()[0] == C(std::sqrt(0.5), 0) && hop.data()[3].real() == -std::sqrt(0.5) && std::signbit(hop.data()[3].imag()));
  // broadcast + initial index
  qip::HipBuilder<double> b;
  auto r = b.register_(3);
  b.h(r);
  EXPECT(b.pipeline().size() == 3 && b.pipeline()[1].indices[0] == 1);
}

// qip-iterators' own unit tests run apply_op on i32 (matrix_ops.rs:271-374), its benches on f64: the generic element type
static void generic_element_checks() {
  namespace it = qip::iterators;
  using IOp = qip::BasicMatrixOp<int32_t>;
  using ROp = qip::BasicMatrixOp<double>;
  {
    // test_counting (matrix_ops.rs:337-348): [1, 2, 3, 4] on qubit 0 of 3 = kron(mat, I4); make_op_matrix as the reference builds it
    const size_t n = 3, N = 8;
    const std::vector<int32_t> data{1, 2, 3, 4};
    auto op = IOp::new_matrix({0}, data);
    for (size_t i = 0; i < N; ++i) {
      std::vector<int32_t> in(N, 0), out(N, 0);
      in[i] = 1;
      it::apply_op(n, op, in, out, 0, 0);
      for (size_t r = 0; r < N; ++r) EXPECT(out[r] == ((r & 3) == (i & 3) ? data[(r >> 2) * 2 + (i >> 2)] : 0));
    }
    // test_counting_order / _flipped (:350-374)
    std::vector<int32_t> d16(16);
    for (int i = 0; i < 16; ++i) d16[i] = i;
    bool same01 = true, same10 = true;
    for (size_t i = 0; i < 4; ++i) {
      std::vector<int32_t> in(4, 0), a(4, 0), b(4, 0);
      in[i] = 1;
      it::apply_op(2, IOp::new_matrix({0, 1}, d16), in, a, 0, 0);
      it::apply_op(2, IOp::new_matrix({1, 0}, d16), in, b, 0, 0);
      for (size_t r = 0; r < 4; ++r) {
        same01 = same01 && a[r] == d16[r * 4 + i];
        same10 = same10 && b[r] == d16[r * 4 + i];
      }
    }
    EXPECT(same01 && !same10);
  }
  {
    // random real ops against the oracle's real restatement: whole vector (group kernel), a window (literal kernel), a row
    const size_t n = 10, N = size_t(1) << n;
    std::mt19937_64 rng(9);
    std::normal_distribution<double> g;
    std::vector<double> x(N), m4(16), m2(4);
    for (auto& v : x) v = g(rng);
    for (auto& v : m4) v = g(rng);
    for (auto& v : m2) v = g(rng);
    std::vector<ROp> ops{ROp::new_matrix({7, 2}, m4), ROp::new_control({9}, {0}, ROp::new_matrix({0}, m2)), ROp::new_swap({1}, {8}),
                         ROp::new_sparse({4}, {{{1, 2.5}, {0, -1.0}}, {{0, 0.5}}})};
    for (const auto& op : ops) {
      auto c = op.to_c();
      std::vector<double> got(N, 0.25), want(N, 0.25);
      it::apply_op(n, op, x, got, 0, 0);
      qip_oracle_apply_op_f64((uint32_t)n, &c->op, x.data(), N, want.data(), N, 0, 0, 1, 1);
      EXPECT(got == want);
      std::vector<double> xin(x.begin() + 100, x.begin() + 700), gw(300), ww(300);
      it::apply_op_overwrite(n, op, xin, gw, 100, 64);
      qip_oracle_apply_op_f64((uint32_t)n, &c->op, xin.data(), xin.size(), ww.data(), ww.size(), 100, 64, 0, 1);
      EXPECT(gw == ww);
      EXPECT(it::apply_op_row(n, op, xin, 17, 100, 64) == ww[17]);
    }
    // i32 wraps (two's complement), as the oracle's restatement does
    std::vector<int32_t> big{1 << 30, 2147483647}, out{0, 1}, want{0, 1};
    auto w = qip::BasicMatrixOp<int32_t>::new_matrix({0}, {4, 0, 0, 1});
    auto cw = w.to_c();
    it::apply_op(1, w, big, out, 0, 0);
    qip_oracle_apply_op_i32(1, &cw->op, big.data(), 2, want.data(), 2, 0, 0, 1, 1);
    EXPECT(out == want && out[0] == 0 && out[1] == INT32_MIN);
  }
  std::printf("generic element types: i32 reference vectors, f64 vs the oracle (whole vector, window, row)\n");
}

static void gpu_checks() {
  generic_element_checks();
  // test_apply_identity / test_apply_swap_mat / test_apply_swap_mat_first (matrix_ops.rs:306-344)
  {
    auto op = Op::new_matrix({0}, from_reals({1, 0, 0, 1}));
    auto in = from_reals({1, 0});
    auto out = from_reals({0, 0});
    qip::apply_op<double>(1, op, in, out, 0, 0);
    EXPECT(in == out);
    auto flip = Op::new_matrix({0}, from_reals({0, 1, 1, 0}));
    out = from_reals({0, 0});
    qip::apply_op<double>(1, flip, in, out, 0, 0);
    EXPECT(out == from_reals({0, 1}));
    auto in4 = from_reals({1, 0, 0, 0});
    auto out4 = from_reals({0, 0, 0, 0});
    qip::apply_op<double>(2, flip, in4, out4, 0, 0);
    EXPECT(out4 == from_reals({0, 0, 1, 0}));
    out4 = from_reals({0, 0, 0, 0});
    qip::apply_op<double>(2, Op::new_matrix({1}, from_reals({0, 1, 1, 0})), in4, out4, 0, 0);
    EXPECT(out4 == from_reals({0, 1, 0, 0}));
  }
  // test_counting_order (qip-iterators matrix_ops.rs:350-362): column c of the op matrix
  {
    std::vector<C> data;
    for (int i = 0; i < 16; ++i) data.emplace_back(i, 0);
    auto op = Op::new_matrix({0, 1}, data);
    for (size_t c = 0; c < 4; ++c) {
      std::vector<C> in(4), out(4);
      in[c] = 1;
      qip::apply_op<double>(2, op, in, out, 0, 0);
      for (size_t r = 0; r < 4; ++r) EXPECT(out[r] == data[r * 4 + c]);
    }
  }
  // a builder circuit vs the oracle, gate by gate, same descriptors
  {
    const size_t n = 10;
    qip::HipBuilder<double> b;
    auto ra = b.register_(5);
    auto rb = b.register_(5);
    b.h(ra);
    b.cnot(qip::Register{{ra.indices[0]}}, rb);
    b.t(rb);
    b.rz(ra, 0.37);
    b.swap_op(qip::Register{{0, 1}}, qip::Register{{8, 9}});
    b.y(qip::Register{{4}});
    std::mt19937_64 rng(7);
    std::normal_distribution<double> nd;
    std::vector<C> u(16);
    for (auto& v : u) v = C(nd(rng), nd(rng));
    b.apply_matrix(qip::Register{{3, 7}}, u);
    auto mh = b.measure_stochastic(qip::Register{{2, 6}});
    auto res = b.calculate_state_with_init({{&ra, 0b00101}, {&rb, 0b10000}});
    // oracle replay
    size_t index = 0;
    for (size_t k = 0; k < 5; ++k) index |= ((0b00101 >> k) & 1) << (n - 1 - ra.indices[k]);
    for (size_t k = 0; k < 5; ++k) index |= ((0b10000 >> k) & 1) << (n - 1 - rb.indices[k]);
    std::vector<C> st(size_t(1) << n), arena(size_t(1) << n);
    st[index] = 1;
    for (const auto& e : b.pipeline()) {
      if (e.obj == qip::HipBuilder<double>::Obj::StochasticMeasurement) continue;
      auto c = qip::HipBuilder<double>::lower(e).to_c();
      qip_oracle_apply_op_c64((uint32_t)n, &c->op, (const qip_c64*)st.data(), st.size(), (qip_c64*)arena.data(),
                              arena.size(), 0, 0, 0, 1);
      st.swap(arena);
    }
    double maxd = 0, psum = 0;
    for (size_t i = 0; i < st.size(); ++i) maxd = std::max(maxd, std::abs(st[i] - res.first[i]));
    EXPECT(maxd <= 1e-12);
    EXPECT(res.second.size() == 1 && res.second[mh.second].stochastic && res.second[0].probs.size() == 4);
    std::vector<double> want(4, 0.0);
    for (size_t i = 0; i < st.size(); ++i) {
      const size_t m = ((i >> (n - 1 - 2)) & 1) | (((i >> (n - 1 - 6)) & 1) << 1);
      want[m] += std::norm(st[i]);
    }
    for (int m = 0; m < 4; ++m) {
      if (!(std::abs(res.second[0].probs[m] - want[m]) < 1e-12 * (1.0 + want[m])))
        std::printf("  probs[%d] = %.17g, oracle-side sum = %.17g\n", m, res.second[0].probs[m], want[m]);
      EXPECT(std::abs(res.second[0].probs[m] - want[m]) < 1e-12 * (1.0 + want[m]));
      psum += res.second[0].probs[m];
    }
    std::printf("builder circuit: max|delta| vs oracle = %.3e, sum of probs = %.15f\n", maxd, psum);
  }
  // errors surface as CircuitError, not crashes
  EXPECT(throws([] {
    std::vector<C> in(4), out(4);
    qip::apply_op<double>(2, Op::new_matrix({2}, from_reals({0, 1, 1, 0})), in, out, 0, 0);
  }, "out of range"));
  {
    // the sharded state through the C ABI with the built-in RCCL transport, world = 1 (unique id, communicator,
    // all-reduce; more ranks need more GPUs): same circuit as a plain HipState, same amplitudes
    const size_t n = 10;
    const double s = std::sqrt(0.5);
    std::vector<Op> ops;
    for (size_t t = 0; t < n; ++t) ops.push_back(Op::new_matrix({t}, from_reals({s, s, s, -s})));
    ops.push_back(Op::new_control({0}, {9}, Op::new_matrix({9}, from_reals({0, 1, 1, 0}))));
    ops.push_back(Op::new_swap({1, 2}, {8, 7}));
    qip::HipState<double> ref(n);
    ref.init_basis(5);
    ref.apply_ops(ops);
    qip::DistState<double> ds(n, 0, 0, 1, qip::DistState<double>::unique_id());
    ds.init_basis(5);
    ds.apply_ops(ops);
    // (the uncontrolled swap is a relabelling on a sharded state: the shard is gathered through the layout, also at world 1)
    const auto a = ref.download(), b = ds.download_shard(n);
    const auto where = ds.shard_logical_indices();
    double maxd = 0;
    bool moved = false;
    for (size_t i = 0; i < b.size(); ++i) {
      maxd = std::max(maxd, std::abs(a[where[i]] - b[i]));
      moved = moved || where[i] != i;
    }
    EXPECT(maxd == 0.0);
    EXPECT(moved);
    EXPECT(std::abs(ds.norm_sqr() - 1.0) < 1e-12);
    const auto pr = ds.measure_probs({0, 9}), pw = ref.measure_probs({0, 9});
    for (int m = 0; m < 4; ++m) EXPECT(std::abs(pr[m] - pw[m]) < 1e-13);
    const qip_hip_dist_stats stats = ds.take_stats();
    EXPECT(ds.layout().size() == n && stats.remaps == 0);
    EXPECT(stats.rccl_ranks == 1 && stats.rccl_rank == 0 && stats.piece_bytes == (1ull << 30));  // read back from the communicator
    EXPECT(ds.soft_measure({0, 9}, 0.3) == ref.soft_measure({0, 9}, 0.3));
    {
      qip::HipState<double> twin(n);
      twin.copy_from(ref);
      EXPECT(twin.max_abs_diff(ref) == std::make_pair(0.0, (uint64_t)0));
      twin.apply_ops({ops[3]});
      EXPECT(twin.max_abs_diff(ref).second > 0);
    }
    std::printf("sharded state (world 1, RCCL transport): max|delta| vs single-GPU state = %.1e\n", maxd);
    // a run of Swap ops composes to one bit permutation: [1,2] <-> [8,7] exchanges qubits 1/8 and 2/7 = index bits 8/1 and 7/2
    std::vector<uint32_t> pi(n);
    for (uint32_t d = 0; d < n; ++d) pi[d] = d;
    std::swap(pi[8], pi[1]);
    std::swap(pi[7], pi[2]);
    ref.permute_bits(pi);           // undoes the circuit's closing swap ...
    ref.apply_ops({ops.back()});    // ... and the Swap op puts it back
    const auto c = ref.download();
    double maxp = 0;
    for (size_t i = 0; i < a.size(); ++i) maxp = std::max(maxp, std::abs(a[i] - c[i]));
    EXPECT(maxp == 0.0);
    EXPECT(throws([&] { ref.permute_bits({0, 1}); }, "all n index bits"));
  }
}

int main(int argc, char** argv) {
  const bool gpu = argc > 1 && !std::strcmp(argv[1], "gpu");
  host_checks();
  if (gpu) {
    if (qip_hip_device_count() < 1) {
      std::printf("FAIL: gpu mode without a HIP device\n");
      return 2;
    }
    gpu_checks();
  } else {
    // without a device the compute entry points must fail loudly
    if (qip_hip_device_count() == 0)
      EXPECT(throws([] { qip::HipState<double> st(3); }, "no CPU fallback"));
  }
  std::printf("%s (%d failures)\n", failures ? "FAILED" : "PASSED", failures);
  return failures ? 1 : 0;
}
