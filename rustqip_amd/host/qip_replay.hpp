// qip_replay.hpp — C++ reader for the flat circuit-replay format "qipc 1" (SURVEY.md §8 row f2).
//
// The format is specified in rustqip_amd/replay.py (the Python twin reads and writes it); the Rust writer a
// RustQIP program would use is bindings/rust/qip-hip/src/replay.rs.  A file holds what
// LocalBuilder::calculate_state_with_init would have executed: the MatrixOps its run loop lowers every pipeline
// entry to (qip/src/builder.rs:436-498) and the measurement stages between them (:501-511).
// Header-only; depends on qip_hip.hpp (the host mirror) only.
#pragma once
#include <cstdlib>
#include <istream>
#include <sstream>
#include <string>
#include <variant>
#include <vector>

#include "qip_hip.hpp"

namespace qip {
namespace replay {

struct Measure {
  std::vector<size_t> indices;
  double rand_u01 = 0;
};
struct Probs {
  std::vector<size_t> indices;
};
template <typename P> using Item = std::variant<MatrixOp<P>, Measure, Probs>;

template <typename P> struct Circuit {
  size_t n = 0;
  size_t init = 0;
  std::vector<Item<P>> items;
};

namespace detail {
struct Tokens {
  std::vector<std::string> t;
  size_t i = 0;
  size_t lineno = 0;
  [[noreturn]] void fail(const std::string& what) const {
    throw CircuitError("line " + std::to_string(lineno) + ": " + what);
  }
  const std::string& take() {
    if (i >= t.size()) fail("unexpected end of statement");
    return t[i++];
  }
  size_t uint() {
    const std::string& s = take();
    if (s.empty() || s.find_first_not_of("0123456789") != std::string::npos)
      fail("expected a non-negative integer, found '" + s + "'");
    return (size_t)std::strtoull(s.c_str(), nullptr, 10);
  }
  double num() {
    const std::string& s = take();
    char* end = nullptr;
    const double v = std::strtod(s.c_str(), &end);  // correctly rounded: shortest-repr decimals survive exactly
    if (end == s.c_str() || *end != '\0') fail("expected a number, found '" + s + "'");
    return v;
  }
  void done() const {
    if (i != t.size()) fail(std::to_string(t.size() - i) + " unexpected trailing token(s)");
  }
};

template <typename P> MatrixOp<P> parse_op(Tokens& tk, const std::string& word) {
  using C = std::complex<P>;
  if (word == "matrix") {
    const size_t k = tk.uint();
    if (k > 15) tk.fail("matrix on more than 15 qubits");
    std::vector<size_t> idx(k);
    for (auto& v : idx) v = tk.uint();
    std::vector<C> dat(size_t(1) << (2 * k));
    for (auto& z : dat) {
      const double re = tk.num(), im = tk.num();
      z = C((P)re, (P)im);
    }
    return make_matrix_op<P>(std::move(idx), std::move(dat));
  }
  if (word == "sparse") {
    const size_t k = tk.uint();
    if (k > 30) tk.fail("sparse matrix on more than 30 qubits");
    std::vector<size_t> idx(k);
    for (auto& v : idx) v = tk.uint();
    typename MatrixOp<P>::SparseRows rows(size_t(1) << k);
    for (auto& row : rows) {
      const size_t nnz = tk.uint();
      for (size_t e = 0; e < nnz; ++e) {
        const size_t col = tk.uint();
        const double re = tk.num(), im = tk.num();
        row.emplace_back(col, C((P)re, (P)im));
      }
    }
    return make_sparse_matrix_op<P>(std::move(idx), std::move(rows));
  }
  if (word == "swap") {
    const size_t h = tk.uint();
    std::vector<size_t> a(h), b(h);
    for (auto& v : a) v = tk.uint();
    for (auto& v : b) v = tk.uint();
    return make_swap_op<P>(std::move(a), b);
  }
  if (word == "control") {
    const size_t nc = tk.uint();
    std::vector<size_t> c(nc);
    for (auto& v : c) v = tk.uint();
    const std::string inner = tk.take();
    return make_control_op<P>(std::move(c), parse_op<P>(tk, inner));
  }
  tk.fail("unknown statement '" + word + "'");
}
}  // namespace detail

template <typename P> Circuit<P> load(std::istream& in) {
  Circuit<P> circ;
  bool header = false, have_n = false;
  std::string line;
  size_t lineno = 0;
  while (std::getline(in, line)) {
    ++lineno;
    const size_t hash = line.find('#');
    if (hash != std::string::npos) line.resize(hash);
    detail::Tokens tk;
    tk.lineno = lineno;
    std::istringstream ls(line);
    for (std::string w; ls >> w;) tk.t.push_back(w);
    if (tk.t.empty()) continue;
    const std::string word = tk.take();
    if (!header) {
      if (word != "qipc" || tk.uint() != 1) tk.fail("expected the header 'qipc 1'");
      header = true;
    } else if (word == "n") {
      circ.n = tk.uint();
      have_n = true;
    } else if (!have_n) {
      tk.fail("'n <qubits>' must come before '" + word + "'");
    } else if (word == "init") {
      circ.init = tk.uint();
      if (circ.n < 64 && (circ.init >> circ.n)) tk.fail("init index does not fit the qubit count");
    } else if (word == "measure") {
      Measure m;
      m.indices.resize(tk.uint());
      for (auto& v : m.indices) v = tk.uint();
      m.rand_u01 = tk.num();
      circ.items.emplace_back(std::move(m));
    } else if (word == "probs") {
      Probs p;
      p.indices.resize(tk.uint());
      for (auto& v : p.indices) v = tk.uint();
      circ.items.emplace_back(std::move(p));
    } else {
      circ.items.emplace_back(detail::parse_op<P>(tk, word));
    }
    tk.done();
  }
  if (!have_n) throw CircuitError("no 'n <qubits>' statement");
  return circ;
}

/// What one measurement statement produced, in statement order.
struct Result {
  bool stochastic = false;
  size_t measured = 0;
  double prob = 0;
  std::vector<double> probs;
};

/// Replay on the GPU: runs of gates go through qip_hip_state_apply_ops (so option "tile" applies), measurement
/// statements in between.  The state is left holding the final amplitudes.
template <typename P> std::vector<Result> run(const Circuit<P>& circ, HipState<P>& st) {
  std::vector<Result> out;
  std::vector<std::unique_ptr<typename MatrixOp<P>::CView>> views;
  std::vector<qip_op> batch;
  auto flush = [&] {
    if (!batch.empty()) check(qip_hip_state_apply_ops(st.handle(), batch.data(), batch.size()));
    batch.clear();
    views.clear();
  };
  st.init_basis(circ.init);
  for (const auto& it : circ.items) {
    if (const auto* op = std::get_if<MatrixOp<P>>(&it)) {
      views.push_back(op->to_c());
      batch.push_back(views.back()->op);
    } else if (const auto* m = std::get_if<Measure>(&it)) {
      flush();
      Result r;
      std::tie(r.measured, r.prob) = st.measure(m->indices, -1, m->rand_u01);
      out.push_back(std::move(r));
    } else {
      flush();
      Result r;
      r.stochastic = true;
      r.probs = st.measure_probs(std::get<Probs>(it).indices);
      out.push_back(std::move(r));
    }
  }
  flush();
  return out;
}

}  // namespace replay
}  // namespace qip
