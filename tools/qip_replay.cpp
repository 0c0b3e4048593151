// qip_replay — run a "qipc 1" circuit file (rustqip_amd/replay.py, rustqip_amd/host/qip_replay.hpp) on the GPU.
//
//   qip_replay [--tile 0|1|2] [--amps K] circuit.qipc
//
// Prints one line per measurement statement, the squared norm, the first K amplitudes (default 0) and the time
// of the replay.  Build: make -C tools qip_replay   (g++; links rustqip_amd/lib/libqip_hip.so)
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <tuple>

#include "qip_replay.hpp"

int main(int argc, char** argv) {
  int64_t tile = 1;
  size_t amps = 0;
  const char* path = nullptr;
  for (int a = 1; a < argc; ++a) {
    if (!std::strcmp(argv[a], "--tile") && a + 1 < argc) tile = std::atoll(argv[++a]);
    else if (!std::strcmp(argv[a], "--amps") && a + 1 < argc) amps = (size_t)std::atoll(argv[++a]);
    else path = argv[a];
  }
  if (!path) {
    std::fprintf(stderr, "usage: qip_replay [--tile 0|1|2] [--amps K] circuit.qipc\n");
    return 2;
  }
  try {
    std::ifstream in(path);
    if (!in) throw qip::CircuitError(std::string("cannot open ") + path);
    const auto circ = qip::replay::load<double>(in);
    qip::HipState<double> st(circ.n);
    st.set_option("tile", tile);
    const auto t0 = std::chrono::steady_clock::now();
    const auto results = qip::replay::run(circ, st);
    st.sync();
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    for (const auto& r : results) {
      if (r.stochastic) {
        std::printf("probs");
        for (double p : r.probs) std::printf(" %.17g", p);
        std::printf("\n");
      } else {
        std::printf("measure %zu %.17g\n", r.measured, r.prob);
      }
    }
    std::printf("norm_sqr %.17g\n", st.norm_sqr());
    if (amps) {
      std::vector<std::complex<double>> head(std::min(amps, size_t(1) << circ.n));
      qip::check(qip_hip_state_download(st.handle(), head.data(), 0, head.size()));
      for (size_t i = 0; i < head.size(); ++i) std::printf("amp %zu %.17g %.17g\n", i, head[i].real(), head[i].imag());
    }
    std::printf("n %zu statements %zu ms %.3f\n", circ.n, circ.items.size(), ms);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "qip_replay: %s\n", e.what());
    return 1;
  }
  return 0;
}
