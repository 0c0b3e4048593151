/* qip_jitc — the helper process of the run-time compiler (include/qip_hip.h, qip_hip_jit_stats2).
 *
 *     qip_jitc  [-u]  <fma 0|1> <segment source file> <code object file>  [<fma> <source> <object> ...]
 *
 * -u: remove each source file once it has been dealt with (background jobs nobody waits for: qip_hip option "tile_auto").
 *
 * libqip_hip.so spawns several of these side by side for the segments of a tile plan that are new: hiprtc serialises the
 * compilations of ONE process, separate processes scale with the host's cores.  Host code only — no device is touched; each
 * job is one call of qip_hip_jit_compile_file, which writes the code object in the disk cache's format (temporary name +
 * rename).  Exit status = number of jobs that failed (their messages on stderr); the library compiles those itself. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "../../include/qip_hip.h"

int main(int argc, char** argv) {
  int failed = 0, first = 1, unlink_sources = 0;
  if (argc > 1 && !strcmp(argv[1], "-u")) {
    unlink_sources = 1;
    first = 2;
  }
  if (argc - first < 3 || (argc - first) % 3 != 0) {
    fprintf(stderr, "usage: %s [-u] <fma 0|1> <source> <object> [...]\n", argc ? argv[0] : "qip_jitc");
    return 64;
  }
  for (int i = first; i + 2 < argc; i += 3) {
    if (qip_hip_jit_compile_file(argv[i + 1], atoi(argv[i]), argv[i + 2]) != QIP_OK) {
      fprintf(stderr, "qip_jitc: %s: %s\n", argv[i + 1], qip_hip_last_error());
      ++failed;
    }
    if (unlink_sources) (void)unlink(argv[i + 1]);
  }
  return failed > 255 ? 255 : failed;
}
