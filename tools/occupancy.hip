// Occupancy the runtime computes for the tile-sweep kernels (blocks per CU), and device limits.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include tools/occupancy.hip -o tools/occupancy
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../rustqip_amd/csrc/qip_kernels.h"
using namespace qipk;
template <typename K> static void show(const char* name, K kern, size_t lds) {
  int nb = -1;
  hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, kBlock, lds);
  hipFuncAttributes a;
  hipFuncGetAttributes(&a, reinterpret_cast<const void*>(kern));
  printf("%-40s lds=%zu  blocks/CU=%d (%s)  numRegs=%d sharedStatic=%zu maxThreads=%d\n", name, lds, nb,
         hipGetErrorString(e), a.numRegs, a.sharedSizeBytes, a.maxThreadsPerBlock);
}
int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("%s CUs=%d ldsPerBlock=%zu ldsPerCU=%zu regsPerBlock=%d regsPerCU=%d maxThreadsPerCU=%d clock=%d kHz\n", p.gcnArchName,
         p.multiProcessorCount, p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor, p.regsPerBlock,
         p.regsPerMultiprocessor, p.maxThreadsPerMultiProcessor, p.clockRate);
  show("k_tile_passes<double,NT>", k_tile_passes<double, true>, 32768);
  show("k_tile_passes<float,NT>", k_tile_passes<float, true>, 16384);
  show("k_tile_gates<double,NT>", k_tile_gates<double, true>, 32768);
  return 0;
}
