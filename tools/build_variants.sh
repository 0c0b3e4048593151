# A/B builds of libqip_hip.so that differ in compile-time switches of qip_kernels.h (here: the diagonal-run loop of k_tile_passes):
#   bash tools/build_variants.sh          -> rustqip_amd/lib/libqip_hip_v{PREFETCH}{ASM}.so for 11, 10, 00 (the default build is 01)
# run one with QIP_HIP_LIB=<path> (rustqip_amd/_ffi.py).  The objects of the other translation units are shared with the main build.
cd "$(dirname "$0")/../rustqip_amd" || exit 1
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC"
for v in "1 1" "1 0" "0 0"; do
  set -- $v
  ( /opt/rocm/bin/hipcc $F -DQIP_DIAG_PREFETCH=$1 -DQIP_DIAG_ASM=$2 -c csrc/qip_circuit.hip -o build/qip_circuit_v$1$2.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -Wl,--version-script=csrc/exports.map -o lib/libqip_hip_v$1$2.so \
      build/qip_core.o build/qip_launch.o build/qip_tile_sched.o build/qip_circuit_v$1$2.o build/qip_host.o build/qip_measure.o build/qip_dist.o -ldl ) &
done
wait
ls -la lib/
