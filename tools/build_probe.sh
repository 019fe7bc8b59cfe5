#!/bin/sh
# Build the tuning probes (not part of the library) into tools/bin/.
cd "$(dirname "$0")/.." && mkdir -p tools/bin
HIPCC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w"
# the GEMM probe links the library's own kernels
$HIPCC -Iinclude -Ipair-net_amd/csrc tools/gemm_probe.hip pair-net_amd/csrc/gemm.hip -o tools/bin/gemm_probe
$HIPCC -Iinclude -Ipair-net_amd/csrc tools/gemm_glds_probe.hip pair-net_amd/csrc/gemm.hip -o tools/bin/gemm_glds_probe
$HIPCC -Iinclude -Ipair-net_amd/csrc tools/gemm_split_probe.hip pair-net_amd/csrc/gemm.hip -o tools/bin/gemm_split_probe
for p in mfma_probe mfma_power_probe gather_probe lds_gather_probe grid_barrier_probe placement_probe; do
  $HIPCC tools/$p.hip -o tools/bin/$p
done
$HIPCC -shared -fPIC tools/mfma_hammer.hip -o tools/bin/libmfma_hammer.so
