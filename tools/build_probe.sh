#!/bin/sh
# Build the GEMM tuning probe against the library's own kernels.
cd "$(dirname "$0")/.." && mkdir -p tools/bin && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -Iinclude -Ipair-net_amd/csrc \
  tools/gemm_probe.hip pair-net_amd/csrc/gemm.hip -o tools/bin/gemm_probe
