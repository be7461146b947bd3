#!/bin/bash
# build throw-away variants of the library with -DVITA_LAB=<n> into tools/lab/v<n>/ (results may be wrong; timing only)
set -e
for v in "$@"; do
  mkdir -p tools/lab/v$v
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -w -DVITA_LAB=$v $LABFLAGS -shared -o tools/lab/v$v/libvita_hip.so vita_amd/csrc/vita_api.hip vita_amd/csrc/vita_gemm_f32.hip vita_amd/csrc/vita_gemm_f32_dma.hip vita_amd/csrc/vita_elementwise.hip
done
