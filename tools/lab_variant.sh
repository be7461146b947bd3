#!/bin/bash
# build throw-away variants of the library into tools/lab/<name>/ (timing only):  tools/lab_variant.sh <name> <extra hipcc flags...>
set -e
name=$1; shift
mkdir -p tools/lab/$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -w "$@" -shared -o tools/lab/$name/libvita_hip.so vita_amd/csrc/vita_api.hip vita_amd/csrc/vita_gemm_f32.hip vita_amd/csrc/vita_gemm_f32_dma.hip vita_amd/csrc/vita_gemm_f32_t.hip vita_amd/csrc/vita_elementwise.hip vita_amd/csrc/vita_train.hip vita_amd/csrc/vita_loss_step.hip vita_amd/csrc/vita_api_bf16.hip vita_amd/csrc/vita_contrastive.hip vita_amd/csrc/vita_meanflow.hip vita_amd/csrc/vita_gemm_f32_small.hip vita_amd/csrc/vita_ode_persistent.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 "$@" tools/bench_gemm.cpp -o tools/lab/$name/bench_gemm -Ltools/lab/$name -lvita_hip -Wl,-rpath,'$ORIGIN'
