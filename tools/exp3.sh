#!/bin/bash
for L in vita_amd tools/lab/v7; do
echo "== $L"
LD_LIBRARY_PATH=$L timeout -k 5 60 tools/bench_gemm 4096 2048 512 0 0 0 50
LD_LIBRARY_PATH=$L timeout -k 5 60 tools/bench_gemm 4096 2048 512 1 0 0 50
LD_LIBRARY_PATH=$L timeout -k 5 60 tools/bench_gemm 4096 2048 512 0 3 0 50
LD_LIBRARY_PATH=$L timeout -k 5 60 tools/bench_gemm 4096 2048 512 1 3 0 50
LD_LIBRARY_PATH=$L timeout -k 5 60 tools/bench_gemm 4096 2048 512 0 2 0 50
done
