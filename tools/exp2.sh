#!/bin/bash
for L in vita_amd tools/lab/v7 tools/lab/v11; do
echo "== $L"
for K in 512 2048 8192; do
LD_LIBRARY_PATH=$L timeout -k 5 60 tools/bench_gemm 64 128 $K 0 0 0 50
done
for K in 512 2048 8192; do
LD_LIBRARY_PATH=$L timeout -k 5 60 tools/bench_gemm 4096 512 $K 0 0 0 30
done
done
