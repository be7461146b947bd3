#!/bin/bash
for v in 0 "$@"; do
  if [ $v = 0 ]; then L=vita_amd; else L=tools/lab/v$v; fi
  echo "== variant $v"
  LD_LIBRARY_PATH=$L timeout -k 5 60 tools/bench_gemm 64 128 2048 0 0 1 50
  LD_LIBRARY_PATH=$L timeout -k 5 60 tools/bench_gemm 4096 512 2048 0 0 1 50
done
