#!/bin/bash
for v in "$@"; do
  echo "== $v"
  TILES=256 timeout -k 5 60 tools/lab/$v/bench_gemm 4096 2048 512 0 3 0 30 || exit 1
  TILES=256 timeout -k 5 60 tools/lab/$v/bench_gemm 4096 512 2048 0 0 1 30 || exit 1
done
