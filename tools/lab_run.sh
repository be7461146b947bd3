#!/bin/bash
# usage: tools/lab_run.sh <variant dirs under tools/lab ...>  — the hook_package B=4096 GEMM shapes on each variant
for v in "$@"; do
  echo "== $v"
  for a in "4096 2048 512 0 3 0" "4096 512 2048 0 0 1" "4096 512 512 0 0 0" "4096 336 512 0 0 0" "4096 2048 512 1 3 0"; do
    timeout -k 5 60 tools/lab/$v/bench_gemm $a 30 || exit 1
  done
done
