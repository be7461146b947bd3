#!/bin/bash
# small-M GEMM shapes of the flow net with and without the split-K scratch
for M in 25 128 512; do
  for a in "2048 512 0 3 0" "512 2048 0 0 1" "512 512 0 0 0"; do
    for sk in 0 1; do
      echo -n "splitk=$sk "; SPLITK=$sk timeout -k 5 60 tools/bench_gemm $M $a 200 || exit 1
    done
  done
done
