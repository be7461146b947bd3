#!/bin/bash
B=tools/bench_gemm
for pads in "0 0" "32 0" "0 32" "32 32" "4 4" "64 64"; do
timeout -k 5 60 $B 4096 512 2048 0 0 1 50 $pads
done
for pads in "0 0" "32 0" "32 32"; do
timeout -k 5 60 $B 4096 2048 512 0 0 0 50 $pads
timeout -k 5 60 $B 4096 2048 512 1 3 0 50 $pads
timeout -k 5 60 $B 4096 512 512 0 0 0 50 $pads
done
