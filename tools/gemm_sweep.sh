#!/bin/bash
# fc1 / fc2 / in_proj / decoder-out / obs_encoder shapes of the hook_package B=4096 path
set -e
B=tools/bench_gemm
timeout -k 5 60 $B 4096 2048 512 1 3 0 50
timeout -k 5 60 $B 4096 512 2048 0 0 1 50
timeout -k 5 60 $B 4096 512 512 0 0 0 50
timeout -k 5 60 $B 4096 512 512 1 0 1 50
timeout -k 5 60 $B 4096 336 512 0 0 0 50
timeout -k 5 60 $B 4096 512 533 0 0 0 50
timeout -k 5 60 $B 4096 2048 512 0 0 0 50
