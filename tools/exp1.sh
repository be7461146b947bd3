#!/bin/bash
B=tools/bench_gemm
timeout -k 5 60 $B 512 512 2048 0 0 1 50
timeout -k 5 60 $B 64 128 2048 0 0 1 50
timeout -k 5 60 $B 2048 512 2048 0 0 1 50
timeout -k 5 60 $B 4096 512 2048 0 0 1 50
timeout -k 5 60 $B 8192 512 2048 0 0 1 50
timeout -k 5 60 $B 16384 512 2048 0 0 1 20
timeout -k 5 60 $B 16384 2048 512 0 0 0 20
