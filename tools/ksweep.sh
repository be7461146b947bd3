#!/bin/bash
# BASELINE configs[3]: robomimic_square, B=4096 per GPU, Euler steps 1..100 -> one bench.py JSON line per K
out=${1:-gpurun_out/ksweep.jsonl}
: > $out
for K in 1 2 4 6 8 10 16 25 50 100; do
  steps=$(( 240 / (K + 5) + 2 ))
  timeout -k 10 200 python3 bench.py --task robomimic_square --ode-steps $K --steps $steps --warmup 2 --no-train --no-cpu-baseline >> $out 2>> $out.err || exit 1
done
