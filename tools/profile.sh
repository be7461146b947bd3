#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile.sh <round-tag>
# 1. rocprofv3 --kernel-trace --stats over the default bench.py command -> gpurun_out/prof_<tag>/
# 2. counter passes (one rocprofv3 run each; FETCH_SIZE and WRITE_SIZE cannot share a pass) over the dominant GEMMs,
#    through the standalone C-ABI driver tools/bench_gemm                  -> gpurun_out/pmc_<tag>_<kernel>/
# tools/pmc_summary.py turns the CSVs into profiles/<tag>_pmc.json and profiles/traffic.json.
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/prof_$tag
mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $tag -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out.log 2>&1 || { tail -20 $out.log; exit 1; }
pass() {  # pass <dir> <name> <counters...> -- <bench_gemm args>
  local d=$1 n=$2; shift 2
  local ctr=()
  while [ "$1" != "--" ]; do ctr+=("$1"); shift; done
  shift
  timeout -s KILL 90 rocprofv3 --pmc "${ctr[@]}" --output-format csv -d $d -o $n -- tools/bench_gemm "$@" 5 > $d/$n.log 2>&1 || { tail -5 $d/$n.log; return 1; }
}
for k in "fc1 4096 2048 512 0 3 0" "fc2 4096 512 2048 0 0 1"; do
  set -- $k
  name=$1; shift
  d=gpurun_out/pmc_${tag}_$name
  mkdir -p $d
  pass $d fetch FETCH_SIZE -- "$@" || exit 1
  pass $d write WRITE_SIZE -- "$@" || exit 1
  pass $d sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -- "$@" || exit 1
  pass $d tcc TCC_HIT_sum TCC_MISS_sum -- "$@" || exit 1
done
# the persistent sampler (one launch per solve): the same counter passes over tools/persistent_only.py
d=gpurun_out/pmc_${tag}_persistent
mkdir -p $d
ppass() {  # ppass <name> <counters...>
  local n=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $d -o $n -- python3 tools/persistent_only.py 6 > $d/$n.log 2>&1 || { tail -5 $d/$n.log; return 1; }
}
ppass fetch FETCH_SIZE || exit 1
ppass write WRITE_SIZE || exit 1
ppass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE || exit 1
ppass tcc TCC_HIT_sum TCC_MISS_sum || exit 1
echo profile done
