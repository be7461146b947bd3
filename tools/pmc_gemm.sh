#!/bin/bash
# usage: tools/pmc_gemm.sh <tag> <bench_gemm args...>   -> gpurun_out/pmc_<tag>/
# one counter pass per rocprofv3 run (<= 8 SQ counters per pass)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_$tag
mkdir -p $out
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_WAIT_INST_LDS --output-format csv -d $out -o p1 -- tools/bench_gemm "$@" 5 > $out/p1.log 2>&1 || exit 1
timeout -s KILL 90 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d $out -o p2 -- tools/bench_gemm "$@" 5 > $out/p2.log 2>&1 || exit 1
ls $out
