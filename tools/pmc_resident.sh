#!/bin/bash
# counter passes over the weights-resident serving kernel (tools/serve_trace.py: generate_actions at B = 25, one kernel per call)
# usage (on the GPU box, from the repo root): tools/pmc_resident.sh <round-tag>
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
d=gpurun_out/pmc_${tag}_resident
mkdir -p $d
rpass() {  # rpass <name> <counters...>
  local n=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $d -o $n -- python3 tools/serve_trace.py > $d/$n.log 2>&1 || { tail -5 $d/$n.log; return 1; }
}
rpass fetch FETCH_SIZE || exit 1
rpass write WRITE_SIZE || exit 1
rpass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE || exit 1
rpass lds SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU || exit 1
rpass tcc TCC_HIT_sum TCC_MISS_sum || exit 1
echo pmc resident done
