#!/bin/bash
# rocprofv3 hardware-counter passes over one command, one counter group per pass (as the pool requires); kernel trace first.
# usage: tools/prof/pmc_kernel.sh <outdir under gpurun_out> <python script and args ...>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
shift
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o p -- python3 "$@" > $O/trace.log 2>&1 || { echo "trace failed"; tail -5 $O/trace.log; exit 1; }
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU" \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $O/pmc$i -o p -- python3 "$@" > $O/pmc$i.log 2>&1 || { echo "pass $i failed"; tail -5 $O/pmc$i.log; exit 1; }
  echo "pass $i ok: $grp"
done
