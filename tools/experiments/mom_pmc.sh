#!/bin/bash
# hardware-counter passes over tools/mom_bench.py (one counter group per pass, as the pool requires)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_WAIT_ANY" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum SQ_LDS_BANK_CONFLICT"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout -k 10 180 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/mompmc/$tag -o p -- python3 $R/tools/mom_bench.py --cells 512 --reps 3 > $R/gpurun_out/mompmc_$tag.log 2>&1
  echo "$tag rc=$?"
done
