#!/bin/bash
# round 3: vectorised BiCGStab update kernels (k_mom_pw2) against round 1's, parity first
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_momentum.py tests/test_gpu_multirank.py -x -q -k "momentum or Momentum or abf or time_steps" > gpurun_out/r03_mom_test.log 2>&1; tail -3 gpurun_out/r03_mom_test.log
for v in "FLUCA_MOM_PW=1" "FLUCA_MOM_PW=2" "FLUCA_MOM_PW_BLOCKS=1024" "FLUCA_MOM_PW_BLOCKS=4096" "FLUCA_MOM_PW=1" "FLUCA_MOM_PW=2"; do
  env $v python tools/mom_bench.py --reps 5 --modes 0 2>/dev/null | tee -a gpurun_out/r03_mom_variants_c.txt
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03_mom_trace -o p -- python3 $GRAFT_REPO_ROOT/tools/mom_bench.py --reps 5 --modes 4 > $GRAFT_REPO_ROOT/gpurun_out/r03_mom_trace.log 2>&1
head -30 $GRAFT_REPO_ROOT/gpurun_out/r03_mom_trace/p_kernel_stats.csv
