#!/bin/bash
# round 3: k_mom2 with scaled cell-major tables + coefficient-form rows: parity, then timing per plan
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_momentum.py -x -q > gpurun_out/r03_mom_test.log 2>&1; tail -3 gpurun_out/r03_mom_test.log
for v in "FLUCA_MOM_CHUNKS=2" "FLUCA_MOM_CHUNKS=4" "FLUCA_MOM_CHUNKS=8" "FLUCA_MOM_KERNEL=1"; do
  env $v python tools/mom_bench.py --reps 10 2>/dev/null | tee -a gpurun_out/r03_mom_variants_b.txt
done
