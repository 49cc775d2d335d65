#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in "FLUCA_MOM_PW=2" "FLUCA_MOM_PW=3" "FLUCA_MOM_PW=3 FLUCA_MOM_CHUNKS=8" "FLUCA_MOM_PW=3 FLUCA_MOM_CHUNKS=16" "FLUCA_MOM_PW=2" "FLUCA_MOM_PW=3"; do
  env $v python tools/mom_bench.py --reps 5 --modes 0 2>/dev/null | tee -a gpurun_out/r03_mom_variants_d.txt
done
FLUCA_MOM_PW=3 python -m pytest tests/test_gpu_momentum.py -x -q -k "solve or bcgs or abf" 2>&1 | tail -3
