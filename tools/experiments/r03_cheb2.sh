#!/bin/bash
# round 3: k_cheb2 plan variants at 512^3 (z chunks), same box
cd $GRAFT_REPO_ROOT
for v in "FLUCA_CHEB2_NCHUNK=2" "FLUCA_CHEB2_NCHUNK=4" "FLUCA_CHEB2_NCHUNK=8" "FLUCA_CHEB2_NCHUNK=3" "FLUCA_CHEB2_NCHUNK=2" "FLUCA_CHEB2_NCHUNK=4"; do
  echo "== $v" | tee -a gpurun_out/r03_cheb2_plan.txt
  env $v python tools/cheb_bench.py 512 100 2>/dev/null | grep "fuse=2" | tee -a gpurun_out/r03_cheb2_plan.txt
done
