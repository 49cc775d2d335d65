#!/bin/bash
# round 3: multigrid cycle with the parent-centred / tiled prolongation, the dots formed by the last smoothing sweep, r -= alpha q on the first smoothing step
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_mg.py tests/test_momentum_fixtures.py tests/test_gpu_cheb2.py tests/test_gpu_multirank.py -x -q -k "not decomposed_solve_matches" > gpurun_out/r03_mg3_tests.log 2>&1 || { tail -30 gpurun_out/r03_mg3_tests.log; exit 1; }
tail -2 gpurun_out/r03_mg3_tests.log
for rep in 1 2; do
  for v in "FLUCA_MG_PROLONG_TILE=0 FLUCA_MG_FUSED_DOTS=0" "FLUCA_MG_PROLONG_TILE=1 FLUCA_MG_FUSED_DOTS=0" "FLUCA_MG_PROLONG_TILE=2 FLUCA_MG_FUSED_DOTS=0" "FLUCA_MG_PROLONG_TILE=2 FLUCA_MG_FUSED_DOTS=1"; do
    echo "== $v" >> gpurun_out/r03_mg3.txt
    env $v timeout -k 10 300 python tools/mg_bench.py --cells 512 --skip-jacobi --prolong 1 --smooth 3 >> gpurun_out/r03_mg3.txt 2>/dev/null || exit 1
  done
done
cat gpurun_out/r03_mg3.txt
