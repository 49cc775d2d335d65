#!/bin/bash
# round 3: residual and restriction in one pass (st_body MODE 11)
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_mg.py tests/test_momentum_fixtures.py tests/test_gpu_multirank.py -x -q -k "not decomposed_solve_matches" > gpurun_out/r03_mg4_tests.log 2>&1 || { tail -30 gpurun_out/r03_mg4_tests.log; exit 1; }
tail -2 gpurun_out/r03_mg4_tests.log
for rep in 1 2 3; do
  for v in "FLUCA_MG_FUSED_RESTRICT=0" "FLUCA_MG_FUSED_RESTRICT=1"; do
    echo "== $v" >> gpurun_out/r03_mg4.txt
    env $v timeout -k 10 300 python tools/mg_bench.py --cells 512 --skip-jacobi --prolong 1 --smooth 3 >> gpurun_out/r03_mg4.txt 2>/dev/null || exit 1
  done
done
cat gpurun_out/r03_mg4.txt
