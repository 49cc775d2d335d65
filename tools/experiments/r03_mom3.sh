#!/bin/bash
# round 3: k_mom3 (v0interp formed in the kernel) -- parity tests, z-chunk sweep, counters
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_momentum.py tests/test_momentum_fixtures.py tests/test_gpu_timestep.py -x -q > gpurun_out/r03_mom3_tests.log 2>&1 || { tail -30 gpurun_out/r03_mom3_tests.log; exit 1; }
tail -2 gpurun_out/r03_mom3_tests.log
for c in 0 2 8; do
  echo "== FLUCA_MOM_CHUNKS=$c" >> gpurun_out/r03_mom3_chunks.txt
  FLUCA_MOM_CHUNKS=$c timeout -k 10 300 python tools/mom_bench.py --cells 512 --fly 1 >> gpurun_out/r03_mom3_chunks.txt 2>/dev/null || exit 1
done
cat gpurun_out/r03_mom3_chunks.txt
bash tools/prof/pmc_kernel.sh r03_mom3_pmc $GRAFT_REPO_ROOT/tools/mom_bench.py --cells 512 --fly 1 --nosolve --reps 5 && python tools/prof/pmc_table.py gpurun_out/r03_mom3_pmc k_mom3 1.0 > gpurun_out/r03_mom3_pmc/table.json
