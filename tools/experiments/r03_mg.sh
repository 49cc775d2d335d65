#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_mg.py -x -q > gpurun_out/r03_mg_test.log 2>&1; tail -4 gpurun_out/r03_mg_test.log
python -m pytest tests/test_gpu_multirank.py -x -q -k "multigrid" >> gpurun_out/r03_mg_test.log 2>&1; tail -4 gpurun_out/r03_mg_test.log
for pr in 0 1; do for nu in 3 2 1; do python tools/mg_bench.py --cells 512 --skip-jacobi --prolong $pr --smooth $nu 2>/dev/null | tee -a gpurun_out/r03_mg_bench.txt; done; done
