#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_mg.py -x -q > gpurun_out/r03_mg_test.log 2>&1; tail -2 gpurun_out/r03_mg_test.log
for pr in 0 1 0 1; do python tools/mg_bench.py --cells 512 --skip-jacobi --prolong $pr --smooth 3 2>/dev/null | tee -a gpurun_out/r03_mg_bench2.txt; done
