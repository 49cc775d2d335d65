#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_multirank.py -x -q -k "decomposed_solve and 10" > gpurun_out/r03_cgsr_test.log 2>&1; tail -5 gpurun_out/r03_cgsr_test.log
# price of the multi-rank path in RCCL loopback: default pair against the single-reduction form
for n in 512 256; do for sr in 0 1; do python tools/experiments/loopback_bench.py --cells $n --axes 3 --single-reduction $sr 2>/dev/null | tee -a gpurun_out/r03_cgsr_loopback.txt; done; done
