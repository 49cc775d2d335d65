#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "# local (single rank)" | tee gpurun_out/r03_ibm_scale.txt
python tools/ibm_bench.py --loopback 0 2>/dev/null | tee -a gpurun_out/r03_ibm_scale.txt
echo "# RCCL loopback (multi-rank path: interpolation ends with the all-reduce of U)" | tee -a gpurun_out/r03_ibm_scale.txt
python tools/ibm_bench.py --loopback 1 2>/dev/null | grep markers | tee -a gpurun_out/r03_ibm_scale.txt
