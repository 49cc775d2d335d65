#!/bin/bash
cd $GRAFT_REPO_ROOT
tools/prof/pmc_kernel.sh r03_cheb2_pmc $GRAFT_REPO_ROOT/tools/cheb_bench.py 512 40 || exit 1
python tools/prof/pmc_table.py gpurun_out/r03_cheb2_pmc "k_c" 0.3 > gpurun_out/r03_cheb2_pmc_table.json
