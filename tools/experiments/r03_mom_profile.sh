#!/bin/bash
# round 3: rocprofv3 kernel trace + counter passes of the momentum block as committed (tools/mom_bench.py, 512^3)
cd $GRAFT_REPO_ROOT
tools/prof/pmc_kernel.sh r03_mom_final $GRAFT_REPO_ROOT/tools/mom_bench.py --reps 5 --modes 4 || exit 1
python tools/prof/pmc_table.py gpurun_out/r03_mom_final "k_mom" 1.0 > gpurun_out/r03_mom_final_table.json
tail -3 gpurun_out/r03_mom_final/trace.log
