#!/bin/bash
# k_cg_A / k_cg_Bq with at most 128 VGPRs (two 512-thread blocks per CU; spills) against the default (147 / 152: one block per CU), same box, alternating
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for d in "" "FL_CGA_WPE=4" "FL_CGA_WPE=4 FL_CGB_WPE=4"; do
  touch fluca_amd/csrc/fl_kernels.hip
  FL_DEFINES="$d" python -c "from fluca_amd import build; build.build()" > gpurun_out/r03_wpe_build.log 2>&1 || { tail -5 gpurun_out/r03_wpe_build.log; exit 1; }
  python bench.py --steps 100 --warmup 20 --skip-cpu --skip-extras --skip-configs --placement off 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('== [$d]', round(d['value'],1), 'it/s', round(d['ms_per_step'],4), 'ms/it')"
done
done
