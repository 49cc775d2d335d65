#!/bin/bash
# k_cheb2 data-movement ablations (rebuilt on the box; results are garbage, only the time counts): FL_CHEB2_ABL bits: 1 no ring loads, 2 no stores,
# (the FL_CHEB2_ABL switches were temporary edits of fl_cheb2.hip -- ring loads replaced by constants, stores by a sum, the staging block and the barrier
#  compiled out -- and are not in the tree; this script documents how profiles/r03_cheb2_diagnosis.txt (d) was produced)
# 4 no LDS staging, 16 no barrier
cd $GRAFT_REPO_ROOT
for a in 0 1 2 4 16 3 20 23; do
  touch fluca_amd/csrc/fl_cheb2.hip
  FL_DEFINES="FL_CHEB2_ABL=$a" python -c "from fluca_amd import build; build.build()" > gpurun_out/r03_abl_build.log 2>&1 || { tail -5 gpurun_out/r03_abl_build.log; exit 1; }
  echo "== ABL=$a $(python tools/cheb_bench.py 512 100 2>/dev/null | grep 'fuse=2' | tail -1 | cut -c1-110)"
done
