#!/bin/bash
# k_cheb2 ablations (rebuilt on the box): ABL = 1 no Jacobi divisions in step 1's tile part, 2 = also no ring-1 step-1 computation ... (see fl_cheb2.hip)
cd $GRAFT_REPO_ROOT
python tools/cheb_bench.py 512 100 2>/dev/null | grep "fuse=2" | tail -1
for d in "FL_CHEB2_ABL=1" "FL_CHEB2_ABL=2" "FL_CHEB2_ABL=3"; do
  touch fluca_amd/csrc/fl_cheb2.hip
  FL_DEFINES="$d" python -c "from fluca_amd import build; build.build()" > gpurun_out/r03_abl_build.log 2>&1 || { tail -5 gpurun_out/r03_abl_build.log; exit 1; }
  echo "== $d"; python tools/cheb_bench.py 512 100 2>/dev/null | grep "fuse=2" | tail -1
done
