#!/bin/bash
# k_mom3: non-temporal hint on the tile's own per-plane loads (FL_MOM3_NT: 0 none, 1 all waves, 2 inner rows only), same box, alternating
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in 0 1 2; do
  touch fluca_amd/csrc/fl_momentum.hip
  FL_DEFINES="FL_MOM3_NT=$v" python -c "from fluca_amd import build; build.build()" > gpurun_out/r03_wpe_build.log 2>&1 || { tail -5 gpurun_out/r03_wpe_build.log; exit 1; }
  echo "== FL_MOM3_NT=$v $(python tools/mom_bench.py --cells 512 --fly 1 --nosolve --modes 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['kernel_ms_mode0'],4), round(d['kernel_ms_mode1'],4), round(d['kernel_ms_mode2'],4), 'stream', round(d['stream15r3w_ms_2048'],3))")"
done
done
