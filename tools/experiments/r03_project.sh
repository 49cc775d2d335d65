#!/bin/bash
# round 3: fl_poisson_project with the six updates in one pass over p (k_project_all) against one kernel per output array
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_poisson.py tests/test_gpu_golden.py tests/test_gpu_timestep.py tests/test_host_mirror.py tests/test_gpu_momentum.py tests/test_gpu_multirank.py -x -q -k "not decomposed_solve_matches and not multigrid and not mg" > gpurun_out/r03_project_tests.log 2>&1 || { tail -30 gpurun_out/r03_project_tests.log; exit 1; }
tail -2 gpurun_out/r03_project_tests.log
python - <<'PY'
import os, time, torch, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
PY
for f in 0 1 0 1; do
  FLUCA_PROJECT_FUSED=$f python - <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from fluca_amd.poisson import Poisson
P = Poisson.uniform((512,) * 3, [(0, 1)] * 3, [1, 1, 1, 1, 4, 1], 1e-3)
g = torch.Generator(device="cuda").manual_seed(1)
rnd = lambda n: torch.rand(n, dtype=torch.float64, device="cuda", generator=g)
p = rnd(P.ncell); v = [rnd(P.ncell) for _ in range(3)]; V = [rnd(P.nface[d]) for d in range(3)]
P.project(p, v, V); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): P.project(p, v, V)
e1.record(); torch.cuda.synchronize()
print("FLUCA_PROJECT_FUSED=%s  fl_poisson_project at 512^3: %.3f ms" % (os.environ.get("FLUCA_PROJECT_FUSED"), e0.elapsed_time(e1) / 10))
PY
done
