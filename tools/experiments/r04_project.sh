#!/bin/bash
# round 4: fl_poisson_project (stage 2 of PCApply_ABF) at 512^3: k_project_six (FLUCA_PROJECT_FUSED=3 on the caller's p, 2 on the padded copy) against
# round 3's k_project_all (1); FLUCA_PROJECT_VAR=nt,nxcd,blocks_per_xcd (the sweep under profiles/ was run with two more fields: rows per pass in front, VGPR cap behind).  Output: gpurun_out/r04_project.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_project.txt
: > $O
run() {
  FLUCA_PROJECT_FUSED=$1 FLUCA_PROJECT_VAR=$2 timeout -k 10 120 python3 - >> $O 2>/dev/null <<'PY' || exit 1
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from fluca_amd.poisson import Poisson
for bc in ([1, 1, 1, 1, 4, 1], [1, 2, 1, 1, 3, 3]):
    P = Poisson.uniform((512,) * 3, [(0, 1)] * 3, bc, 1e-3)
    g = torch.Generator(device="cuda").manual_seed(1)
    rnd = lambda n: torch.rand(n, dtype=torch.float64, device="cuda", generator=g)
    p = rnd(P.ncell); v = [rnd(P.ncell) for _ in range(3)]; V = [rnd(P.nface[d]) for d in range(3)]
    P.project(p, v, V); torch.cuda.synchronize()
    ms = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): P.project(p, v, V)
        e1.record(); torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1) / 10)
    print("FUSED=%s VAR=%-12s bc %s  fl_poisson_project at 512^3: %.3f ms (best of 3 x 10; %.0f GB/s of 104 B per cell)" % (os.environ.get("FLUCA_PROJECT_FUSED"), os.environ.get("FLUCA_PROJECT_VAR", ""), "cavity " if bc[1] == 1 else "channel", min(ms), 104 * P.ncell / min(ms) / 1e6))
    P.close()
PY
}
run 1 ""
run 2 ""
run 3 ""
for var in ${VARS:-0,8,1024 1,1,1024 1,8,256 1,8,2048 1,8,4096}; do run 3 $var; done
run 1 ""
cat $O
