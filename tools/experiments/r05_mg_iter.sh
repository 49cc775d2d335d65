#!/bin/bash
# round 5: the kernels of ONE iteration of the multigrid-PCG at 512^3, in launch order (everything >= 20 us), from a rocprofv3 kernel trace of tools/mg_bench.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05_mg_iter -o p -- python3 $R/tools/mg_bench.py --cells 512 --skip-jacobi > $R/gpurun_out/r05_mg_iter.log 2>&1
grep cells $R/gpurun_out/r05_mg_iter.log
python3 - <<'PY'
import csv, os
R = os.environ["GRAFT_REPO_ROOT"]
rows = list(csv.DictReader(open(R + "/gpurun_out/r05_mg_iter/p_kernel_trace.csv")))
t = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), r.get("Grid_Size_X") or r.get("Grid_Size")) for r in rows)
starts = [i for i, x in enumerate(t) if "k_mg_scal_set" in x[2]]
seg = t[starts[-1]:]
wall = (seg[-1][1] - seg[0][0]) / 1e6
print("solve: wall %.3f ms, kernels %.3f ms, %d launches" % (wall, sum((e - s) / 1e6 for s, e, _, _ in seg), len(seg)))
idx = [i for i, x in enumerate(seg) if "k_mg_pwd<4>" in x[2]]
a, b = idx[2], idx[3]
tot = small = 0.0
for s, e, n, g in seg[a:b]:
    d = (e - s) / 1e3
    tot += d
    if d > 20:
        print("%9.1f us  %-60s grid %s" % (d, n[:60], g))
    else:
        small += d
print("one iteration: kernels %.1f us (of which %.1f us in %d launches under 20 us), wall %.1f us, %d launches" % (tot, small, sum(1 for s, e, _, _ in seg[a:b] if (e - s) / 1e3 <= 20), (seg[b][0] - seg[a][0]) / 1e3, b - a))
PY
