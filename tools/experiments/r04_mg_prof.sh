#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04_mg_prof -o p -- python3 $R/tools/mg_bench.py --cells 512 --skip-jacobi  > $R/gpurun_out/r04_mg_prof.log 2>&1
grep cells $R/gpurun_out/r04_mg_prof.log
python3 - <<'PY'
import csv, collections, os
R=os.environ["GRAFT_REPO_ROOT"]
rows=list(csv.DictReader(open(R+"/gpurun_out/r04_mg_prof/p_kernel_trace.csv")))
# the second solve: kernels after the midpoint of the k_mg_dots sequence
t=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].split("(")[0].replace("void ","")) for r in rows]
t.sort()
dots=[i for i,x in enumerate(t) if "k_mg_dots" in x[2]]
half=dots[len(dots)//2]
# find the start of the second solve: first k_pad_copy before dots[half]... simply take from dots[half]-? use index of first kernel after the (len/2)-th dots minus 1
seg=t[dots[len(dots)//2-1]+1:dots[-1]+1]
tot=collections.defaultdict(lambda:[0,0.0])
for s,e,n in seg:
    tot[n][0]+=1; tot[n][1]+=(e-s)/1e6
wall=(seg[-1][1]-seg[0][0])/1e6
print("segment wall ms",round(wall,3),"kernel ms",round(sum(v[1] for v in tot.values()),3),"launches",len(seg))
for n,v in sorted(tot.items(), key=lambda kv:-kv[1][1])[:22]:
    print(f"{v[1]:9.3f} ms {v[0]:5d}  {n[:110]}")
PY
