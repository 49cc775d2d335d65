#!/bin/bash
# round 5: kernels of ONE multigrid-PCG solve at 512^3 (the second solve of tools/mg_bench.py: from its k_mg_scal_set to its last kernel)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05_mg_prof -o p -- python3 $R/tools/mg_bench.py --cells 512 --skip-jacobi  > $R/gpurun_out/r05_mg_prof.log 2>&1
grep cells $R/gpurun_out/r05_mg_prof.log
python3 - <<'PY'
import csv, collections, os
R=os.environ["GRAFT_REPO_ROOT"]
rows=list(csv.DictReader(open(R+"/gpurun_out/r05_mg_prof/p_kernel_trace.csv")))
t=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].split("(")[0].replace("void ","")) for r in rows]
t.sort()
starts=[i for i,x in enumerate(t) if "k_mg_scal_set" in x[2]]
seg=t[starts[-1]:]
tot=collections.defaultdict(lambda:[0,0.0])
for s,e,n in seg:
    tot[n][0]+=1; tot[n][1]+=(e-s)/1e6
wall=(seg[-1][1]-seg[0][0])/1e6
print("segment wall ms",round(wall,3),"kernel ms",round(sum(v[1] for v in tot.values()),3),"launches",len(seg))
gaps=sorted(((seg[i+1][0]-seg[i][1])/1e3, seg[i][2][:40], seg[i+1][2][:40]) for i in range(len(seg)-1))
print("largest gaps (us):", [(round(g,1),a,b) for g,a,b in gaps[-6:]])
print("sum of gaps > 3 us (ms):", round(sum(g for g,_,_ in gaps if g>3)/1e3,3))
for n,v in sorted(tot.items(), key=lambda kv:-kv[1][1])[:26]:
    print(f"{v[1]:9.3f} ms {v[0]:5d}  {n[:110]}")
PY
