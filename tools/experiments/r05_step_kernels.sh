#!/bin/bash
# round 5 (step 24, kspA from the previous velocity): which kernels a whole 512^3 time step of the sphere configuration spends its time in (fractional step, multigrid on S, Chebyshev on A):
# rocprofv3 kernel trace of examples/flow_configs.c, then the kernels of the LAST step grouped by name.  Output: gpurun_out/r05_step_kernels${TAG}.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_stepprof
rm -rf $O && mkdir -p $O
FLUCA_STEP_TIMING=1 timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o k -- $R/fluca_amd/lib/flow_configs -config sphere -n 512 -ns_max_steps ${STEPS:-24} -ns_abf_momentum_guess_previous -ns_ksp_type preonly -ns_abf_schur_pc_type mg -ns_abf_momentum_ksp_type chebyshev $EXTRA > $O/trace.log 2>&1
echo rc=$?
grep "^step\|^config\|fluca step" $O/trace.log
python3 - <<'PY' > $R/gpurun_out/r05_step_kernels${TAG}.txt
import csv, collections, os
R = os.environ["GRAFT_REPO_ROOT"]
rows = list(csv.DictReader(open(R + "/gpurun_out/r05_stepprof/trace/k_kernel_trace.csv")))
t = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")) for r in rows)
# the last step: everything after the last-but-one launch of the pressure update
pu = [i for i, x in enumerate(t) if "k_pressure_update" in x[2]]
seg = t[pu[-2] + 1: pu[-1] + 1] if len(pu) >= 2 else t
tot = collections.defaultdict(lambda: [0, 0.0])
for s, e, n in seg:
    tot[n][0] += 1
    tot[n][1] += (e - s) / 1e6
wall = (seg[-1][1] - seg[0][0]) / 1e6
print("last step: wall %.3f ms, kernels %.3f ms, %d launches" % (wall, sum(v[1] for v in tot.values()), len(seg)))
for n, v in sorted(tot.items(), key=lambda kv: -kv[1][1])[:60]:
    print("%9.3f ms %5d  %8.3f ms each  %s" % (v[1], v[0], v[1] / v[0], n[:120]))
PY
cat $R/gpurun_out/r05_step_kernels${TAG}.txt
