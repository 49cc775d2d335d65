#!/bin/bash
# round 5: the kernels of the LAST time step of the 512^3 sphere run in launch order (runs of the same kernel collapsed), to see which copies and vector
# updates the host mirror puts between the solver kernels.  Output: gpurun_out/r05_step_sequence.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_stepseq
rm -rf $O && mkdir -p $O
FLUCA_STEP_TIMING=1 timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o k -- $R/fluca_amd/lib/flow_configs -config sphere -n 512 -ns_max_steps ${STEPS:-12} -ns_ksp_type preonly -ns_abf_schur_pc_type mg -ns_abf_momentum_ksp_type chebyshev -ns_abf_momentum_guess_previous > $O/trace.log 2>&1
grep "fluca step" $O/trace.log | tail -2
python3 - <<'PY' > $R/gpurun_out/r05_step_sequence.txt
import csv, os
R = os.environ["GRAFT_REPO_ROOT"]
rows = list(csv.DictReader(open(R + "/gpurun_out/r05_stepseq/trace/k_kernel_trace.csv")))
t = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")) for r in rows)
pu = [i for i, x in enumerate(t) if "k_pressure_update" in x[2]]
seg = t[pu[-2] + 1: pu[-1] + 1]
t0 = seg[0][0]
out, run = [], None
for s, e, n in seg:
    d = (e - s) / 1e3
    if run and run[0] == n:
        run[1] += 1; run[2] += d
    else:
        if run: out.append(run)
        run = [n, 1, d, (s - t0) / 1e6]
if run: out.append(run)
print("last step: %.3f ms wall, %d launches" % ((seg[-1][1] - t0) / 1e6, len(seg)))
for n, c, d, at in out:
    if d >= 150 or "k_mom3" in n:
        print("%8.3f ms  +%9.1f us  x%-4d %s" % (at, d, c, n[:90]))
PY
cat $R/gpurun_out/r05_step_sequence.txt
