#!/bin/bash
# rocprofv3 per-kernel summary of whole CNLinear time steps (examples/flow_configs.c, channel, GMRES outer solve)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/stepprof
N=${1:-512}
mkdir -p $O
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o k -- $R/fluca_amd/lib/flow_configs -config channel -n $N -ns_max_steps 2 -ns_abf_schur_pc_type mg -ns_ksp_rtol 1e-4 -ns_ksp_gmres_restart 12 > $O/trace.log 2>&1
echo rc=$?
grep "step\|config" $O/trace.log
