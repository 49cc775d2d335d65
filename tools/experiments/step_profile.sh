#!/bin/bash
# rocprofv3 per-kernel summary of whole CNLinear time steps (examples/flow_configs.c)
#   bash tools/experiments/step_profile.sh 512 channel  -ns_ksp_rtol 1e-4 -ns_ksp_gmres_restart 12     (GMRES outer solve)
#   bash tools/experiments/step_profile.sh 512 cylinder -ns_ksp_type preonly                           (fractional step + IBM)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/stepprof
N=${1:-512}
CFG=${2:-channel}
shift 2
mkdir -p $O
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o k -- $R/fluca_amd/lib/flow_configs -config $CFG -n $N -ns_max_steps 3 -ns_abf_schur_pc_type mg "$@" > $O/trace.log 2>&1
echo rc=$?
grep "^step\|^config" $O/trace.log
