#!/bin/bash
# rocprofv3 per-kernel summary of the multigrid-PCG solve at 512^3 (tools/mg_bench.py --skip-jacobi)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/mgprof
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o k -- python3 $R/tools/mg_bench.py --cells 512 --skip-jacobi > $O/trace.log 2>&1
echo rc=$?
cat $O/trace.log | tail -2
