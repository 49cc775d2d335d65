#!/bin/bash
# kernel durations vs iteration time of Jacobi-PCG at 256^3 (BASELINE config 2)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/cg256
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o k -- python3 $R/tools/ksp_bench.py 256 > $O/trace.log 2>&1
echo rc=$?
grep "n= 256" $O/trace.log
