#!/bin/bash
# rocprofv3 summary + HBM-traffic counters of the Chebyshev-Jacobi step (BASELINE config 3: 512^3 channel) and of the other
# Krylov kernels, from tools/ksp_bench.py 512
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/chebprof
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o k -- python3 $R/tools/ksp_bench.py 512 > $O/trace.log 2>&1 && \
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o k -- python3 $R/tools/ksp_bench.py 512 > $O/fetch.log 2>&1 && \
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o k -- python3 $R/tools/ksp_bench.py 512 > $O/write.log 2>&1
echo rc=$?
grep "n= 512" $O/trace.log
