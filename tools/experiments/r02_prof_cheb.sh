#!/bin/bash
# rocprofv3: per-kernel durations and HBM counters of the Chebyshev-Jacobi sweeps of BASELINE config 3 (tools/cheb_bench.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_chebprof
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o k -- python3 $R/tools/cheb_bench.py 512 40 > $O/trace.log 2>&1 && \
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o k -- python3 $R/tools/cheb_bench.py 512 40 > $O/fetch.log 2>&1 && \
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o k -- python3 $R/tools/cheb_bench.py 512 40 > $O/write.log 2>&1
echo rc=$?
grep "fuse=" $O/trace.log
ls $O/trace $O/fetch | head
