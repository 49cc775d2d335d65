#!/bin/bash
# rocprofv3 summary + HBM-traffic counters of the momentum block (tools/mom_bench.py, 512^3)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/momprof2
mkdir -p $O
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o mom -- python3 $R/tools/mom_bench.py --cells 512 --reps 10 > $O/trace.log 2>&1 && \
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o mom -- python3 $R/tools/mom_bench.py --cells 512 --reps 3 > $O/fetch.log 2>&1 && \
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o mom -- python3 $R/tools/mom_bench.py --cells 512 --reps 3 > $O/write.log 2>&1
echo rc=$?
grep cells $O/trace.log
