#!/bin/bash
# rocprofv3 per-kernel summary of the bench command (profiles/r01_rocprof)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/benchprof
mkdir -p $O
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o k -- python3 $R/bench.py --steps 100 --warmup 10 --skip-cpu > $O/bench_under_rocprof.json 2> $O/err.log
echo rc=$?
tail -c 1500 $O/bench_under_rocprof.json
