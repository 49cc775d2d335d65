#!/bin/bash
# round 3 evidence: rocprofv3 per-kernel summary of the bench command, then FETCH_SIZE / WRITE_SIZE passes (separate processes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_prof
mkdir -p $O
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o k -- python3 $R/bench.py --steps 100 --warmup 10 --skip-cpu > $O/bench_under_rocprof.json 2> $O/trace.err
echo "trace rc=$?"
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o k -- python3 $R/bench.py --steps 8 --warmup 2 --skip-cpu --skip-extras > $O/bench_fetch.json 2> $O/fetch.err
echo "fetch rc=$?"
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o k -- python3 $R/bench.py --steps 8 --warmup 2 --skip-cpu --skip-extras > $O/bench_write.json 2> $O/write.err
echo "write rc=$?"
ls -la $O/trace $O/fetch $O/write | head -20
