#!/bin/bash
# round 5 evidence: rocprofv3 per-kernel summary of the bench command, FETCH_SIZE / WRITE_SIZE passes of it (separate processes), and the
# counter passes of the momentum operator (tools/prof/pmc_kernel.sh over tools/mom_bench.py).  Raw files under gpurun_out/r05_prof*;
# summarised on the development box by tools/experiments/pmc_summary.py and tools/prof/pmc_mom3_json.py (they stamp the kernels' source hashes).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_prof
mkdir -p $O
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o k -- python3 $R/bench.py --steps 100 --warmup 10 --skip-cpu > $O/bench_under_rocprof.json 2> $O/trace.err
echo "trace rc=$?"
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o k -- python3 $R/bench.py --steps 8 --warmup 2 --skip-cpu --skip-extras > $O/bench_fetch.json 2> $O/fetch.err
echo "fetch rc=$?"
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o k -- python3 $R/bench.py --steps 8 --warmup 2 --skip-cpu --skip-extras > $O/bench_write.json 2> $O/write.err
echo "write rc=$?"
cd $R
bash tools/prof/pmc_kernel.sh r05_mom3_pmc $R/tools/mom_bench.py --cells 512 --fly 1 --nosolve --reps 5 --dif 0 && python tools/prof/pmc_table.py gpurun_out/r05_mom3_pmc k_mom3 1.0 > gpurun_out/r05_mom3_pmc/table.json
ls $O/trace $O/fetch $O/write gpurun_out/r05_mom3_pmc | head -30
