#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02c
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_poisson.py tests/test_gpu_mg.py tests/test_gpu_ksp.py tests/test_gpu_timestep.py -x -q > $O/tests.log 2>&1
rc=$?
echo "tests rc=$rc"; tail -4 $O/tests.log
[ $rc -eq 0 ] || exit $rc
for t in 512 256; do
  FLUCA_CGA_TARGET=$t timeout -k 10 200 python bench.py --steps 20 --warmup 5 --skip-cpu --skip-extras --skip-configs > $O/bench_t$t.json 2> $O/bench_t$t.err
  echo "target $t rc=$?"; python -c "
import json;d=json.load(open('$O/bench_t$t.json'));print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['placement'])"
  FLUCA_CGA_TARGET=$t timeout -k 10 200 python bench.py --steps 200 --warmup 20 --skip-cpu --skip-extras --skip-configs > $O/bench200_t$t.json 2> $O/bench200_t$t.err
  python -c "
import json;d=json.load(open('$O/bench200_t$t.json'));print('K=200', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done
