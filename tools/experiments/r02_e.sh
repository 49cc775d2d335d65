#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02e
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_poisson.py tests/test_gpu_mg.py tests/test_gpu_ksp.py tests/test_gpu_multirank.py tests/test_gpu_rccl_loopback.py -x -q > $O/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -3 $O/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r02e/bench.json"))
print("value", d["value"], "ms/step", d["ms_per_step"], "k_cg_A", d["roofline"]["avg_launch_ms"], "unplaced", d.get("value_unplaced"))
for k,v in d["configs"].items(): print(k, v["value"], v["ms_per_step"], v["roofline"]["frac"])
print("tts", d["time_to_solution"])
PY
(for n in 128 256 512; do echo "## $n^3"; timeout -k 10 200 python tools/experiments/sweep256.py $n 2>&1 | grep -v amdgpu; done) > $O/plan_sweep.txt
