#!/bin/bash
# round 3: the driver's bench command (placement verbose), then the new full-size and contract tests
cd $GRAFT_REPO_ROOT
FLUCA_PLACEMENT_VERBOSE=1 python bench.py --steps 20 --warmup 20 > gpurun_out/r03_bench_a.json 2> gpurun_out/r03_bench_a.err; echo "bench rc=$?"
tail -c 1500 gpurun_out/r03_bench_a.err
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_poisson.py tests/test_gpu_bench_contract.py -x -q -k "fullsize or c3_ or c4_ or placement or single_gpu_line or small_handles" > gpurun_out/r03_tests_a.log 2>&1; tail -15 gpurun_out/r03_tests_a.log
