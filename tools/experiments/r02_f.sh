#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02f
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_cheb2.py tests/test_gpu_ksp.py tests/test_gpu_mg.py -x -q > $O/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -3 $O/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python tools/cheb_bench.py 512 100 2>&1 | grep -v amdgpu | tee $O/cheb_bench.log
