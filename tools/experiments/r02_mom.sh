#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_mom
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_momentum.py tests/test_host_mirror.py tests/test_gpu_timestep.py -x -q > $O/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -3 $O/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python tools/mom_bench.py --cells 512 --stored > $O/mom_stored.json 2> $O/err1.log; cat $O/mom_stored.json
timeout -k 10 200 python tools/mom_bench.py --cells 512 > $O/mom_v0.json 2> $O/err2.log; cat $O/mom_v0.json
