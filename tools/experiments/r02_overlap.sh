#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02d
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_rccl_loopback.py tests/test_gpu_bench_contract.py -x -q > $O/tests.log 2>&1
rc=$?
echo "tests rc=$rc"; tail -4 $O/tests.log
[ $rc -eq 0 ] || exit $rc
for ov in 1 0; do
  FLUCA_OVERLAP=$ov timeout -k 10 200 python tools/experiments/loopback_bench.py --cells 512 --axes 3 > $O/loopback_ov$ov.json 2> $O/loopback_ov$ov.err
  echo "overlap=$ov rc=$?"; cat $O/loopback_ov$ov.json
done
