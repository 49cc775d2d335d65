#!/bin/bash
# round 2, first GPU call: parity of the fused two-step Chebyshev kernel, then its rate at 512^3
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02a
mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_cheb2.py tests/test_gpu_ksp.py tests/test_gpu_mg.py -x -q > $O/tests.log 2>&1
echo "tests rc=$?" | tee -a $O/tests.log
tail -5 $O/tests.log
timeout -k 10 200 python tools/cheb_bench.py 512 100 > $O/cheb_bench.log 2>&1
echo "bench rc=$?"
cat $O/cheb_bench.log
for nc in 1 4 8; do
  FLUCA_CHEB2_NCHUNK=$nc timeout -k 10 200 python tools/cheb_bench.py 512 100 > $O/cheb_bench_nc$nc.log 2>&1
  echo "nchunk=$nc"; grep "fuse=2" $O/cheb_bench_nc$nc.log
done
