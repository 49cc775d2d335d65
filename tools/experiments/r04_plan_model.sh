#!/bin/bash
# does the cost model  rounds x (planes per block + prologue) x (1 + 0.04 (rounds - 1))  predict the tilings?  384^3 and two non-cubic grids
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_plan_model.txt
: > $O
for rep in 1 2; do
  for plan in "" 2,8,3 2,8,7 2,8,6 2,8,14 2,8,10; do
    FLUCA_CG_PLAN=$plan timeout -k 10 100 python3 $R/tools/cg_rate.py --cells 384 --iters 200 --reps 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('cg 384^3 FLUCA_CG_PLAN=%-7s %.1f it/s (%.4f ms)' % ('$plan', d['its_per_s'], d['ms_per_iter']))" >> $O || exit 1
  done
  for nc in "" 3 7 6 10; do
    FLUCA_CHEB2_NCHUNK=$nc timeout -k 10 100 python3 $R/tools/ksp_bench.py 384 2>/dev/null | grep "n= 384 cavity   chebyshev" | sed "s/^/cheb2 FLUCA_CHEB2_NCHUNK=$nc  /" >> $O || exit 1
  done
done
cat $O
