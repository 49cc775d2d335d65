#!/bin/bash
# k_cg_Bq on a tiling of its own (FLUCA_CGBQ_PLAN) beside k_cg_A's round-4 plan, 256^3 and 384^3
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_bq_plan.txt
: > $O
for rep in 1 2; do
for n in 256 384; do
  for bq in "" 2,4,4 2,4,8 1,4,4 1,4,8 2,8,4 2,4,2 2,4,3 2,4,6 2,8,6 2,8,3; do
    FLUCA_CGBQ_PLAN=$bq timeout -k 10 100 python3 $R/tools/cg_rate.py --cells $n --iters 300 --reps 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('n=%d FLUCA_CGBQ_PLAN=%-6s %.1f it/s (%.4f ms)' % (d['cells'], '$bq', d['its_per_s'], d['ms_per_iter']))" >> $O || exit 1
  done
done
done
cat $O
