#!/bin/bash
# the round-4 block-count rule of plan_cg_A against round 2's (forced through FLUCA_CG_PLAN) on every solver that walks that plan
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_plan_check.txt
: > $O
for n in 256 384; do
  old=2,4,4; [ $n = 384 ] && old=2,4,2
  for plan in "" $old; do
    echo "# ksp_bench $n FLUCA_CG_PLAN=$plan" >> $O
    FLUCA_CG_PLAN=$plan timeout -k 10 200 python3 $R/tools/ksp_bench.py $n 2>/dev/null | grep "n= *$n" >> $O || exit 1
  done
done
for plan in "" 2,4,4; do
  echo "# mg_bench 512^3 (level 1 is 256^3) FLUCA_CG_PLAN=$plan" >> $O
  FLUCA_CG_PLAN=$plan timeout -k 10 200 python3 $R/tools/mg_bench.py 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('mg iters %d seconds %.5f' % (d['mg']['iters'], d['mg']['seconds']))" >> $O || exit 1
done
cat $O
