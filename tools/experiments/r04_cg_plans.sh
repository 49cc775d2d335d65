#!/bin/bash
# Tiling of k_cg_A / k_cg_Bq in the solver across grid sizes (FLUCA_CG_PLAN=ry,nw,nchunk; FLUCA_CGBQ_CHUNKS for k_cg_Bq alone).
# Output: gpurun_out/r04_cg_plans.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_cg_plans.txt
: > $O
run() { FLUCA_CG_PLAN=$2 FLUCA_CGBQ_CHUNKS=$3 timeout -k 10 120 python3 $R/tools/cg_rate.py --cells $1 --iters $4 --reps 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('n=%4d plan=%-8s bq_chunks=%-3s %9.4f ms/it %9.1f it/s' % (d['cells'], d['env'].get('FLUCA_CG_PLAN',''), d['env'].get('FLUCA_CGBQ_CHUNKS',''), d['ms_per_iter'], d['its_per_s']))" >> $O || exit 1; }
for n in 128 192 256 320 384 512; do
  its=400; [ $n -ge 384 ] && its=200
  run $n "" "" $its
  for plan in 2,8,2 2,8,3 2,8,4 2,8,6 2,8,8 2,8,12 2,8,16 2,4,2 2,4,4 2,4,6 2,4,8 2,4,12 2,4,16 1,4,2 1,4,4 1,4,6 1,4,8; do
    # keep the cases with 192 .. 1100 blocks
    nb=$(python3 -c "
ry,nw,nc=map(int,'$plan'.split(','))
n=$n
print(((n+127)//128)*((n+ry*nw-1)//(ry*nw))*nc)")
    [ $nb -ge 192 ] && [ $nb -le 1100 ] && run $n $plan "" $its
  done
done
for bq in 4 8 16; do run 256 2,8,8 $bq 400; done
for bq in 4 8 16; do run 256 2,4,8 $bq 400; done
cat $O
