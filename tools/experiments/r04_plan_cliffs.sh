#!/bin/bash
# is there a cliff left?  the CG pair's default tiling against its neighbours on grid sizes between the measured cubes
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_plan_cliffs.txt
: > $O
for n in 288 352 416 448 480; do
  t=$(( ((n+127)/128) * ((n+15)/16) ))
  lo=$(( 256 / t )); [ $lo -lt 1 ] && lo=1
  for plan in "" 2,8,$lo 2,8,$((lo+1)) 2,8,$((2*lo)) 2,8,$((2*lo+1)) 2,4,$((2*lo)) 2,4,$((4*lo)); do
    FLUCA_CG_PLAN=$plan timeout -k 10 100 python3 $R/tools/cg_rate.py --cells $n --iters 200 --reps 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); n=d['cells']; print('n=%d tiles16=%d FLUCA_CG_PLAN=%-7s %8.1f it/s  %.4f ms  %.2f TB/s moved (59 B/cell)' % (n, $t, '$plan', d['its_per_s'], d['ms_per_iter'], 59*n**3/d['ms_per_iter']/1e9))" >> $O || exit 1
  done
done
cat $O
