#!/bin/bash
# config 2 inside the bench process (after the 512^3 headline, on the bench's stream) with round 2's plan against round 4's, and stand-alone
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_c2_in_bench.txt
: > $O
for plan in "" 2,4,4 "" 2,4,4; do
  FLUCA_CG_PLAN=$plan timeout -k 10 300 python3 $R/bench.py --steps 20 --warmup 5 --skip-cpu --skip-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['configs']['C2']
print('bench: FLUCA_CG_PLAN=%-6s headline %.1f it/s  C2 %.1f it/s (%.4f ms)  k_cg_A %.4f ms  k_cg_Bq %.4f ms' % ('$plan', d['value'], c['value'], c['ms_per_step'], c['roofline']['kernels'][0]['avg_launch_ms'], c['roofline']['avg_launch_ms']))" >> $O || exit 1
  FLUCA_CG_PLAN=$plan timeout -k 10 100 python3 $R/tools/cg_rate.py --cells 256 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('alone: FLUCA_CG_PLAN=%-6s %.1f it/s (%.4f ms)' % ('$plan', d['its_per_s'], d['ms_per_iter']))" >> $O || exit 1
done
cat $O
