#!/bin/bash
# k_mom_apply, careful A/B of the three candidates of mom_order.sh (alternating, three rounds)
for r in 1 2 3; do for cfg in "0 4" "0 8" "1 4" "1 8"; do
  set -- $cfg
  echo -n "round $r order $1 chunks $2: "
  FLUCA_MOM_ORDER=$1 FLUCA_MOM_CHUNKS=$2 python3 tools/mom_bench.py --cells 512 --reps 20 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('apply+pad %.3f ms  bcgs it %.3f ms  stream %.3f' % (d['apply_ms_incl_pad'], d['ms_per_iter'], d['stream15r3w_ms_2048']))"
done; done
