#!/bin/bash
# k_mom_apply: block order (0 chunk fastest, 1 chunk-major + XCD-contiguous) x z chunks, 512^3
for o in 0 1; do for c in 4 2 1 8; do
  echo "== FLUCA_MOM_ORDER=$o FLUCA_MOM_CHUNKS=$c"
  FLUCA_MOM_ORDER=$o FLUCA_MOM_CHUNKS=$c python3 tools/mom_bench.py --cells 512 2>/dev/null | tail -1
done; done
