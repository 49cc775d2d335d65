#!/bin/bash
# x-update of the q-free CG pair: batched (two updates every second iteration, default) against one per iteration
for xb in 1 0 1 0; do
  echo "== FLUCA_CG_XBATCH=$xb"
  FLUCA_CG_XBATCH=$xb python3 tools/experiments/cg_variants.py 128 256 512 2>/dev/null | grep "variant=0"
done
