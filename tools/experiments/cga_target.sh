#!/bin/bash
# block-count target of plan_cg_A (k_cg_A and k_cg_Bq walk the same tiles) with the final kernel pair, 256^3 and 512^3
for t in 256 512 1024 256 512; do
  echo "== FLUCA_CGA_TARGET=$t"
  FLUCA_CGA_TARGET=$t python3 tools/experiments/cg_variants.py 256 512 2>/dev/null | grep "variant=0"
done
