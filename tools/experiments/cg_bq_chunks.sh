#!/bin/bash
# z chunks of k_cg_Bq at 256^3 / 512^3 (variant 0 only): FLUCA_CGBQ_CHUNKS = 0 (k_cg_A's), 1, 2, 4, 8
for c in 0 1 2 4 8; do
  echo "== FLUCA_CGBQ_CHUNKS=$c"
  FLUCA_CGBQ_CHUNKS=$c python3 tools/experiments/cg_variants.py 256 512 | grep "variant=0"
done
