#!/bin/bash
# does the arena placement still pay with the q-free CG pair?  same box, alternating
for pl in 1 0 1 0; do
  echo "== FLUCA_PLACEMENT=$pl"
  FLUCA_PLACEMENT=$pl python3 tools/experiments/cg_variants.py 512 2>/dev/null | grep "variant="
done
