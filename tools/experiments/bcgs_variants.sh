#!/bin/bash
# BiCGStab on S: products never stored (default) against V0 / T0 stored (FLUCA_BCGS_VARIANT=2), same box, alternating
for v in 2 0 2 0; do
  echo "== FLUCA_BCGS_VARIANT=$v"
  FLUCA_BCGS_VARIANT=$v python3 tools/ksp_bench.py 128 256 512 2>/dev/null | grep "bcgs"
done
