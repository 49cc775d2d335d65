#!/bin/bash
# k_cheb2 tile height: 128 x 16 (one 512-thread block per CU) against 128 x 8 (two 256-thread blocks per CU), C3 at 512^3
for cfg in "8 0" "4 2" "4 4" "8 0" "4 2" "4 3"; do
  set -- $cfg
  echo "== FLUCA_CHEB2_NW=$1 FLUCA_CHEB2_NCHUNK=$2"
  FLUCA_CHEB2_NW=$1 FLUCA_CHEB2_NCHUNK=$2 python3 tools/cheb_bench.py 512 100 2>/dev/null | grep -i "fuse\|steps/s" | tail -2
done
