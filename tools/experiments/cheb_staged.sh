#!/bin/bash
# one-step Chebyshev-Jacobi kernel: LDS-staged walk (k_cheb_st, default) against round 1's k_cheb, C3 at 512^3 and 256^3
for n in 512 256; do for st in 0 1 0 1; do
  echo -n "n=$n FLUCA_CHEB_STAGED=$st: "
  FLUCA_CHEB_STAGED=$st python3 tools/cheb_bench.py $n 100 2>/dev/null | grep "fuse=0" | tail -1
done; done
