#!/bin/bash
# register-budget sweep of k_mom_apply on the GPU box: rebuilds fl_momentum.hip with 2, 3, 4 waves/SIMD and times it
set -e
for w in 2 3; do
  FL_MOM_WPE=$w python -c "
import os
from fluca_amd import build as b
os.remove(os.path.join(b.LIBDIR, 'fl_momentum.hip.o'))
b.build()"
  echo "WPE=$w"
  timeout -k 10 200 python tools/mom_bench.py --cells 512
done
