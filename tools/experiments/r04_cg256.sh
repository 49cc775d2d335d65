#!/bin/bash
# BASELINE config 2 (256^3 Jacobi-PCG): tiling sweep in the solver (FLUCA_CG_PLAN), then the kernel trace of the default with the idle
# time at the two kernel boundaries of an iteration.  Output: gpurun_out/r04_cg256.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_cg256.txt
: > $O
echo "# tools/cg_rate.py --cells 256 (best of 3 x 400 iterations); FLUCA_CG_PLAN=ry,nw,nchunk (empty = the shipped plan)" >> $O
for plan in "" 2,4,4 2,4,2 2,4,8 2,4,16 2,8,4 2,8,8 2,8,16 2,8,32 1,4,2 1,4,4 1,4,8 1,4,16; do
  FLUCA_CG_PLAN=$plan timeout -k 10 120 python3 $R/tools/cg_rate.py --cells 256 >> $O 2>/dev/null || exit 1
done
echo "# single-reduction CG, shipped plan" >> $O
timeout -k 10 120 python3 $R/tools/cg_rate.py --cells 256 --single-reduction 1 >> $O 2>/dev/null || exit 1
cd /tmp && export TMPDIR=/tmp
D=$R/gpurun_out/r04_cg256_trace
rm -rf $D && mkdir -p $D
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o k -- python3 $R/tools/cg_rate.py --cells 256 --reps 1 > $D/run.log 2>&1 || exit 1
echo "# rocprofv3 --kernel-trace of the shipped plan: idle time at kernel boundaries" >> $O
python3 $R/tools/prof/kernel_gaps.py $(find $D -name "*kernel_trace.csv" | head -1) 100 >> $O
cat $O
