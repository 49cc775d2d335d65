#!/bin/bash
# round 5: k_schur_var against the number of blocks per XCD (and the occupancy bound of the kernel): kernel time and fetched bytes per cell at 512^3.
# The rows an XCD covers per loop trip must divide its band of y (64 rows at 512^3), or its waves straddle two planes and the z window of p leaves the L2.
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for cfg in ${CONFIGS:-64:1 48:1 32:1}; do
  set -- ${cfg/:/ }
  (cd $R && FL_DEFINES="FL_SV_BLOCKS_PER_XCD=$1 FL_SV_MINBLOCKS=$2" python -m fluca_amd.build > /dev/null 2>&1)
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/svs_t$1_$2 -o p -- python3 $R/tools/schur_var_bench.py 512 > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/svs_f$1_$2 -o p -- python3 $R/tools/schur_var_bench.py 512 > /dev/null 2>&1
  python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("$R/gpurun_out/svs_t$1_$2/p_kernel_stats.csv")) if "schur_var" in r["Name"]]
acc=[float(r["Counter_Value"]) for r in csv.DictReader(open("$R/gpurun_out/svs_f$1_$2/p_counter_collection.csv")) if "schur_var" in r["Kernel_Name"]]
print("blocks per XCD $1, min blocks per CU $2: %.3f ms, fetched %.1f B/cell" % (float(rows[0]["AverageNs"])/1e6, max(acc)*2*1024/512**3))
PY
done
