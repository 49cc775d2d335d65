#!/bin/bash
# round 3: the momentum operator kernel -- counters of k_mom2, then plan / hint variants, then the no-arithmetic build
R=$GRAFT_REPO_ROOT
cd $R
$R/tools/prof/pmc_kernel.sh r03_mom_pmc $R/tools/mom_bench.py --nosolve --modes 4 --reps 3 || exit 1
python tools/prof/pmc_table.py gpurun_out/r03_mom_pmc k_mom 2.0 > gpurun_out/r03_mom_pmc_table.json
for v in "FLUCA_MOM_CHUNKS=1" "FLUCA_MOM_CHUNKS=2" "FLUCA_MOM_CHUNKS=4" "FLUCA_MOM_CHUNKS=8" "FLUCA_MOM_NT=0" "FLUCA_MOM_ORDER=0"; do
  env $v python tools/mom_bench.py --nosolve --modes 1 --reps 10 2>/dev/null | tee -a gpurun_out/r03_mom_variants.txt
done
touch fluca_amd/csrc/fl_momentum.hip
FL_DEFINES="FL_MOM2_ABL" python -c "from fluca_amd import build; build.build()" > gpurun_out/r03_abl_build.log 2>&1 || exit 1
echo "ABL build" | tee -a gpurun_out/r03_mom_variants.txt
python tools/mom_bench.py --nosolve --modes 1 --reps 10 2>/dev/null | tee -a gpurun_out/r03_mom_variants.txt
