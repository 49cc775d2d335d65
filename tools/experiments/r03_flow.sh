#!/bin/bash
# round 3: whole time steps of the BASELINE set-ups with the round's kernels (k_mom3 through NSStep, the shortened multigrid cycle)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_flow_configs.txt
echo "# round 3: examples/flow_configs.c on one MI355X; -ns_ksp_type preonly -ns_abf_schur_pc_type mg  (fractional step: one PCApply_ABF per step)" > $O
for c in sphere cylinder; do
  timeout -k 10 300 fluca_amd/lib/flow_configs -config $c -n 512 -ns_max_steps 4 -ns_ksp_type preonly -ns_abf_schur_pc_type mg >> $O 2>&1 || exit 1
done
echo "# the same steps with the stored-path momentum kernel (FLUCA_MOM_KERNEL=2)" >> $O
FLUCA_MOM_KERNEL=2 timeout -k 10 300 fluca_amd/lib/flow_configs -config sphere -n 512 -ns_max_steps 4 -ns_ksp_type preonly -ns_abf_schur_pc_type mg >> $O 2>&1 || exit 1
echo "# default outer GMRES to 1e-4" >> $O
timeout -k 10 300 fluca_amd/lib/flow_configs -config cavity -n 256 -ns_max_steps 3 -ns_abf_schur_pc_type mg >> $O 2>&1
cat $O
