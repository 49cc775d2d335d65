#!/bin/bash
# round 4: KSPCHEBYSHEV on the momentum block against Jacobi-BiCGStab -- the solver alone (tools/mom_bench.py) and whole 512^3 time steps
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_mom_cheb.txt
echo "# tools/mom_bench.py --cells 512 (state handed over with v0: k_mom3; --dif 2.56 = nu dt / h^2 of the 512^3 flow configurations)" > $O
timeout -k 10 400 python tools/mom_bench.py --cells 512 --reps 5 >> $O 2>&1 || { tail -5 $O; exit 1; }
echo "# examples/flow_configs.c -config sphere -n 512, fractional step, multigrid on S; momentum block: default (bcgs + jacobi) / chebyshev" >> $O
timeout -k 10 300 fluca_amd/lib/flow_configs -config sphere -n 512 -ns_max_steps 4 -ns_ksp_type preonly -ns_abf_schur_pc_type mg >> $O 2>&1 || { tail -5 $O; exit 1; }
timeout -k 10 300 fluca_amd/lib/flow_configs -config sphere -n 512 -ns_max_steps 4 -ns_ksp_type preonly -ns_abf_schur_pc_type mg -ns_abf_momentum_ksp_type chebyshev >> $O 2>&1 || { tail -5 $O; exit 1; }
cat $O
