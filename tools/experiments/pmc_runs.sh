R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum GRBM_UTCL2_BUSY" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_sum" "TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --output-format csv -d $R/gpurun_out/pmcx_$i -- python3 $R/bench.py --steps 8 --warmup 2 --skip-cpu > /dev/null 2>&1
  echo "pass $i: $ctrs rc=$?"
done
