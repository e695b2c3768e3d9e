# A few PMC passes over the window kernel (8192-window bench step) to see what the waves wait for.  Every pass under its own timeout.
cd /root/repo; mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_SALU" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum" \
           "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum GRBM_GUI_ACTIVE" \
           ${PMC_MORE:+"TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum"} \
           ${PMC_MORE:+"TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCC_TAG_STALL_sum TCP_TOTAL_CACHE_ACCESSES_sum"}; do
  i=$((i+1)); d=/root/repo/gpurun_out/probe_$i; rm -rf $d
  timeout ${PMC_TIMEOUT:-90} rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -- python /root/repo/bench.py --steps 1 --warmup 0 --cpu-sample 0 --windows 8192 --no-configs --in-flight 1 > $d.log 2>&1 || echo "pass $i: rc $?"
  for k in window_kernel build_kernel; do grep "$k" $d/*/*counter_collection.csv 2>/dev/null | awk -F, -v k=$k '{n=NF; printf "%s %s %.4g\n", k, $(n-3), $(n-2)}' | sort | uniq -c | head -12; done
done
