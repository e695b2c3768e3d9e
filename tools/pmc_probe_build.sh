# PMC passes over the LDS build kernel (default bench step, one batch at a time): instruction mix, what the waves wait for, LDS conflicts.
cd /root/repo; mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_INSTS_WAVE32_LDS SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_WAVES SQ_LEVEL_WAVES"; do
  i=$((i+1)); d=/root/repo/gpurun_out/probeb_$i; rm -rf $d
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -- python /root/repo/bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-configs --in-flight 1 > $d.log 2>&1
  for k in build_kernel window_kernel; do
    grep $k $d/*/*counter_collection.csv | awk -F, -v k=$k '{n=NF; a[$(n-3)]+=$(n-2); c[$(n-3)]++} END {for (x in a) printf "%s %s %.4g\n", k, x, a[x]/c[x]}'
  done
done
