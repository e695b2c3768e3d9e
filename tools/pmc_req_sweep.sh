# L1->L2 request counts of the window kernel per stop-phase (8192-window bench step): where the transactions come from
cd /root/repo; mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
for st in 3 4 5 6 7 8 0; do
  d=/root/repo/gpurun_out/req_$st; rm -rf $d
  LANCET_STOP_PHASE=$st timeout 300 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum --output-format csv -d $d -- python /root/repo/bench.py --steps 1 --warmup 0 --cpu-sample 0 --windows 8192 > $d.log 2>&1
  grep window_kernel $d/*/*counter_collection.csv | awk -F, -v st=$st '{n=NF; printf "stop %s %s %.4g\n", st, $(n-3), $(n-2)}'
done
