cd /root/repo; mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
for st in 3 4 5 6 7 8 0; do for c in FETCH_SIZE WRITE_SIZE; do
  d=/root/repo/gpurun_out/stop_${st}_$c; rm -rf $d
  LANCET_STOP_PHASE=$st timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- python /root/repo/bench.py --steps 1 --warmup 0 --cpu-sample 0 > $d.log 2>&1
  grep window_kernel $d/*/*counter_collection.csv | awk -F, -v st=$st '{n=NF; printf "stop %s %s %.1f MB  %.1f ms\n", st, $(n-3), $(n-2)/1024, ($(n)-$(n-1))/1e6}'
done; done
