# HBM-side traffic of the window kernel for the default bench workload: two passes (FETCH_SIZE, WRITE_SIZE), KB per launch
cd /root/repo; mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  d=/root/repo/gpurun_out/pmc_$c; rm -rf $d
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- python /root/repo/bench.py --steps 1 --warmup 0 --cpu-sample 0 $BENCH_ARGS > $d.log 2>&1
  grep window_kernel $d/*/*counter_collection.csv | awk -F, '{n=NF; printf "%s %.1f MB  %.1f ms\n", $(n-3), $(n-2)/1024, ($(n)-$(n-1))/1e6}'
done
