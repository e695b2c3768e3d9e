# HBM-side traffic of the kernels for the default bench workload: two passes (FETCH_SIZE, WRITE_SIZE), KB per launch
cd /root/repo; mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  d=/root/repo/gpurun_out/pmc_$c; rm -rf $d
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- python /root/repo/bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-configs $BENCH_ARGS > $d.log 2>&1
  for k in "build_kernel(" "build_kernel_large(" "window_kernel(" "prep_kernel("; do
    grep -F "$k" $d/*/*counter_collection.csv | awk -F, -v k=$k -v c=$c '{n=NF; printf "%s %s %.1f MB  %.1f ms\n", k, c, $(n-2)/1024, ($(n)-$(n-1))/1e6}'
  done
done
