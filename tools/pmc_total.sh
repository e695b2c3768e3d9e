# HBM-side traffic of the kernels for the default bench workload: two passes (FETCH_SIZE, WRITE_SIZE), MB per launch.
# (rocprofv3 --pmc serialises kernels: the build service gives up after 300 ms without progress and the window slots build those
#  graphs themselves -- the traffic of a batch is unchanged by who builds a graph, the kernel times under PMC are not the bench's.)
cd /root/repo; mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  d=/root/repo/gpurun_out/pmc_$c; rm -rf $d
  timeout ${PMC_TIMEOUT:-150} rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- python /root/repo/bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-configs --in-flight 1 $BENCH_ARGS > $d.log 2>&1 || echo "pass $c: rc $?"
  for k in "build_kernel(" "build_kernel_large(" "window_kernel(" "svc_kernel(" "prep_kernel("; do
    grep -F "$k" $d/*/*counter_collection.csv | awk -F, -v k=$k -v c=$c '{n=NF; v[NR]=$(n-2); t[NR]=($(n)-$(n-1))/1e6} END {if (NR) { s=0; for (i=1;i<=NR;i++) s+=v[i]; printf "%s %s mean %.1f MB over %d launches (last %.1f MB, %.1f ms)\n", k, c, s/NR/1024, NR, v[NR]/1024, t[NR]} }'
  done
done
