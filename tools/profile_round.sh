# Round profile set on the GPU box: bench line, rocprofv3 kernel-trace stats of the same command (one batch at a time on the GPU,
# which is what the roofline figures of the bench line are measured on), PMC traffic passes.
TAG=${1:-r2_x}
cd /root/repo; mkdir -p gpurun_out/$TAG
python bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err; tail -1 gpurun_out/$TAG/bench.json | cut -c1-400
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/$TAG/kt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/$TAG/kt -- python /root/repo/bench.py --cpu-sample 0 --no-configs --in-flight 1 > /root/repo/gpurun_out/$TAG/kt.log 2>&1
cat /root/repo/gpurun_out/$TAG/kt/*/*kernel_stats.csv | head -12 | tee /root/repo/gpurun_out/$TAG/kernel_trace_stats.csv
BENCH_ARGS="--in-flight 1" bash /root/repo/tools/pmc_total.sh | tee /root/repo/gpurun_out/$TAG/pmc.txt
