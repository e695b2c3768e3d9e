# Round profile set on the GPU box: bench line, rocprofv3 kernel-trace stats of the same command (one batch at a time on the GPU,
# which is what the roofline figures of the bench line are measured on), PMC traffic passes, kernel resources, the re-run tier's kernel.
TAG=${1:-r3_x}
cd /root/repo; mkdir -p gpurun_out/$TAG
timeout 600 python bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err; tail -1 gpurun_out/$TAG/bench.json | cut -c1-300
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/$TAG/kt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/$TAG/kt -- python /root/repo/bench.py --cpu-sample 0 --no-configs --in-flight 1 > /root/repo/gpurun_out/$TAG/kt.log 2>&1
cat /root/repo/gpurun_out/$TAG/kt/*/*kernel_stats.csv | head -14 | tee /root/repo/gpurun_out/$TAG/kernel_trace_stats.csv
rm -rf /root/repo/gpurun_out/$TAG/ktfat
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/$TAG/ktfat -- python /root/repo/tools/fat_probe.py > /root/repo/gpurun_out/$TAG/fat_probe.log 2>&1
tail -3 /root/repo/gpurun_out/$TAG/fat_probe.log
cat /root/repo/gpurun_out/$TAG/ktfat/*/*kernel_stats.csv | head -10 | tee /root/repo/gpurun_out/$TAG/fat_kernel_trace_stats.csv
PMC_TIMEOUT=200 BENCH_ARGS="" bash /root/repo/tools/pmc_total.sh | tee /root/repo/gpurun_out/$TAG/pmc.txt
bash /root/repo/tools/kernel_resources.sh > /root/repo/gpurun_out/$TAG/kernel_resources.txt 2>&1; grep -A5 "window_kernel\|build_kernelP" /root/repo/gpurun_out/$TAG/kernel_resources.txt | head -30
