# rocprofv3 kernel-trace stats of one-batch-at-a-time bench steps (the roofline's kernels alone on the GPU)
TAG=${1:-kt_x}
mkdir -p /root/repo/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/$TAG/kt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/$TAG/kt -- python /root/repo/bench.py --settle 0 --cpu-sample 0 --no-configs --in-flight 1 > /root/repo/gpurun_out/$TAG/kt.log 2>&1
cat /root/repo/gpurun_out/$TAG/kt/*/*kernel_stats.csv | head -8 | cut -c1-40,150-400 | tee /root/repo/gpurun_out/$TAG/kernel_trace_stats.txt
