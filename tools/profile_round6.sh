# Round-6 profile set on the GPU box (profiles/README.md): the default bench line, rocprofv3 --kernel-trace --stats of the same workload with
# one batch at a time on the GPU (what roofline.kernel_ms is measured on), the PMC traffic passes (FETCH_SIZE / WRITE_SIZE, separate passes)
# for the headline workload AND the three side configurations, kernel resources, the end-to-end scans (500 kb, and the 5 Mb contig when its
# BAMs are there), the SQ counter passes of the build kernel.
TAG=${1:-r6_final}
cd /root/repo; O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | cut -c1-300
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/$O/kt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/kt -- python /root/repo/bench.py --steps 20 --settle 0 --cpu-sample 0 --no-configs --no-bam --in-flight 1 > /root/repo/$O/kt.log 2>&1
cat /root/repo/$O/kt/*/*kernel_stats.csv | head -14 > /root/repo/$O/kernel_trace_stats.csv; head -8 /root/repo/$O/kernel_trace_stats.csv
for cfg in "headline|" "60x|--windows 8192 --cov 60" "config4|--windows 4096 --cov 100 --cov-normal 40 --str-fraction 0.3 --lowcomplex-fraction 0.05" "config5|--windows 16384 --linked"; do
  name=${cfg%%|*}; args=${cfg#*|}
  echo "== PMC passes: $name ($args)" | tee -a /root/repo/$O/pmc.txt
  PMC_TIMEOUT=240 BENCH_ARGS="--no-bam $args" bash /root/repo/tools/pmc_total.sh 2>&1 | tee -a /root/repo/$O/pmc.txt
  grep -h "^{" /root/repo/gpurun_out/pmc_FETCH_SIZE.log | tail -1 > /root/repo/$O/pmc_bench_$name.json
done
cd /root/repo
bash tools/kernel_resources.sh > $O/kernel_resources.txt 2>&1
timeout 200 python tools/quick_gpu.py bench 32768 > $O/phases_headline.txt 2>&1
for c in bench60 bench4 bench5; do timeout 200 python tools/quick_gpu.py $c 8192 > $O/phases_$c.txt 2>&1; done
# the N-rank path end to end on the one-GPU box (both ranks on device 0, gather over gloo): a check of the communication thread, not a measurement
LANCET_BENCH_ONE_GPU=1 timeout 300 python bench.py --gpus 2 --steps 10 --settle 0 --windows 8192 --cpu-sample 0 --no-configs --no-bam > $O/bench_2rank_onegpu.json 2> $O/bench_2rank_onegpu.err; tail -n 1 $O/bench_2rank_onegpu.json | cut -c1-200
python tools/traffic_json.py $O $TAG --keep profiles/r5_traffic.json > $O/traffic.json
rm -rf $O/kt gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
