# BASELINE.md config 3 (whole-genome synthetic 30x/30x, windows sharded across GPUs, records gathered into the VariantDB on rank 0) as a
# ONE-GPU proxy: 24 contigs x 1 Mb (chr1 .. chr22, chrX, chrY: 239 544 windows of 600 bp every 100 bp), one tumor / normal BAM pair over all of
# them, `lancet_gpu --bed` over the whole table.  The same VCF must come out of one engine, of two engines on the GPU, of batches of 8192 and
# 32768 windows, of the N-process route with one rank (RCCL) and with two ranks on the one GPU (payloads over the test transport: RCCL
# refuses two ranks per device) -- and of the oracle (tools/e2e_parity.py: every record + the VCF).   bash tools/config3_proxy.sh > gpurun_out/r6_config3_proxy.txt
cd /root/repo; mkdir -p gpurun_out
export LANCET_HOST_TIMING=1
D=build/scan24x1m
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
[ -f $D/tumor.bam ] || python tools/make_scan_bams.py $D 1000000 30 30 14 24 | tail -n 3
ls -la $D | awk '{print $5, $9}' | tail -8
run() {   # label, env, args...
  local label="$1"; shift; local envs="$1"; shift
  sleep 5
  echo "== lancet_gpu --bed $D/regions.bed $* --active-region-off   [$label]"
  ( time env $envs ./lancet_amd/bin/lancet_gpu --tumor $D/tumor.bam --normal $D/normal.bam --ref $D/ref.fa --bed $D/regions.bed "$@" --active-region-off --date-line "Sun Sep 27 05:27:00 2026" > gpurun_out/c3.vcf 2> gpurun_out/c3.log ) 2>&1 | grep -E "real|user" | tr '\n' ' '; echo
  grep -h "lancet_gpu\]" gpurun_out/c3.log | tail -n 4
  grep -v "^##fileDate\|^##cmdline" gpurun_out/c3.vcf | md5sum; grep -vc "^#" gpurun_out/c3.vcf
}
run "one engine" "X=1" --devices 0 --batch-windows 32768
run "one engine, again (warm file cache)" "X=1" --devices 0 --batch-windows 32768
run "two engines on the GPU" "X=1" --devices 0,0 --batch-windows 32768
run "two engines, batches of 8192" "X=1" --devices 0,0 --batch-windows 8192
run "N-process route, 1 rank, RCCL" "X=1" --ranks 1 --batch-windows 32768
run "N-process route, 2 ranks on the one GPU, test transport" "LANCET_COMM_TEST_FILES=1" --ranks 2 --devices 0,0 --batch-windows 16384
run "N-process route, 4 ranks on the one GPU, test transport" "LANCET_COMM_TEST_FILES=1" --ranks 4 --devices 0,0,0,0 --batch-windows 8192
echo "== every record and the VCF against the oracle (tools/e2e_parity.py, batches of 32768)"
python tools/e2e_parity.py $D $D/regions.bed --batch-windows 32768 --active-region-off 2> gpurun_out/c3_parity.err | cut -c 1-1600
tail -n 3 gpurun_out/c3_parity.err
rm -f gpurun_out/c3.vcf
