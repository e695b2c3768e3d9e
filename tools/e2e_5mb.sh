# BASELINE.md config 2 through the native command-line program: the 5 Mb synthetic contig (49 981 windows of 600 bp every 100 bp), 30x/30x and
# 60x/60x tumor / normal pairs made by tools/make_scan_bams.py (build/scan5m, build/scan5m60: not part of the repository's history, shipped to
# the GPU box with the snapshot).  One engine (a batch's host work and its kernels in turn) and two engines on the GPU (--devices 0,0: the next
# batch is decoded / selected / packed / uploaded under the kernels of the one before), batches of 8192 and of 32768 windows; the Python twin
# of the host side for the VCF's md5 on the 30x pair.      bash tools/e2e_5mb.sh > gpurun_out/r5_e2e_5mb.txt
cd /root/repo; mkdir -p gpurun_out
export LANCET_HOST_TIMING=1 LANCET_UPLOAD_TIMING=1
REG=chr22:1000-4999000
nproc; grep -h "cpu.max" /dev/null; cat /sys/fs/cgroup/cpu.max 2>/dev/null
for D in build/scan5m build/scan5m60; do
  # (the pairs are 134 + 260 MB: made on the box -- a minute or two on its CPUs -- rather than shipped with the snapshot)
  [ -f $D/tumor.bam ] || { cov=30; [ $D = build/scan5m60 ] && cov=60; python tools/make_scan_bams.py $D 5000000 $cov $cov 14 | tail -n 1; }
  [ -f $D/tumor.bam ] || { echo "== $D: not there"; continue; }
  ls -la $D | awk '{print $5, $9}' | tail -7
  for cfg in "--devices 0 --batch-windows 32768" "--devices 0 --batch-windows 32768" "--devices 0,0 --batch-windows 32768" "--devices 0,0 --batch-windows 8192"; do
    sleep 6            # (a process that allocates tens of GB right after another one released as much waits seconds in hipMalloc: DESIGN_HISTORY.md 7a)
    echo "== $D: lancet_gpu $cfg --active-region-off"
    ( time ./lancet_amd/bin/lancet_gpu --tumor $D/tumor.bam --normal $D/normal.bam --ref $D/ref.fa --reg $REG $cfg --active-region-off --date-line "Sun Sep 27 05:27:00 2026" > gpurun_out/e2e5_$(basename $D).vcf 2> gpurun_out/e2e5.log ) 2>&1 | grep -E "real|user" | tr '\n' ' '; echo
    grep -h "lancet_gpu\]\|alignments kept\|windows: select\|lancet upload" gpurun_out/e2e5.log | tail -n 24
    grep -v "^##fileDate\|^##cmdline" gpurun_out/e2e5_$(basename $D).vcf | md5sum; grep -vc "^#" gpurun_out/e2e5_$(basename $D).vcf
  done
  sleep 6
  echo "== $D: default mode (active regions on), --devices 0 --batch-windows 32768"
  ./lancet_amd/bin/lancet_gpu --tumor $D/tumor.bam --normal $D/normal.bam --ref $D/ref.fa --reg $REG --devices 0 --batch-windows 32768 --date-line "Sun Sep 27 05:27:00 2026" > gpurun_out/e2e5_ar.vcf 2> gpurun_out/e2e5.log
  grep -h "lancet_gpu\]" gpurun_out/e2e5.log | tail -2; grep -vc "^#" gpurun_out/e2e5_ar.vcf
done
D=build/scan5m
if [ -f $D/tumor.bam ]; then
  echo "== the Python twin of the host side (same engine) against the native program on the first megabase of the 30x pair: VCF md5 of both"
  R1=chr22:1000-1000000
  ./lancet_amd/bin/lancet_gpu --tumor $D/tumor.bam --normal $D/normal.bam --ref $D/ref.fa --reg $R1 --active-region-off --date-line "Sun Sep 27 05:27:00 2026" > gpurun_out/e2e5_native1m.vcf 2> gpurun_out/e2e5.log
  grep -h "lancet_gpu\]" gpurun_out/e2e5.log | tail -2
  ( time timeout 900 python -m lancet_amd.cli --tumor $D/tumor.bam --normal $D/normal.bam --ref $D/ref.fa --reg $R1 --active-region-off > gpurun_out/e2e5_python1m.vcf 2> gpurun_out/e2e5_python.log ) 2>&1 | grep real
  tail -n 1 gpurun_out/e2e5_python.log
  grep -v "^##fileDate\|^##cmdline" gpurun_out/e2e5_native1m.vcf | md5sum; grep -v "^##fileDate\|^##cmdline" gpurun_out/e2e5_python1m.vcf | md5sum
fi
rm -f gpurun_out/e2e5_*.vcf
