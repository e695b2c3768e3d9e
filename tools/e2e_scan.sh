# End-to-end region scan with the native command-line program (and the Python one for comparison) on pre-generated
# synthetic BAMs: build/scan500k (python tools/make_scan_bams.py build/scan500k 500000).  Runs on the GPU box.
cd /root/repo; D=build/scan500k; REG=chr22:1000-499000; mkdir -p gpurun_out
export LANCET_HOST_TIMING=1
for mode in "--active-region-off" ""; do
  for it in 1 2; do
    ./lancet_amd/bin/lancet_gpu --tumor $D/tumor.bam --normal $D/normal.bam --ref $D/ref.fa --reg $REG $mode --date-line "Sun Sep 27 05:27:00 2026" > gpurun_out/e2e_native$mode.vcf 2> gpurun_out/e2e_native$mode.log
    grep -h "lancet_gpu" gpurun_out/e2e_native$mode.log | tail -3
  done
done
time python -m lancet_amd.cli --tumor $D/tumor.bam --normal $D/normal.bam --ref $D/ref.fa --reg $REG --active-region-off > gpurun_out/e2e_python.vcf 2> gpurun_out/e2e_python.log; tail -2 gpurun_out/e2e_python.log
grep -v "^##fileDate\|^##cmdline" gpurun_out/e2e_native--active-region-off.vcf | md5sum; grep -v "^##fileDate\|^##cmdline" gpurun_out/e2e_python.vcf | md5sum
grep -vc "^#" gpurun_out/e2e_native--active-region-off.vcf gpurun_out/e2e_native.vcf
