# The native command-line program on the pre-generated 500 kb BAM pair (build/scan500k), twice per mode, with the host side's stage times
cd /root/repo; D=build/scan500k; REG=chr22:1000-499000; mkdir -p gpurun_out
export LANCET_HOST_TIMING=1
for mode in "--active-region-off" ""; do
  for it in 1 2; do
    ./lancet_amd/bin/lancet_gpu --tumor $D/tumor.bam --normal $D/normal.bam --ref $D/ref.fa --reg $REG $mode --date-line "Sun Sep 27 05:27:00 2026" > gpurun_out/e2e_native$mode.vcf 2> gpurun_out/e2e_native$mode.log
    grep -h "lancet_gpu\|tumor.bam\|windows: select" gpurun_out/e2e_native$mode.log | tail -4
  done
done
# the same run four times with 5 s between the processes: the seconds of hipMalloc some of the runs above pay are the driver wiping the VRAM
# the process BEFORE released (DESIGN_HISTORY.md 7a, tools/malloc_probe.hip) -- a process that starts on an idle device pays none
echo "== --active-region-off, 5 s after the previous process, four times"
for it in 1 2 3 4; do
  sleep 5
  ./lancet_amd/bin/lancet_gpu --tumor $D/tumor.bam --normal $D/normal.bam --ref $D/ref.fa --reg $REG --active-region-off --date-line "Sun Sep 27 05:27:00 2026" > gpurun_out/e2e_native--active-region-off.vcf 2> gpurun_out/e2e_native_spaced.log
  grep -h "lancet_gpu\] wall" gpurun_out/e2e_native_spaced.log | tail -1
done
grep -v "^##fileDate\|^##cmdline" gpurun_out/e2e_native--active-region-off.vcf | md5sum
grep -vc "^#" gpurun_out/e2e_native--active-region-off.vcf gpurun_out/e2e_native.vcf
