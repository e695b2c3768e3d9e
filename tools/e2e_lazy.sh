# lancet_gpu on the indexed 500 kb BAM pair: eager load against lazy (per-batch) load in chunks of 700 windows; same VCF
cd /root/repo; D=build/scan500k; REG=chr22:1000-499000; mkdir -p gpurun_out
export LANCET_HOST_TIMING=1
for lazy in 0 1; do
  LANCET_HOST_LAZY=$lazy ./lancet_amd/bin/lancet_gpu --tumor $D/tumor.bam --normal $D/normal.bam --ref $D/ref.fa --reg $REG --active-region-off --batch-windows 700 --date-line "Sun Sep 27 05:27:00 2026" > gpurun_out/e2e_lazy$lazy.vcf 2> gpurun_out/e2e_lazy$lazy.log
  echo "lazy=$lazy: $(grep -c 'alignments kept' gpurun_out/e2e_lazy$lazy.log) loads; $(grep -h '\[lancet_gpu\] wall' gpurun_out/e2e_lazy$lazy.log)"
  grep -v "^##fileDate\|^##cmdline" gpurun_out/e2e_lazy$lazy.vcf | md5sum
done
