# VALU / SALU / LDS instructions of the LDS build kernel up to each phase boundary (stop-after-phase knob): where its instructions go.
cd /root/repo; mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
for st in 102 103 104 105 106 107 108 109 110 111 112 114 0; do
  d=/root/repo/gpurun_out/bphase_$st; rm -rf $d
  LANCET_STOP_PHASE=$st LANCET_AHEAD_DEPTH=0 timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $d -- python /root/repo/bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-configs --in-flight 1 --windows 8192 > $d.log 2>&1
  grep -F '"build_kernel(' $d/*/*counter_collection.csv | awk -F, -v st=$st '{n=NF; a[$(n-3)]+=$(n-2); c[$(n-3)]++; d+=($(n)-$(n-1))} END {printf "stop %s:", st; for (x in a) printf " %s %.4g", x, a[x]/c[x]; printf "\n"}'
done
