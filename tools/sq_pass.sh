# SQ counter passes for build_kernel (and window_kernel) on the bench workload: what the waves do with their cycles, as a measurement.
#   tools/sq_pass.sh [windows]      -> gpurun_out/sq/summary.txt   (means per launch; cycles counters are quad-cycles, guide: MI355X_MICROARCH.md)
# Per phase: LANCET_STOP_PHASE=<marker> abandons every window after that phase of the build kernel (build_lds_impl.h debug_stop), so the
# difference of two runs is the phase between the markers.  rocprofv3 --pmc serialises kernels: the service only waits (LANCET_SVC_HELP=0).
W=${1:-8192}
cd /root/repo; O=/root/repo/gpurun_out/sq; rm -rf $O; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u > $O/sq_counters_available.txt
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
P3="SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_WAVES SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM"
run() {   # tag, stop, counters
  d=$O/$1; rm -rf $d
  LANCET_STOP_PHASE=$2 LANCET_SVC_HELP=0 timeout ${PMC_TIMEOUT:-150} rocprofv3 --kernel-trace --pmc $3 --output-format csv -d $d -- python /root/repo/bench.py --steps 2 --warmup 0 --cpu-sample 0 --no-configs --in-flight 1 --windows $W > $d.log 2>&1 || echo "pass $1: rc $?" >> $O/summary.txt
  python3 - $d $1 >> $O/summary.txt <<'PY'
import csv, glob, sys, collections
d, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for f in glob.glob(d + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if not any(x in k for x in ("build_kernel", "window_kernel")): continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if "Start_Timestamp" in r: dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k in sorted(acc):
    ms = dur[k]; n = max(len(v) for v in acc[k].values())
    print("%s %s launches %d %s" % (tag, k, n, " ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(acc[k].items()))))
PY
  rm -rf $d
}
run full_p1 0 "$P1"
run full_p2 0 "$P2"
run full_p3 0 "$P3"
for st in 102 103 104 105 107 108 109 110 111 112 114; do run stop${st}_p1 $st "$P1"; done
cat $O/summary.txt
