# HBM-side traffic of the kernels for the default bench workload: two passes (FETCH_SIZE, WRITE_SIZE), MB per launch.
# rocprofv3 --pmc serialises kernels, so for these passes
#   * the build service only waits (LANCET_SVC_HELP=0: it would otherwise build the whole batch's first graphs itself, on 24 workgroups,
#     before the build kernel is allowed to start), gives up after 300 ms without progress, and the window slots build the ~400 later
#     graphs themselves -- the window kernel's traffic here is an upper bound of a normal launch's;
#   * trim + pack runs on the device (LANCET_PREP=device): prep_kernel reads and writes a byte count known from the batch, in this
#     code's own access widths (bytes and 4-byte words, coalesced) -- the calibration the guide asks for before trusting an absolute.
cd /root/repo; mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  d=/root/repo/gpurun_out/pmc_$c; rm -rf $d
  LANCET_SVC_HELP=0 LANCET_PREP=device timeout ${PMC_TIMEOUT:-150} rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- python /root/repo/bench.py --steps 1 --warmup 0 --settle 0 --cpu-sample 0 --no-configs --in-flight 1 $BENCH_ARGS > $d.log 2>&1 || echo "pass $c: rc $?"
  for k in "build_kernel(" "build_kernel_large(" "window_kernel(" "svc_kernel(" "prep_kernel("; do
    grep -F "$k" $d/*/*counter_collection.csv | awk -F, -v k=$k -v c=$c '{n=NF; v[NR]=$(n-2); t[NR]=($(n)-$(n-1))/1e6} END {if (NR) { s=0; for (i=1;i<=NR;i++) s+=v[i]; printf "%s %s mean %.1f MB over %d launches (last %.1f MB, %.1f ms)\n", k, c, s/NR/1024, NR, v[NR]/1024, t[NR]} }'
  done
done
python3 - <<'PY'
import json
try:
    d = [json.loads(l) for l in open("/root/repo/gpurun_out/pmc_FETCH_SIZE.log") if l.startswith("{")][-1]
    R = d["config"]["reads_per_gpu"]; W = d["config"]["windows_per_gpu"]
    rd = (2 * 150 + 4 + 4 + 8) * R / 1e6          # bases + qualities (150 each), offsets, four flag bytes, two word offsets per read
    wr = (10 + 5 + 1) * 4 * R / 1e6               # 10 words of bases, 5 of quality mask, the info word
    print("prep_kernel calibration: the batch's %d reads of 150 bases are %.1f MB to read and %.1f MB to write per launch (compare the prep_kernel lines above)" % (R, rd, wr))
except Exception as ex:
    print("prep_kernel calibration: no bench line (%s)" % ex)
PY
