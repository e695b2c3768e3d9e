"""Tuning aid: the kernels of a rocprofv3 --kernel-trace run in time order (start, end, duration in ms from the first start; queue; name).
    python tools/trace_timeline.py <dir with *_kernel_trace.csv> [first_ms] [last_ms]"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r.get("Stream_Id", "?"), r["Kernel_Name"].split("(")[0][:28]))
rows.sort()
t0 = rows[0][0]
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0; hi = float(sys.argv[3]) if len(sys.argv) > 3 else 1e18
for s, e, q, st, n in rows:
    a = (s - t0) / 1e6
    if lo <= a <= hi: print(f"{a:10.3f} {(e - t0) / 1e6:10.3f} {(e - s) / 1e6:8.3f}  q{q} s{st}  {n}")
