#!/usr/bin/env python3
"""profiles/<tag>_traffic.json from one run of tools/profile_round4.sh: reads <dir>/pmc.txt (the FETCH_SIZE / WRITE_SIZE means per
kernel that tools/pmc_total.sh printed, one block per configuration) and <dir>/pmc_bench_<config>.json (the bench line of the same
pass: windows, coverage, algorithmic bytes), writes the records bench.py looks its `roofline.traffic` up in; the records of earlier
rounds (the file given as --keep) follow for comparison.

    python tools/traffic_json.py gpurun_out/r4_final r4_final --keep profiles/r4_traffic.json > profiles/r4_traffic.json.new"""
import json
import os
import re
import sys

NOTE = ("tools/profile_round4.sh -> tools/pmc_total.sh: two separate rocprofv3 passes (--kernel-trace --pmc FETCH_SIZE | WRITE_SIZE, no other "
        "trace domains) over `bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-configs --in-flight 1` with this configuration's arguments; KiB per "
        "launch as rocprofv3 reports them (printed as MiB in profiles/%s_pmc.txt), means over the launches; build_kernel + build_kernel_large + "
        "window_kernel of one launch.  Calibration in the same passes: prep_kernel (LANCET_PREP=device) moves a byte count known from the batch; "
        "FETCH_SIZE reports %.3f of its reads, WRITE_SIZE %.3f of its writes; traffic = FETCH_SIZE / %.3f + WRITE_SIZE.  Kernels are serialised "
        "under --pmc: the build service only waits and gives up, the window kernel builds the later graphs itself -- its share is an upper bound of "
        "a normal launch's.")


def main():
    d, tag = sys.argv[1], sys.argv[2]
    keep = sys.argv[sys.argv.index("--keep") + 1] if "--keep" in sys.argv else None
    blocks, cur = {}, None
    for line in open(os.path.join(d, "pmc.txt")):
        m = re.match(r"== PMC passes: (\S+) \((.*)\)", line)
        if m:
            cur = m.group(1); blocks[cur] = {"args": m.group(2), "mb": {}, "cal": None}; continue
        m = re.match(r"(\S+\() (FETCH_SIZE|WRITE_SIZE) mean ([0-9.]+) MB", line)
        if m and cur:
            blocks[cur]["mb"][f"{m.group(1)} {m.group(2)}"] = float(m.group(3)); continue
        m = re.search(r"are ([0-9.]+) MB to read and ([0-9.]+) MB to write", line)
        if m and cur:
            blocks[cur]["cal"] = (float(m.group(1)), float(m.group(2)))
    out = []
    for name, b in blocks.items():
        mb = b["mb"]
        fetch = sum(mb.get(f"{k}( FETCH_SIZE", 0.0) for k in ("build_kernel", "build_kernel_large", "window_kernel"))
        write = sum(mb.get(f"{k}( WRITE_SIZE", 0.0) for k in ("build_kernel", "build_kernel_large", "window_kernel"))
        fcal = wcal = 1.0
        if b["cal"] and "prep_kernel( FETCH_SIZE" in mb:
            # (pmc.txt prints MiB -- KiB / 1024 --, the calibration line decimal MB)
            fcal = round(mb["prep_kernel( FETCH_SIZE"] * 1.048576 / b["cal"][0], 4); wcal = round(mb["prep_kernel( WRITE_SIZE"] * 1.048576 / b["cal"][1], 4)
        with open(os.path.join(d, f"pmc_bench_{name}.json")) as fh:
            line = json.loads(fh.read())
        W = line["config"]["windows_per_gpu"]; cov = line["config"]["coverage"]
        alg = line["roofline"]["algorithmic_bytes_per_launch"]
        traffic = (fetch / fcal + write) * 1024 * 1024
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
        from lancet_amd import workload
        rec = {"round": tag, "config": name, "windows": W, "coverage": cov[0], "kernel_fingerprint": workload.kernel_fingerprint(),
               "kernel_ms_of_the_pass": line["roofline"].get("per_kernel_ms")}
        if cov[1] != cov[0]:
            rec["coverage_normal"] = cov[1]
        m = re.search(r"--str-fraction (\S+)", b["args"])
        if m:
            rec["str_fraction"] = float(m.group(1))
        if "--linked" in b["args"]:
            rec["linked"] = True
        rec.update({"FETCH_SIZE_KB": fetch * 1024, "WRITE_SIZE_KB": write * 1024, "per_kernel_MB": mb, "fetch_calibration": fcal, "write_calibration": wcal,
                    "algorithmic_bytes_per_launch": alg, "traffic_bytes": int(traffic), "traffic_over_algorithmic": round(traffic / alg, 3),
                    "bytes_per_window_raw": int((fetch + write) * 1024 * 1024 / W), "bytes_per_window_calibrated": int(traffic / W),
                    "note": NOTE % (tag, fcal, wcal, fcal)})
        out.append(rec)
    old = []
    if keep and os.path.exists(keep):
        old = [r for r in json.load(open(keep))["measurements"] if r.get("round") != tag]
    print(json.dumps({"how": "tools/profile_round5.sh + tools/traffic_json.py on MI355X (see each record's note); the records of earlier rounds follow for comparison",
                      "measurements": out + old}, indent=1))


if __name__ == "__main__":
    main()
