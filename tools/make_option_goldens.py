#!/usr/bin/env python3
"""Which windows the REFERENCE assembles, and with how many reads, under different command-line options: its own -v
"== Processing" lines on the committed flt_small BAMs (tests/golden/flt_small.*), one list per option set, written to
tests/golden/flt_small.options.txt (JSON).  Same reference binary and checks as tools/make_golden.py."""
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_golden as mg  # noqa: E402

OPTION_SETS = [
    [], ["--max-avg-cov", "22"], ["--active-region-off", "--max-avg-cov", "24"], ["--min-map-qual", "5"], ["--max-as-xs-diff", "2"],
    ["--max-as-xs-diff", "9"], ["--window-size", "500", "--padding", "100"], ["--window-size", "300"], ["--padding", "0"],
    ["--min-alt-count-tumor", "6"], ["--min-base-qual", "35"], ["--min-alt-count-tumor", "2", "--min-base-qual", "5"],
    ["--XA-tag-filter"], ["--primary-alignment-only", "--active-region-off"], ["--quality-range", "!"],
]
REGION = "chr22:900-3300"

if __name__ == "__main__":
    mg.check_reference_is_unmodified()
    out = {}
    with tempfile.TemporaryDirectory(prefix="lancet_opts_") as td:
        for f in ("tumor.bam", "normal.bam", "fa"):
            shutil.copy(os.path.join(mg.GOLDEN, f"flt_small.{f}"), os.path.join(td, f"x.{f}"))
        for rg in ("tumor", "normal"):
            mg.run([mg.BAMTOOLS, "index", "-in", os.path.join(td, f"x.{rg}.bam")])
        for opts in OPTION_SETS:
            r = subprocess.run([mg.REF_BIN, "--tumor", "x.tumor.bam", "--normal", "x.normal.bam", "--ref", "x.fa", "--reg", REGION,
                                "--num-threads", "1", "-v"] + opts, capture_output=True, text=True, cwd=td)
            if r.returncode != 0:
                raise SystemExit("reference failed on " + " ".join(opts))
            out[" ".join(opts)] = [f"{m.group(1)} {m.group(2)} {m.group(3)}" for m in
                                   re.finditer(r"== Processing \d+: (\S+) numsequences: (\d+) mapped: (\d+)", r.stderr)]
            print(" ".join(opts) or "(defaults)", len(out[" ".join(opts)]), "windows")
    json.dump({"region": REGION, "what": "hdr numsequences mapped per assembled window, from the reference's -v", "option_sets": out},
              open(os.path.join(mg.GOLDEN, "flt_small.options.txt"), "w"), indent=0)
