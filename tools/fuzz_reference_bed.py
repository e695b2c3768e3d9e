#!/usr/bin/env python3
"""Differential run of the region / BED handling against the REFERENCE ITSELF (this container only; binary of
tools/make_golden.py): random BED files (unsorted, overlapping, touching, tiny and contig-long intervals, comment and blank
lines, intervals near the contig ends) with or without --reg on the two-contig inputs of the `bed2` golden, random
--window-size / --padding / read filters.  The windows the reference assembles, in its order, with its read counts (the
"== Processing" lines of -v) against lancet_host_tile_regions + lancet_host_batch (include/lancet_host.h).
Reference code under test: loadBed / loadRefs / the window table (src/Lancet.cc:189-362, 852-857, src/Microassembler.cc:779).

    python tools/fuzz_reference_bed.py [first_seed] [n]"""
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_golden as mg  # noqa: E402
from lancet_amd import host  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
CONTIGS = {"chr21": 3600, "chr22": 4000}


def random_bed(rng):
    lines = []
    if rng.random() < 0.3:
        lines.append("# a comment line")
    for _ in range(int(rng.integers(1, 7))):
        c = str(rng.choice(list(CONTIGS)))
        L = CONTIGS[c]
        kind = rng.random()
        if kind < 0.15:
            a = int(rng.integers(0, 200)); b = a + int(rng.integers(50, 900))                  # at the contig start
        elif kind < 0.3:
            b = L - int(rng.integers(0, 200)); a = max(0, b - int(rng.integers(50, 900)))       # at the contig end
        elif kind < 0.4:
            a = int(rng.integers(300, L - 300)); b = a + int(rng.integers(1, 30))                # tiny
        elif kind < 0.45:
            a, b = 0, L                                                                        # the whole contig
        else:
            a = int(rng.integers(200, L - 800)); b = a + int(rng.integers(100, 1500))
        sa, sb = str(a), str(min(b, L))
        v = rng.random()                       # what std::stoi lets through (reference src/Lancet.cc:343-344)
        if v < 0.06: sa = " " + sa
        elif v < 0.12: sa = "+" + sa
        elif v < 0.18: sb = sb + "abc"
        elif v < 0.24 and a < 400: sa = str(a - 600)          # negative start: clamped to 1 after the padding
        line = f"{c}\t{sa}\t{sb}" + ("\tname%d\t0\t+" % len(lines) if rng.random() < 0.2 else "")
        lines.append(line + ("\r" if rng.random() < 0.05 and line.count("\t") > 2 else ""))
    return "\n".join(lines) + "\n"


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    mg.check_reference_is_unmodified()
    bad = []
    with tempfile.TemporaryDirectory(prefix="lancet_fzbed_") as td:
        # the reference reads <bam>.bai next to the BAM: work on copies
        for f in ("bed2.tumor.bam", "bed2.tumor.bam.bai", "bed2.normal.bam", "bed2.normal.bam.bai", "bed2.fa", "bed2.fa.fai"):
            subprocess.run(["cp", os.path.join(G, f), td], check=True)
        T, N, FA = (os.path.join(td, f) for f in ("bed2.tumor.bam", "bed2.normal.bam", "bed2.fa"))
        for seed in range(first, first + n):
            rng = np.random.default_rng(70000 + seed)
            opts, kw = [], {}
            def add(flag, field, val):
                opts.extend([flag, str(val)]); kw[field] = val
            if rng.random() < 0.4: add("--window-size", "window_size", int(rng.choice([300, 450, 600])))
            if rng.random() < 0.5: add("--padding", "padding", int(rng.choice([0, 100, 250, 400])))
            if rng.random() < 0.3: add("--min-map-qual", "min_map_qual", int(rng.integers(0, 40)))
            if rng.random() < 0.6: opts.append("--active-region-off"); kw["active_region"] = 0
            use_bed = rng.random() < 0.85
            regions = []
            if not use_bed or rng.random() < 0.4:
                c = str(rng.choice(list(CONTIGS))); L = CONTIGS[c]
                if rng.random() < 0.15: regions = [c]
                else:
                    a = int(rng.integers(1, L - 700)); regions = [f"{c}:{a}-{min(L, a + int(rng.integers(100, 1800)))}"]
            bed = os.path.join(td, f"r{seed}.bed")
            cmd = [mg.REF_BIN, "--tumor", T, "--normal", N, "--ref", FA, "--num-threads", "1", "-v"] + opts
            bed_text = ""
            if use_bed:
                bed_text = random_bed(rng); open(bed, "w").write(bed_text); cmd += ["--bed", bed]
            if regions: cmd += ["--reg", regions[0]]
            r = subprocess.run(cmd, capture_output=True, text=True, cwd=td)
            if r.returncode != 0:
                print(f"bed{seed}: reference failed rc={r.returncode} ({' '.join(opts)} bed={bed_text!r} reg={regions})"); continue
            want = [f"{m.group(1)} {m.group(2)} {m.group(3)} {m.group(4)}" for m in re.finditer(r"== Processing (\d+): (\S+) numsequences: (\d+) mapped: (\d+)", r.stderr)]
            o = host.default_opts(**kw)
            H = host.NativeHost(T, N, FA)
            try:
                hdrs = H.tile_regions(regions, o, bed=bed if use_bed else None)
                b, idx = H.batch(0, len(hdrs), o)
                nr = np.diff(b.read_begin.astype(np.int64))
                got = [f"{w + 1} {b.hdr[w]} {int(nr[w])} {int(b.mapped[b.read_begin[w]:b.read_begin[w + 1]].sum())}" for w in range(b.n_windows)]
                got = [g for g in got if not g.endswith(" 0")]
            except Exception as ex:          # noqa: BLE001
                got = [f"native failed: {ex}"]
            H.close()
            ok = got == want
            print(f"bed{seed}: {' '.join(opts)} bed={bed_text!r} reg={regions}: reference {len(want)} windows, native {len(got)}: {'ok' if ok else 'MISMATCH'}")
            if not ok:
                sw, sg = set(want), set(got)
                print("    only reference:", sorted(sw - sg)[:4], "only native:", sorted(sg - sw)[:4])
                bad.append(seed)
            sys.stdout.flush()
    print("mismatches:", bad)


if __name__ == "__main__":
    main()
