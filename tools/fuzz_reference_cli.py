#!/usr/bin/env python3
"""Differential run of the host side against the REFERENCE ITSELF (this container only): random reads with the tag / flag /
CIGAR variety of tests/read_variety.py, random read-filter / tiling / active-region options; the windows the reference
assembles and its read counts per window (its -v "== Processing" lines) against the native host side (include/lancet_host.h).

    python tools/fuzz_reference_cli.py [first_seed] [n]"""
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_golden as mg  # noqa: E402
import read_variety  # noqa: E402
from lancet_amd import host, synth  # noqa: E402


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    mg.check_reference_is_unmodified()
    bad = []
    for seed in range(first, first + n):
        rng = np.random.default_rng(5000 + seed)
        linked = bool(rng.random() < 0.3)
        data = synth.make_tumor_normal(ref_len=int(rng.integers(3000, 4600)), cov_t=float(rng.choice([20, 34, 60])), cov_n=float(rng.choice([16, 28, 45])),
                                       ref_seed=100 + seed, tumor_seed=300 + seed, normal_seed=500 + seed, read_len=int(rng.choice([76, 100, 150])),
                                       insert_mean=float(rng.choice([200, 300, 400])), somatic_every=int(rng.choice([400, 900])), germline_every=int(rng.choice([300, 700])),
                                       error_rate=float(rng.choice([0.0, 0.004, 0.012])), str_fraction=float(rng.choice([0.0, 0.1])))
        reads = {rg: read_variety.decorate(synth.pairs_to_sorted_reads(data[rg]), rng, linked) for rg in ("tumor", "normal")}
        if rng.random() < 0.25:      # a stretch of unmapped-flagged reads: windows without a mapped read leave their reads in the graph (H6)
            lo = int(rng.integers(900, 1800)); hi = lo + int(rng.integers(650, 1000))
            for rg in reads:
                reads[rg] = [synth.SamRead(r.qname, r.flag | (0x4 if lo <= r.pos - 1 < hi else 0), r.rname, r.pos, r.mapq, r.cigar, r.seq, r.qual, r.tags) for r in reads[rg]]
        # --rg-file (reference src/Microassembler.cc:29-48, 296-302, 611-616): every read gets RG:Z:rgA / rgB / rgC or none, the file names a
        # random subset (also the empty one and one no read carries)
        rg_listed = None
        if rng.random() < 0.25:
            for rg in reads:
                out = []
                for r in reads[rg]:
                    tags = dict(r.tags); u = rng.random()
                    if u < 0.4: tags["RG"] = "rgA"
                    elif u < 0.7: tags["RG"] = "rgB"
                    elif u < 0.9: tags["RG"] = "rgC"
                    else: tags.pop("RG", None)
                    out.append(synth.SamRead(r.qname, r.flag, r.rname, r.pos, r.mapq, r.cigar, r.seq, r.qual, tags))
                reads[rg] = out
            rg_listed = [g for g in ("rgA", "rgB", "rgC", "rgZ") if rng.random() < 0.5]
        opts, kw = [], {}
        def add(flag, field, val, conv=lambda v: v):
            opts.extend([flag, str(val)]); kw[field] = conv(val)
        if linked:
            opts.append("--linked-reads"); kw["linked"] = 1
        if rng.random() < 0.5: add("--min-map-qual", "min_map_qual", int(rng.integers(0, 40)))
        if rng.random() < 0.3: add("--max-avg-cov", "max_avg_cov", int(rng.integers(15, 80)))
        if rng.random() < 0.3: add("--window-size", "window_size", int(rng.choice([300, 450, 600])))
        if rng.random() < 0.3: add("--padding", "padding", int(rng.choice([0, 100, 250, 400])))
        if rng.random() < 0.4: add("--min-alt-count-tumor", "min_evidence", int(rng.integers(1, 8)))
        if rng.random() < 0.4: add("--min-base-qual", "min_qual_call", int(rng.integers(3, 40)), lambda v: v + 33)
        if rng.random() < 0.3: add("--max-k", "max_k", int(rng.choice([35, 61, 101])))
        if rng.random() < 0.4: opts.append("--XA-tag-filter"); kw["xa_filter"] = 1
        if rng.random() < 0.4: opts.append("--primary-alignment-only"); kw["primary_alignment_only"] = 1
        if rng.random() < 0.3: opts.append("--active-region-off"); kw["active_region"] = 0
        if rng.random() < 0.2: opts.extend(["--max-as-xs-diff", str(int(rng.integers(0, 12)))])        # (inert in the reference's main())
        # VariantDB / VCF filter thresholds (reference src/Variant.hh:42-56; they only shape the FILTER column and the header)
        flt = {}
        def addf(flag, field, val):
            opts.extend([flag, str(val)]); flt[field] = val
        if "min_evidence" in kw: flt["min_alt_cnt_tumor"] = kw["min_evidence"]
        if rng.random() < 0.3: addf("--max-alt-count-normal", "max_alt_cnt_normal", int(rng.integers(0, 4)))
        if rng.random() < 0.3: addf("--min-vaf-tumor", "min_vaf_tumor", round(float(rng.choice([0.01, 0.1, 0.25])), 2))
        if rng.random() < 0.3: addf("--max-vaf-normal", "max_vaf_normal", round(float(rng.choice([0.0, 0.05, 0.2])), 2))
        if rng.random() < 0.3: addf("--min-coverage-tumor", "min_cov_tumor", int(rng.integers(2, 30)))
        if rng.random() < 0.3: addf("--min-coverage-normal", "min_cov_normal", int(rng.integers(2, 30)))
        if rng.random() < 0.2: addf("--max-coverage-tumor", "max_cov_tumor", int(rng.integers(20, 200)))
        if rng.random() < 0.2: addf("--max-coverage-normal", "max_cov_normal", int(rng.integers(20, 200)))
        if rng.random() < 0.3: addf("--min-phred-fisher", "min_phred_fisher", round(float(rng.choice([1.0, 5.0, 12.5])), 1))
        if rng.random() < 0.3: addf("--min-phred-fisher-str", "min_phred_fisher_str", round(float(rng.choice([10.0, 25.0, 40.0])), 1))
        if rng.random() < 0.3: addf("--min-strand-bias", "min_strand_bias", int(rng.integers(0, 4)))
        L = len(data["ref"]); a = int(rng.integers(500, 1000)); region = f"{data['rname']}:{a}-{min(L - 300, a + int(rng.integers(800, 2500)))}"
        if rng.random() < 0.1: region = data["rname"]
        with tempfile.TemporaryDirectory(prefix=f"lancet_fzcli_{seed}_") as td:
            fa = os.path.join(td, "ref.fa"); synth.write_fasta(fa, data["rname"], data["ref"])
            bams = {}
            for sample, rg in (("TUMOR", "tumor"), ("NORMAL", "normal")):
                sam, bam = os.path.join(td, f"{rg}.sam"), os.path.join(td, f"{rg}.bam")
                with open(sam, "w") as f:
                    extra_rg = [f"@RG\tID:{x}\tSM:{sample}\tPL:ILLUMINA" for x in ("rgA", "rgB", "rgC")] if rg_listed is not None else []
                    f.write("\n".join(["@HD\tVN:1.6\tSO:coordinate", f"@SQ\tSN:{data['rname']}\tLN:{L}", f"@RG\tID:{rg}\tSM:{sample}\tPL:ILLUMINA"] + extra_rg
                                      + [read_variety.sam_line(r) for r in reads[rg]]) + "\n")
                mg.run([mg.TEST_VIEW, "-b", "-p", bam, sam], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                mg.run([mg.BAMTOOLS, "index", "-in", bam]); bams[rg] = bam
            rgfile = None
            if rg_listed is not None:
                rgfile = os.path.join(td, "rg.txt")
                open(rgfile, "w").write("".join(g + "\n" for g in rg_listed))
                opts = opts + ["--rg-file", rgfile]
            r = subprocess.run([mg.REF_BIN, "--tumor", bams["tumor"], "--normal", bams["normal"], "--ref", fa, "--reg", region, "--num-threads", "1", "-v"] + opts,
                               capture_output=True, text=True, cwd=td)
            if r.returncode != 0:
                print(f"cli{seed}: reference failed ({opts})"); continue
            want = [f"{m.group(1)} {m.group(2)} {m.group(3)} {m.group(4)}" for m in re.finditer(r"== Processing (\d+): (\S+) numsequences: (\d+) mapped: (\d+)", r.stderr)]
            o = host.default_opts(**kw)
            H = host.NativeHost(bams["tumor"], bams["normal"], fa)
            if rgfile: H.set_rg_file(rgfile)
            hdrs = H.tile(region, o)
            b, idx = H.batch(0, len(hdrs), o)
            nr = np.diff(b.read_begin.astype(np.int64))
            got = [f"{w + 1} {b.hdr[w]} {int(nr[w])} {int(b.mapped[b.read_begin[w]:b.read_begin[w + 1]].sum())}" for w in range(b.n_windows)]
            got = [g for g in got if not g.endswith(" 0")]          # (a window without a mapped read prints nothing, but uses up a number)
            ok = got == want
            vcf_ok = True
            if ok and b.n_windows and seed % 2 == 0:          # every other case: the whole host path below the seam as well
                import ctypes as C
                from oracle import oracle
                from lancet_amd import abi, engine
                p = abi.default_params(lr_mode=int(linked), max_k=kw.get("max_k", 101), min_qual_call=kw.get("min_qual_call", 17 + 33))
                ov, _, _ = oracle.run(b, p)
                f = abi.LancetFilters(); engine.lib().lancet_filters_default(C.byref(f))
                for field, val in flt.items(): setattr(f, field, val)
                db = engine.VariantDB(f)
                db.add_records(ov, [data["rname"]], bx_names=b.bx_names if linked else None)
                body = lambda t: "".join(l + "\n" for l in t.splitlines() if not l.startswith(("##fileDate", "##cmdline", "##reference")))
                vcf_ok = body(db.vcf(sample_normal="NORMAL", sample_tumor="TUMOR")) == body(r.stdout)
            H.close()
            ok = ok and vcf_ok
            print(f"cli{seed}: {region} {' '.join(opts)}: reference {len(want)} windows, native {len(got)}, vcf {'same' if vcf_ok else 'DIFFERENT'}: {'ok' if ok else 'MISMATCH'}")
            if not ok:
                sw, sg = set(want), set(got)
                print("    only reference:", sorted(sw - sg)[:3], "only native:", sorted(sg - sw)[:3])
                bad.append(seed)
            sys.stdout.flush()
    print("mismatches:", bad)


if __name__ == "__main__":
    main()
