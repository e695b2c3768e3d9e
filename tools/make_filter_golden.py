#!/usr/bin/env python3
"""Golden case `flt_small`: the REFERENCE ITSELF (same binary and recipe as tools/make_golden.py) on reads that exercise
every read filter of extractReads / isActiveRegion -- XT:A:R, XA, AS/XS ties, duplicates, secondary alignments, low MAPQ,
soft clips, unmapped flags -- with `--XA-tag-filter --primary-alignment-only --min-map-qual 20`, active regions on.

Written to tests/golden/: flt_small.{tumor,normal}.bam + .fa (inputs, made by htslib's test_view), .vcf (expected output),
.trace.txt (digest of the reference's -v: which windows were assembled, with how many reads, and every stage result),
.case.txt (the command line, JSON).  Nothing of the reference travels; only these data files do."""
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import read_variety  # noqa: E402
from lancet_amd import synth  # noqa: E402
import make_golden as mg  # noqa: E402

CASES = {
    "flt_small": (dict(ref_len=4200, cov_t=34, cov_n=28, ref_seed=71, tumor_seed=171, normal_seed=271, somatic_every=800, germline_every=600),
                  False, ["--XA-tag-filter", "--primary-alignment-only", "--min-map-qual", "20"], "chr22:900-3300"),
    # --linked-reads --primary-alignment-only on BX/HP-tagged reads (some without barcode / haplotype)
    "lrflt_small": (dict(ref_len=3800, cov_t=32, cov_n=28, ref_seed=73, tumor_seed=173, normal_seed=273, somatic_every=700, germline_every=500,
                         insert_mean=280.0, insert_sd=40.0), True, ["--linked-reads", "--primary-alignment-only"], "chr22:800-2900"),
    # every read that starts in [1500, 2250) carries the unmapped flag: the window chr22:1550-2150 holds no mapped read, the
    # reference's processGraph returns before g.clear() (src/Microassembler.cc:83) and its reads are still in the graph when
    # the next window is loaded
    # --rg-file (reference src/Microassembler.cc:29-48, 296-302, 611-616): reads carry RG:Z:rgA / rgB / rgC or no RG at all, the file
    # names rgA and rgC; active regions on (isActiveRegion reads the tag without clearing the previous read's value)
    "rg_small": (dict(ref_len=4000, cov_t=44, cov_n=36, ref_seed=77, tumor_seed=177, normal_seed=277, somatic_every=600, germline_every=500),
                 False, ["--rg-file", "RGFILE"], "chr22:900-3100"),
    "leak_small": (dict(ref_len=4000, cov_t=30, cov_n=26, ref_seed=75, tumor_seed=175, normal_seed=275, somatic_every=500, germline_every=400),
                   False, ["--active-region-off"], "chr22:1000-3000"),
}

if __name__ == "__main__":
  mg.check_reference_is_unmodified()
  for NAME in (sys.argv[1:] or list(CASES)):
    kwargs, linked, FLAGS, REGION = CASES[NAME]
    data = synth.make_tumor_normal(**kwargs)
    rng = np.random.default_rng(71)
    if NAME == "leak_small":
        unmap = lambda rs: [synth.SamRead(r.qname, r.flag | (0x4 if 1500 <= r.pos - 1 < 2250 else 0), r.rname, r.pos, r.mapq, r.cigar, r.seq, r.qual, r.tags) for r in rs]
        reads = {"tumor": unmap(synth.pairs_to_sorted_reads(data["tumor"])), "normal": unmap(synth.pairs_to_sorted_reads(data["normal"]))}
    elif NAME == "rg_small":
        def tag_rg(rs):
            out = []
            for r in rs:
                tags = dict(r.tags); u = rng.random()
                if u < 0.45: tags["RG"] = "rgA"
                elif u < 0.70: tags["RG"] = "rgB"
                elif u < 0.85: tags["RG"] = "rgC"
                out.append(synth.SamRead(r.qname, r.flag, r.rname, r.pos, r.mapq, r.cigar, r.seq, r.qual, tags))
            return out
        reads = {"tumor": tag_rg(synth.pairs_to_sorted_reads(data["tumor"])), "normal": tag_rg(synth.pairs_to_sorted_reads(data["normal"]))}
    else:
        reads = {"tumor": read_variety.decorate(synth.pairs_to_sorted_reads(data["tumor"]), rng, linked),
                 "normal": read_variety.decorate(synth.pairs_to_sorted_reads(data["normal"]), rng, linked)}
    rname, ref = data["rname"], data["ref"]
    with tempfile.TemporaryDirectory(prefix="lancet_golden_") as td:
        fa = os.path.join(td, "ref.fa")
        synth.write_fasta(fa, rname, ref)
        bams = {}
        for sample, rg in (("TUMOR", "tumor"), ("NORMAL", "normal")):
            sam, bam = os.path.join(td, f"{rg}.sam"), os.path.join(td, f"{rg}.bam")
            with open(sam, "w") as f:
                extra_rg = [f"@RG\tID:{x}\tSM:{sample}\tPL:ILLUMINA" for x in ("rgA", "rgB", "rgC")] if NAME == "rg_small" else []
                f.write("\n".join(["@HD\tVN:1.6\tSO:coordinate", f"@SQ\tSN:{rname}\tLN:{len(ref)}", f"@RG\tID:{rg}\tSM:{sample}\tPL:ILLUMINA"] + extra_rg
                                  + [read_variety.sam_line(r) for r in reads[rg]]) + "\n")
            mg.run([mg.TEST_VIEW, "-b", "-p", bam, sam], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            mg.run([mg.BAMTOOLS, "index", "-in", bam])
            bams[rg] = bam
        rgfile = os.path.join(td, "rg.txt")
        open(rgfile, "w").write("rgA\nrgC\n")
        cmd = [mg.REF_BIN, "--tumor", bams["tumor"], "--normal", bams["normal"], "--ref", fa, "--reg", REGION, "--num-threads", "1", "-v"] + [rgfile if x == "RGFILE" else x for x in FLAGS]
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=td)
        if r.returncode != 0:
            sys.stderr.write(r.stderr[-3000:])
            raise SystemExit("reference failed")
        for rg in ("tumor", "normal"):
            shutil.copy(bams[rg], os.path.join(mg.GOLDEN, f"{NAME}.{rg}.bam"))
        shutil.copy(fa, os.path.join(mg.GOLDEN, f"{NAME}.fa"))
        if NAME == "rg_small":
            shutil.copy(rgfile, os.path.join(mg.GOLDEN, f"{NAME}.rg.txt"))
    vcf = "".join(l + "\n" for l in r.stdout.splitlines()
                  if not l.startswith("##fileDate") and not l.startswith("##cmdline") and not l.startswith("##reference"))
    open(os.path.join(mg.GOLDEN, f"{NAME}.vcf"), "w").write(vcf)
    open(os.path.join(mg.GOLDEN, f"{NAME}.trace.txt"), "w").write(mg.digest_trace(r.stderr))
    json.dump({"region": REGION, "flags": FLAGS + ([] if "--active-region-off" in FLAGS else ["--active-region-on"]),
               "reference_cmd": " ".join(os.path.basename(c) if c.startswith("/tmp") else c for c in cmd),
               "n_vcf_records": sum(1 for l in vcf.splitlines() if not l.startswith("#"))},
              open(os.path.join(mg.GOLDEN, f"{NAME}.case.txt"), "w"), indent=1)
    print(NAME, sum(1 for l in vcf.splitlines() if not l.startswith("#")), "VCF records;", len(reads["tumor"]), "+", len(reads["normal"]), "reads;",
          mg.digest_trace(r.stderr).count("== Processing"), "windows assembled")
