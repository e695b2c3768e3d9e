#!/usr/bin/env python3
"""Golden case `bed2`: the REFERENCE ITSELF (same binary and recipe as tools/make_golden.py) on TWO contigs with a BED file
and a --reg region together -- loadBed + loadRefs into one window table (reference src/Lancet.cc:852-857), BED intervals
padded twice (:343-349 then :233-249), overlapping BED lines tiling the same window header twice, windows of both contigs in
one header-ordered processing sequence, active regions on.

Written to tests/golden/: bed2.{tumor,normal}.bam + .bam.bai (made by htslib's test_view / `bamtools index`), bed2.fa (+ .fai
as htslib's faidx wrote it), bed2.bed (inputs); bed2.vcf (expected output; bed2_bedonly.vcf: the same run without --reg), bed2.trace.txt (digest of the reference's -v),
bed2.case.txt (the command line, JSON).  Nothing of the reference travels; only these data files do."""
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import read_variety  # noqa: E402
from lancet_amd import synth  # noqa: E402
import make_golden as mg  # noqa: E402

NAME = "bed2"
BED = "# two contigs; the two chr22 lines overlap once padded\nchr22\t1300\t1700\nchr21\t900\t1500\nchr22\t1500\t2100\n"
REGION = "chr21:2300-2700"

if __name__ == "__main__":
    mg.check_reference_is_unmodified()
    a = synth.make_tumor_normal(ref_len=3600, cov_t=30, cov_n=26, ref_seed=401, tumor_seed=1401, normal_seed=2401, somatic_every=500, germline_every=400)
    b = synth.make_tumor_normal(ref_len=4000, cov_t=30, cov_n=26, ref_seed=402, tumor_seed=1402, normal_seed=2402, somatic_every=600, germline_every=450)
    ren = lambda rs, name: [synth.SamRead(r.qname + "_" + name, r.flag, name, r.pos, r.mapq, r.cigar, r.seq, r.qual, r.tags) for r in rs]
    reads = {rg: ren(synth.pairs_to_sorted_reads(a[rg]), "chr21") + ren(synth.pairs_to_sorted_reads(b[rg]), "chr22") for rg in ("tumor", "normal")}
    with tempfile.TemporaryDirectory(prefix="lancet_golden_") as td:
        fa = os.path.join(td, "ref.fa")
        with open(fa, "w") as fh:
            for name, seq in (("chr21", a["ref"]), ("chr22", b["ref"])):
                fh.write(f">{name}\n")
                for i in range(0, len(seq), 60):
                    fh.write(seq[i:i + 60] + "\n")
        bed = os.path.join(td, "regions.bed")
        open(bed, "w").write(BED)
        bams = {}
        for sample, rg in (("TUMOR", "tumor"), ("NORMAL", "normal")):
            sam, bam = os.path.join(td, f"{rg}.sam"), os.path.join(td, f"{rg}.bam")
            with open(sam, "w") as f:
                f.write("\n".join(["@HD\tVN:1.6\tSO:coordinate", f"@SQ\tSN:chr21\tLN:{len(a['ref'])}", f"@SQ\tSN:chr22\tLN:{len(b['ref'])}",
                                   f"@RG\tID:{rg}\tSM:{sample}\tPL:ILLUMINA"] + [read_variety.sam_line(r) for r in reads[rg]]) + "\n")
            mg.run([mg.TEST_VIEW, "-b", "-p", bam, sam], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            mg.run([mg.BAMTOOLS, "index", "-in", bam])
            bams[rg] = bam
        cmd = [mg.REF_BIN, "--tumor", bams["tumor"], "--normal", bams["normal"], "--ref", fa, "--bed", bed, "--reg", REGION, "--num-threads", "1", "-v"]
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=td)
        if r.returncode != 0:
            sys.stderr.write(r.stderr[-3000:])
            raise SystemExit("reference failed")
        # the BED file alone (no --reg)
        r2 = subprocess.run([c for c in cmd if c not in ("--reg", REGION, "-v")], capture_output=True, text=True, cwd=td)
        if r2.returncode != 0:
            raise SystemExit("reference failed (BED only)")
        for rg in ("tumor", "normal"):
            shutil.copy(bams[rg], os.path.join(mg.GOLDEN, f"{NAME}.{rg}.bam"))
            shutil.copy(bams[rg] + ".bai", os.path.join(mg.GOLDEN, f"{NAME}.{rg}.bam.bai"))
        shutil.copy(fa, os.path.join(mg.GOLDEN, f"{NAME}.fa"))
        shutil.copy(fa + ".fai", os.path.join(mg.GOLDEN, f"{NAME}.fa.fai"))
        shutil.copy(bed, os.path.join(mg.GOLDEN, f"{NAME}.bed"))
    vcf = "".join(l + "\n" for l in r.stdout.splitlines()
                  if not l.startswith("##fileDate") and not l.startswith("##cmdline") and not l.startswith("##reference"))
    open(os.path.join(mg.GOLDEN, f"{NAME}.vcf"), "w").write(vcf)
    vcf2 = "".join(l + "\n" for l in r2.stdout.splitlines()
                   if not l.startswith("##fileDate") and not l.startswith("##cmdline") and not l.startswith("##reference"))
    open(os.path.join(mg.GOLDEN, f"{NAME}_bedonly.vcf"), "w").write(vcf2)
    open(os.path.join(mg.GOLDEN, f"{NAME}.trace.txt"), "w").write(mg.digest_trace(r.stderr))
    json.dump({"region": REGION, "bed": BED, "flags": ["--active-region-on"],
               "reference_cmd": " ".join(os.path.basename(c) if c.startswith("/tmp") else c for c in cmd),
               "n_vcf_records": sum(1 for l in vcf.splitlines() if not l.startswith("#"))},
              open(os.path.join(mg.GOLDEN, f"{NAME}.case.txt"), "w"), indent=1)
    print(NAME, sum(1 for l in vcf.splitlines() if not l.startswith("#")), "VCF records;", len(reads["tumor"]), "+", len(reads["normal"]), "reads;",
          mg.digest_trace(r.stderr).count("== Processing"), "windows assembled")
