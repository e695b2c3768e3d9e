"""Synthetic tumor/normal BAM + FASTA for end-to-end runs of the command-line programs (test-side BAM writer).

    python tools/make_scan_bams.py OUTDIR [ref_len=500000] [cov_t=30] [cov_n=30]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bam_writer
from lancet_amd import synth

out = sys.argv[1]
ref_len = int(sys.argv[2]) if len(sys.argv) > 2 else 500000
cov_t = float(sys.argv[3]) if len(sys.argv) > 3 else 30.0
cov_n = float(sys.argv[4]) if len(sys.argv) > 4 else 30.0
os.makedirs(out, exist_ok=True)
t = time.time()
data = synth.make_tumor_normal(ref_len=ref_len, cov_t=cov_t, cov_n=cov_n, ref_seed=22, tumor_seed=101, normal_seed=202,
                               read_len=150, insert_mean=400.0, insert_sd=40.0, somatic_every=2000, germline_every=1000)
refs = [(data["rname"], len(data["ref"]))]
bam_writer.write_bam(os.path.join(out, "tumor.bam"), refs, synth.pairs_to_sorted_reads(data["tumor"]), sample="TUMOR", index=True)
bam_writer.write_bam(os.path.join(out, "normal.bam"), refs, synth.pairs_to_sorted_reads(data["normal"]), sample="NORMAL", index=True)
synth.write_fasta(os.path.join(out, "ref.fa"), data["rname"], data["ref"])
print(f"{out}: {ref_len} bp, {cov_t}x/{cov_n}x, {time.time() - t:.1f} s; region {data['rname']}:1000-{ref_len - 1000}")
