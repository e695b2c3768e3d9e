"""`python -m lancet_amd.cli` -- tumor/normal scan with the reference's command line (reference src/Lancet.cc:655-800).

    python -m lancet_amd.cli --tumor T.bam --normal N.bam --ref ref.fa --reg chr22:1000-5000 > out.vcf

Host side (this file, bamio.py, frontend.py): decode the BAMs once, tile the region into windows
(Lancet.cc:189-316), active-region prefilter (Microassembler.cc:253-432), per-window read selection
(:436-655), batching.  Hot path: `lancet_engine_process` on the GPU (there is no CPU path: without the HIP
library or without a device this program stops with an error).  Below it: `lancet_vdb_*` (VariantDB + VCF).
The VCF is byte-identical to the reference's for the same inputs (tests/test_cli.py, reference-made fixture).

Not offered: --bed, --rg-file, --kmer-recovery, --print-graph (graph dumps), --num-threads (accepted, ignored)."""
from __future__ import annotations

import argparse
import ctypes as C
import sys
import time
from typing import List, Optional

from . import abi, bamio, engine, frontend


def build_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(prog="lancet_amd", description="MI355X micro-assembly somatic variant scan (Lancet command line)")
    ap.add_argument("--tumor", "-t", required=True)
    ap.add_argument("--normal", "-n", required=True)
    ap.add_argument("--ref", "-r", required=True)
    ap.add_argument("--reg", "-p", required=True, help="chr:start-end (or chr)")
    ap.add_argument("--min-k", "-k", type=int, default=11)
    ap.add_argument("--max-k", "-K", type=int, default=101)
    ap.add_argument("--trim-lowqual", "-q", type=int, default=10)
    ap.add_argument("--min-base-qual", "-C", type=int, default=17)
    ap.add_argument("--quality-range", "-Q", default="!")
    ap.add_argument("--min-map-qual", "-b", type=int, default=15)
    ap.add_argument("--max-as-xs-diff", "-Z", type=int, default=5)
    ap.add_argument("--tip-len", "-l", type=int, default=11)
    ap.add_argument("--cov-thr", "-c", type=int, default=5)
    ap.add_argument("--cov-ratio", "-x", type=float, default=0.01)
    ap.add_argument("--low-cov", "-d", type=int, default=1)
    ap.add_argument("--max-avg-cov", "-u", type=int, default=10000)
    ap.add_argument("--window-size", "-w", type=int, default=600)
    ap.add_argument("--padding", "-P", type=int, default=250)
    ap.add_argument("--dfs-limit", "-F", type=int, default=1000000)
    ap.add_argument("--max-indel-len", "-T", type=int, default=500)
    ap.add_argument("--max-mismatch", "-M", type=int, default=2)
    ap.add_argument("--num-threads", "-X", type=int, default=1, help="accepted for compatibility; windows are batched on the GPU")
    ap.add_argument("--min-alt-count-tumor", "-a", type=int, default=3)
    ap.add_argument("--max-alt-count-normal", "-m", type=int, default=0)
    ap.add_argument("--min-vaf-tumor", "-e", type=float, default=0.04)
    ap.add_argument("--max-vaf-normal", "-i", type=float, default=0.0)
    ap.add_argument("--min-coverage-tumor", "-o", type=int, default=4)
    ap.add_argument("--max-coverage-tumor", "-y", type=int, default=1000000)
    ap.add_argument("--min-coverage-normal", "-z", type=int, default=10)
    ap.add_argument("--max-coverage-normal", "-j", type=int, default=1000000)
    ap.add_argument("--min-phred-fisher", "-s", type=float, default=5.0)
    ap.add_argument("--min-phred-fisher-str", "-E", type=float, default=25.0)
    ap.add_argument("--min-strand-bias", "-f", type=float, default=1.0)
    ap.add_argument("--max-unit-length", "-U", type=int, default=4)
    ap.add_argument("--min-report-unit", "-N", type=int, default=3)
    ap.add_argument("--min-report-len", "-Y", type=int, default=7)
    ap.add_argument("--dist-from-str", "-D", type=int, default=1)
    ap.add_argument("--linked-reads", "-J", action="store_true")
    ap.add_argument("--primary-alignment-only", "-I", action="store_true")
    ap.add_argument("--XA-tag-filter", "-O", action="store_true", dest="xa_filter")
    ap.add_argument("--active-region-off", "-W", action="store_true")
    ap.add_argument("--verbose", "-v", action="store_true", help="print the reference's per-window stage trace to stderr")
    ap.add_argument("--device", type=int, default=0, help="GPU index (not a reference option)")
    ap.add_argument("--batch-windows", type=int, default=32768, help="windows per engine batch (not a reference option)")
    ap.add_argument("--strict", action="store_true", help="write no VCF when a window exceeded the engine's work space (default: finish, "
                    "list those windows on stderr, exit code 3; not a reference option)")
    return ap


def run(argv: Optional[List[str]] = None, out=None, date_line: Optional[str] = None) -> int:
    args = build_parser().parse_args(argv)
    out = out or sys.stdout
    qoff = ord(args.quality_range[0])
    params = abi.default_params(
        min_k=args.min_k, max_k=args.max_k, max_tip_len=args.tip_len, cov_threshold=args.cov_thr, low_cov_threshold=args.low_cov,
        dfs_limit=args.dfs_limit, max_indel_len=args.max_indel_len, max_mismatch=args.max_mismatch,
        min_qual_trim=args.trim_lowqual + qoff, min_qual_call=args.min_base_qual + qoff, max_unit_len=args.max_unit_length,
        min_report_units=args.min_report_unit, min_report_len=args.min_report_len, dist_from_str=args.dist_from_str,
        lr_mode=int(args.linked_reads), min_cov_ratio=args.cov_ratio)
    eng = engine.Engine(params, device=args.device, trace_words=(1 << 17) if args.verbose else 0)     # raises without a GPU

    hdr_t, tumor = bamio.read_bam(args.tumor)
    hdr_n, normal = bamio.read_bam(args.normal)
    contigs = bamio.read_fasta(args.ref)
    chrom = args.reg.split(":")[0]
    if chrom not in contigs:
        raise SystemExit(f"contig {chrom!r} not in {args.ref}")
    tumor = [r for r in tumor if r.rname == chrom]
    normal = [r for r in normal if r.rname == chrom]
    windows = frontend.tile_region(contigs[chrom], chrom, args.reg, padding=args.padding, window_size=args.window_size)
    # --max-as-xs-diff is accepted and without effect, as in the reference: its main() parses -Z but never hands the value to the
    # assemblers (src/Lancet.cc:865-918 lacks the MAX_DELTA_AS_XS line of :496); the AS/XS filter always uses the default 5
    fp = frontend.ReadFilterParams(min_map_qual=args.min_map_qual, max_delta_as_xs=5,
                                   primary_alignment_only=args.primary_alignment_only, xa_filter=args.xa_filter,
                                   max_avg_cov=args.max_avg_cov)

    filters = abi.LancetFilters()
    engine.lib().lancet_filters_default(C.byref(filters))
    filters.min_phred_fisher_str, filters.min_phred_fisher = args.min_phred_fisher_str, args.min_phred_fisher
    filters.max_vaf_normal, filters.min_vaf_tumor = args.max_vaf_normal, args.min_vaf_tumor
    filters.min_cov_normal, filters.max_cov_normal = args.min_coverage_normal, args.max_coverage_normal
    filters.min_cov_tumor, filters.max_cov_tumor = args.min_coverage_tumor, args.max_coverage_tumor
    filters.min_alt_cnt_tumor, filters.max_alt_cnt_normal = args.min_alt_count_tumor, args.max_alt_count_normal
    filters.min_strand_bias = int(args.min_strand_bias)
    db = engine.VariantDB(filters)

    # windows in the reference's processing order, a batch at a time (records come back in window order, so the
    # addVar replay order is the reference's whatever the batch size)
    ordered = frontend.windows_in_processing_order(windows)
    step = max(1, args.batch_windows)
    n_done = 0
    overflowed: List[str] = []
    leak: list = []                       # reads a window without mapped reads leaves in the reference's graph (frontend.batch_from_sam)
    for lo in range(0, len(ordered), step):
        chunk = ordered[lo:lo + step]
        batch, kept = frontend.batch_from_sam(chunk, tumor, normal, fp, max_k=args.max_k, linked=args.linked_reads,
                                              active_region=not args.active_region_off, min_evidence=args.min_alt_count_tumor,
                                              min_qual_call=args.min_base_qual + qoff, leak=leak)
        if batch.n_windows == 0:
            continue
        _, stats = eng.process(batch)
        bad = [kept[w].hdr for w in range(batch.n_windows) if stats[w]["status"] < 0]
        if bad and args.strict:
            raise SystemExit(f"work-space overflow in {len(bad)} window(s), e.g. {bad[0]}: results withheld (--strict)")
        overflowed += bad               # such a window contributes nothing (the engine drops its records); the run goes on
        vp, n, blob, _ = eng.raw_results()
        if args.linked_reads:
            lp, bp, _ = eng.raw_results_lr()
            db.add_raw_lr(vp, lp, n, blob + b"\0", bp, batch.bx_names, [chrom])
        else:
            db.add_raw(vp, n, blob + b"\0", [chrom])
        if args.verbose:
            sys.stderr.write(eng.trace_text())
        n_done += batch.n_windows
    sample_n = hdr_n["samples"][0] if hdr_n["samples"] else "NA"       # Microassembler::retriveSampleName, src/Microassembler.cc:52-67
    sample_t = hdr_t["samples"][0] if hdr_t["samples"] else "NA"
    cmdline = "lancet " + " ".join(sys.argv[1:] if argv is None else argv)
    out.write(db.vcf(cmdline=cmdline, reference=args.ref, date_line=date_line or time.strftime("%a %b %e %H:%M:%S %Y\n"),
                     sample_normal=sample_n, sample_tumor=sample_t))
    sys.stderr.write(f"[lancet_amd] {len(windows)} windows tiled, {n_done} assembled on GPU {args.device}, {db.size()} variants\n")
    eng.close()
    if overflowed:
        sys.stderr.write(f"[lancet_amd] {len(overflowed)} window(s) exceeded the engine's work space and contributed NO variants:\n"
                         + "".join(f"[lancet_amd]   {h}\n" for h in overflowed))
        return 3
    return 0


if __name__ == "__main__":
    sys.exit(run())
