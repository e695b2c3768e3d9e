#!/usr/bin/env python3
"""Differential run against the REFERENCE ITSELF on random synthetic cases (this container only; the binary of
tools/make_golden.py): for each seed a random configuration (coverages, error rate, read length, insert size, STR / low
complexity content, variant density, duplications, k range, linked reads) goes through the reference and through the
oracle and the emulated kernels; VCF and -v digest must agree.  A case that does not is kept under /tmp for a golden.

    python tools/fuzz_reference.py [first_seed] [n]          (LANCET_FUZZ_LINKED=1: linked reads in every case;
                                                              LANCET_FUZZ_FAT=1: the re-run tier's source in the emulator)"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
import make_golden as mg  # noqa: E402
import golden_util as gu  # noqa: E402
import emu  # noqa: E402
from oracle import oracle, vcf_oracle  # noqa: E402


def random_case(seed):
    rng = np.random.default_rng(seed)
    linked = rng.random() < 0.25 or bool(os.environ.get("LANCET_FUZZ_LINKED"))     # (LANCET_FUZZ_LINKED=1: every case with --linked-reads)
    kw = dict(ref_len=int(rng.integers(3200, 5200)), cov_t=float(rng.choice([18, 30, 45, 70, 110])), cov_n=float(rng.choice([15, 28, 40, 60])),
              ref_seed=1000 + seed, tumor_seed=2000 + seed, normal_seed=3000 + seed, error_rate=float(rng.choice([0.0, 0.003, 0.008, 0.015])),
              read_len=int(rng.choice([76, 100, 125, 150])), insert_mean=float(rng.choice([190, 260, 330, 420])), insert_sd=float(rng.choice([20, 40, 60])),
              somatic_every=int(rng.choice([300, 600, 1200])), germline_every=int(rng.choice([250, 500, 900])),
              str_fraction=float(rng.choice([0.0, 0.0, 0.1, 0.3])), lowcomplex_fraction=float(rng.choice([0.0, 0.0, 0.05])),
              dup_prob=float(rng.choice([0.0, 0.0, 0.5, 1.0])))
    if linked:
        kw["linked"] = True
    flags = ["--linked-reads"] if linked else []
    if rng.random() < 0.4:
        lo = int(rng.choice([11, 13, 17, 21])); hi = int(rng.choice([35, 61, 85, 101]))
        flags += ["--min-k", str(lo), "--max-k", str(hi)]
    if rng.random() < 0.3:
        flags += ["--max-mismatch", str(int(rng.integers(0, 4)))]
    if rng.random() < 0.3:
        flags += ["--cov-thr", str(int(rng.integers(2, 12))), "--low-cov", str(int(rng.integers(0, 4)))]
    if rng.random() < 0.3:
        flags += ["--tip-len", str(int(rng.integers(3, 20)))]
    if rng.random() < 0.2:
        flags += ["--max-indel-len", str(int(rng.integers(20, 300)))]
    if seed >= 1000:        # (second generation of cases: N runs in the contig; even k -- known not to be bit-exact and refused by the engine)
        if rng.random() < 0.3:
            kw["n_runs"] = tuple((int(rng.integers(900, kw["ref_len"] - 900)), int(rng.choice([1, 2, 5, 20, 60]))) for _ in range(int(rng.integers(1, 5))))
        if rng.random() < 0.25:
            lo = int(rng.choice([10, 12, 16, 20])); flags = [f for f in flags]; flags += ["--min-k", str(lo)] if "--min-k" not in flags else []
    if seed >= 2000:        # (third generation: quality thresholds, STR reporting options, coverage ratio, DFS limit)
        if rng.random() < 0.4: flags += ["--trim-lowqual", str(int(rng.integers(2, 25)))]
        if rng.random() < 0.4: flags += ["--min-base-qual", str(int(rng.integers(5, 38)))]
        if rng.random() < 0.3: flags += ["--max-unit-length", str(int(rng.integers(1, 7)))]
        if rng.random() < 0.3: flags += ["--min-report-unit", str(int(rng.integers(1, 6)))]
        if rng.random() < 0.3: flags += ["--min-report-len", str(int(rng.integers(2, 14)))]
        if rng.random() < 0.3: flags += ["--dist-from-str", str(int(rng.integers(0, 5)))]
        if rng.random() < 0.3: flags += ["--cov-ratio", str(round(float(rng.choice([0.0, 0.02, 0.08, 0.2])), 2))]
        if rng.random() < 0.15: flags += ["--dfs-limit", str(int(rng.choice([50, 500, 20000])))]
    if seed >= 4000:        # (fourth generation: very low coverage, long reads, inserts shorter than the reads -- mates overlap fully)
        if rng.random() < 0.4: kw["cov_t"] = float(rng.choice([3, 6, 10])); kw["cov_n"] = float(rng.choice([3, 6, 12]))
        if rng.random() < 0.3: kw["read_len"] = 250
        if rng.random() < 0.5: kw["insert_mean"] = float(kw["read_len"]) * float(rng.choice([0.8, 1.0, 1.3])); kw["insert_sd"] = 15.0
    a = int(rng.integers(700, 1200)); b = a + int(rng.integers(900, 2200))
    if 1000 <= seed < 5000 and "--min-k" in flags and int(flags[flags.index("--min-k") + 1]) % 2 == 0:
        flags[flags.index("--min-k") + 1] = str(int(flags[flags.index("--min-k") + 1]) + 1)      # (generations 2-4 were drawn while even k was refused: kept odd so that they stay the same cases)
    if seed >= 5000:        # (fifth generation: even k with self-complementary k-mers planted in the contig)
        lo = int(rng.choice([10, 12, 14, 16, 20]))
        if "--min-k" in flags: flags[flags.index("--min-k") + 1] = str(lo)
        else: flags += ["--min-k", str(lo)]
        kw["palindromes"] = tuple((int(rng.integers(a, min(b, kw["ref_len"] - 400))), int(rng.choice([lo // 2, lo // 2 + 1, lo // 2 + 2, 12]))) for _ in range(int(rng.integers(2, 9))))
    return kw, f"chr22:{a}-{min(b, kw['ref_len'] - 400)}", flags


def main():
    if os.environ.get("LANCET_FUZZ_FAT"):
        os.environ["LANCET_NO_PREBUILD"] = "1"; emu.FAT[0] = True
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    mg.check_reference_is_unmodified()
    bad = []
    for seed in range(first, first + n):
        kw, region, flags = random_case(seed)
        name = f"fz{seed}"
        td = tempfile.mkdtemp(prefix=f"lancet_fuzz_{seed}_")
        mg.GOLDEN = td; mg.CASES[name] = (kw, region, flags)
        try:
            mg.make_case(name)
        except SystemExit as e:
            print(name, "reference failed:", e); continue
        gu.GOLDEN = td; gu.case_batch.cache_clear()
        meta, batch, kept, _ = gu.case_batch(name)
        if batch.n_windows == 0:
            print(name, 'no window left after the reference-repeat test; skipped'); continue
        p = gu.params(meta)
        lr = gu.case_lr(meta)
        ov, ost, otr = oracle.run(batch, p, verbose=True)
        db = vcf_oracle.VariantDB(lr=lr)
        for rec in ov:
            db.add(vcf_oracle.Variant(batch.chrom[rec["window"]], rec, lr=lr, bx_names=batch.bx_names))
        ok_vcf = db.vcf() == gu.golden_vcf(name)
        ok_tr = gu.digest_trace(otr) == gu.golden_trace(name)
        ev, est, etr = emu.run(batch, p, evt_cap=1 << 18)
        ok_emu = ev == ov and gu.digest_trace(etr) == gu.digest_trace(otr) and all(s["status"] >= 0 for s in est)
        status = "ok" if (ok_vcf and ok_tr and ok_emu) else f"MISMATCH vcf={ok_vcf} trace={ok_tr} emu={ok_emu}"
        print(f"{name}: {batch.n_windows} windows, {len(ov)} records, flags {flags} cov {kw['cov_t']}/{kw['cov_n']} err {kw['error_rate']} L {kw['read_len']}: {status}  [{td}]")
        sys.stdout.flush()
        if status != "ok":
            bad.append((name, td))
    print("mismatches:", bad)


if __name__ == "__main__":
    main()
