"""Pins the oracle (oracle/lancet_oracle.cc + oracle/vcf_oracle.py) against outputs of the reference itself.

tests/golden/<case>.vcf and <case>.trace.txt were written by the unmodified reference binary
(tools/make_golden.py).  The oracle has to reproduce, byte for byte,
  * the VCF header + body (whole-program parity, --num-threads 1, --active-region-off), and
  * the digest of the reference's `-v` trace: every k attempt and its rejection reason, node/edge/span counts
    after each graph stage, source/sink anchors, every path and every transcript with its coverages.
CPU only."""
import numpy as np
import pytest

import golden_util as gu
from lancet_amd import abi
from oracle import oracle, vcf_oracle


@pytest.mark.parametrize("case", gu.CASES)
def test_oracle_reproduces_reference_vcf_and_trace(case):
    meta, batch, kept, (min_k, max_k) = gu.case_batch(case)
    lr = gu.case_lr(meta)                     # --linked-reads case
    variants, stats, trace = oracle.run(batch, gu.params(meta), verbose=True)
    db = vcf_oracle.VariantDB(lr=lr)
    for rec in variants:                      # replay addVar in window-processing order
        db.add(vcf_oracle.Variant(batch.chrom[rec["window"]], rec, lr=lr, bx_names=batch.bx_names))
    assert db.vcf() == gu.golden_vcf(case)
    assert gu.digest_trace(trace) == gu.golden_trace(case)
    assert sum(s["n_variants"] for s in stats) == len(variants)


def test_golden_cases_cover_the_k_loop_branches():
    """The fixtures must exercise: ref-repeat rejections, cycle-driven k bumps, k up to the 40s, multi-path
    windows, complex/ins/del/snv transcripts."""
    all_trace = "".join(gu.golden_trace(c) for c in gu.CASES)
    assert all_trace.count("Cycle found in the graph") > 100
    assert all_trace.count("Repeat in reference sequence") > 100
    assert all_trace.count("Near-perfect repeat in reference") > 100
    body = "".join(l for c in gu.CASES for l in gu.golden_vcf(c).splitlines(True) if not l.startswith("#"))
    for t in ("TYPE=snv", "TYPE=ins", "TYPE=del", "TYPE=complex", "SOMATIC", "SHARED", ";MS="):
        assert t in body or t == ";MS=", t
    ks = {int(l.split("KMERSIZE=")[1].split(";")[0]) for l in body.splitlines()}
    assert max(ks) >= 45 and min(ks) <= 13


def _rand_seq(rng, n):
    return "".join("ACGT"[i] for i in rng.integers(0, 4, size=n))


@pytest.mark.skipif(not oracle.ref_align_available(), reason="oracle/_ref/libalign_ref.so not built")
def test_alignment_restatement_equals_reference_align_cc():
    """oracle global_align_aff vs the reference's own align.cc compiled into oracle/_ref (real reference code)."""
    rng = np.random.default_rng(5)
    for it in range(60):
        n = int(rng.integers(30, 260))     # tiny S hits undefined behaviour in the reference traceback
        s = _rand_seq(rng, n)
        t = list(s)
        for _ in range(int(rng.integers(0, 6))):          # a few edits: sub / ins / del (some long)
            p = int(rng.integers(0, max(1, len(t))))
            r = rng.random()
            if r < 0.34 and t:
                t[p] = "ACGT"[int(rng.integers(0, 4))]
            elif r < 0.67:
                t[p:p] = list(_rand_seq(rng, int(rng.integers(1, 40))))
            elif t:
                del t[p:p + int(rng.integers(1, 40))]
        t = "".join(t) or "A"
        assert oracle.align(s, t) == oracle.ref_align(s, t), (s, t)
    # degenerate shapes
    for s, t in (("A", "A"), ("A", "C"), ("ACGT", "A"), ("A", "ACGT"), ("AAAAAAAA", "AAAA"), ("ACACACAC", "ACAC")):
        assert oracle.align(s, t) == oracle.ref_align(s, t)


def test_std_hash_known_answers():
    """libstdc++ std::hash<std::string> values that node-table order depends on."""
    kat = {"ACGT": 2120050921807290424, "source1": 5824865595435910722, "ACGTACGTACGTA": 613407185026648748,
           "ACGTACGTACGTACGT": 8653344979867301840, "AAAAACCCCCGGGGGTTTTTACGTA": 11793684898630242514}
    for s, h in kat.items():
        assert oracle.std_hash(s) == h
