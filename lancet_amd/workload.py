"""Vectorised synthetic scan workload for bench.py (SURVEY.md §8(d) "Synthetic inputs", config 2 proxy):
a random contig with planted germline/somatic variants, paired 2x150 bp reads at the requested tumor/normal
coverage, tiled into 600 bp windows with stride 100, reads bucketed per window with the reference's containment
rule.  Produces the same `WindowBatch` the front-end produces from SAM records, just much faster (numpy only).
Input generation only -- not part of the hot path."""
from __future__ import annotations

from typing import List, Tuple

import numpy as np

from . import synth
from .frontend import FWD, NML, REV, TMR, WindowBatch

_QUAL_LEVELS = np.array([37, 30, 25, 12], dtype=np.uint8)
_QUAL_P = np.array([0.70, 0.15, 0.08, 0.07])
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_CODE = np.full(256, 4, dtype=np.uint8)
for _i, _c in enumerate(b"ACGT"):
    _CODE[_c] = _i


def _simulate(haps, probs, coverage: float, span: Tuple[int, int], rng, read_len: int, ins_mean: float, ins_sd: float,
              err: float, linked: bool = False):
    """Returns dict of arrays for all reads of one sample: seq[R,L] (ASCII), qual[R,L] (+33), pos0[R], end0[R],
    frag[R], mate[R] (1|2), strand[R]; linked: also bx[R] (barcode code of the fragment, -1: no BX tag) and hp[R] (0|1|2), drawn as
    synth.simulate_sample draws them (a pool of n_frag / 3 barcodes, 10 % of the fragments without one; HP 0 for 30 %, else the haplotype
    with 3 % phasing errors; 5 % of the reads without the tag)."""
    lo, hi = span
    n_frag = int(coverage * (hi - lo) / (2.0 * read_len))
    hsel = rng.choice(len(haps), size=n_frag, p=probs)
    ins = np.maximum(read_len, np.rint(rng.normal(ins_mean, ins_sd, size=n_frag)).astype(np.int64))
    s_ref = rng.integers(lo, np.maximum(lo + 1, hi - ins))
    first_fwd = rng.integers(0, 2, size=n_frag).astype(bool)
    out = {k: [] for k in ("seq", "qual", "pos0", "end0", "frag", "mate", "strand") + (("bx", "hp") if linked else ())}
    ar = np.arange(read_len)
    if linked:
        rng_lr = np.random.default_rng(int(rng.integers(0, 2 ** 31)))
        f_code = rng_lr.integers(0, max(4, n_frag // 3), size=n_frag).astype(np.int64)
        f_code[rng_lr.random(n_frag) < 0.1] = -1
        f_hp = np.where(rng_lr.random(n_frag) < 0.3, 0, np.where(hsel == 0, 1, 2)).astype(np.uint8)
        flip = rng_lr.random(n_frag) < 0.03
        f_hp = np.where(flip, np.where(f_hp == 0, 1, 3 - f_hp), f_hp).astype(np.uint8)
    for h, (hb, hp) in enumerate(haps):
        idx = np.nonzero(hsel == h)[0]
        if idx.size == 0:
            continue
        s = np.searchsorted(hp, s_ref[idx])          # hp is non-decreasing apart from -1 runs
        e = s + ins[idx]
        ok = e <= len(hb)
        idx, s, e = idx[ok], s[ok], e[ok]
        for which in (0, 1):
            a = s if which == 0 else e - read_len
            m = a[:, None] + ar[None, :]
            seq = hb[m]
            rp = hp[m]
            good = (rp[:, 0] >= 0) & (rp[:, -1] >= 0)
            n = int(good.sum())
            seq, rp = seq[good].copy(), rp[good]
            q = _QUAL_LEVELS[rng.choice(4, size=(n, read_len), p=_QUAL_P)]
            em = rng.random((n, read_len)) < err
            codes = _CODE[seq]
            codes = np.where(em, (codes + rng.integers(1, 4, size=codes.shape)) % 4, codes)
            seq = _ACGT[codes]
            q = np.where(em, rng.integers(8, 21, size=q.shape).astype(np.uint8), q)
            out["seq"].append(seq)
            out["qual"].append((q + 33).astype(np.uint8))
            out["pos0"].append(rp[:, 0].astype(np.int64))
            out["end0"].append(rp[:, -1].astype(np.int64) + 1)
            out["frag"].append(idx[good])
            is_first = (which == 0) == first_fwd[idx[good]]
            out["mate"].append(np.where(is_first, 1, 2).astype(np.uint8))
            out["strand"].append(np.full(n, REV if which == 1 else FWD, dtype=np.uint8))
            if linked:
                fr = idx[good]
                out["bx"].append(f_code[fr])
                out["hp"].append(np.where(rng_lr.random(n) < 0.05, 0, f_hp[fr]).astype(np.uint8))
    res = {k: np.concatenate(v) for k, v in out.items()}
    order = np.lexsort((res["mate"], res["frag"], res["pos0"]))     # coordinate order, ties by name
    return {k: v[order] for k, v in res.items()}


def make_scan_batch(n_windows: int, cov_t: float = 30.0, cov_n: float = 30.0, seed: int = 22, read_len: int = 150,
                    window: int = 600, stride: int = 100, error_rate: float = 0.005, somatic_every: int = 2000,
                    germline_every: int = 1000, chrom: str = "chr22", str_fraction: float = 0.0,
                    lowcomplex_fraction: float = 0.0, linked: bool = False) -> WindowBatch:
    margin = 1000
    region_len = window + stride * (n_windows - 1)
    ref_len = region_len + 2 * margin
    ref = synth.random_reference(ref_len, seed, str_fraction, lowcomplex_fraction)      # (0.30 / 0.05)
    variants = synth.plant_variants(ref, seed + 1, somatic_every, germline_every)
    germ = [v for v in variants if not v.somatic]
    h0 = synth.build_haplotype(ref, [])
    h1 = synth.build_haplotype(ref, germ)
    h2 = synth.build_haplotype(ref, variants)
    rng_t = np.random.default_rng(seed + 101)
    rng_n = np.random.default_rng(seed + 202)
    span = (margin - 400, margin + region_len + 400)
    T = _simulate([h0, h1, h2], [0.5, 0.25, 0.25], cov_t, span, rng_t, read_len, 400.0, 40.0, error_rate, linked)
    N = _simulate([h0, h1], [0.5, 0.5], cov_n, span, rng_n, read_len, 400.0, 40.0, error_rate, linked)
    ref_b = np.frombuffer(ref.encode(), dtype=np.uint8)
    starts = margin + 1 + stride * np.arange(n_windows, dtype=np.int64)          # Ref_t::refstart (1-based)
    sel_idx: List[np.ndarray] = []
    sel_lab: List[np.ndarray] = []
    read_begin = np.zeros(n_windows + 1, dtype=np.uint32)
    ranks: List[np.ndarray] = []
    for w in range(n_windows):
        rs, re_ = int(starts[w]), int(starts[w]) + window
        parts = []
        for lab, S in ((TMR, T), (NML, N)):
            lo = int(np.searchsorted(S["pos0"], rs, side="left"))
            hi = int(np.searchsorted(S["pos0"], re_, side="right"))
            k = lo + np.nonzero(S["end0"][lo:hi] <= re_)[0]                  # alstart >= Left && alend <= Right
            parts.append((lab, k))
        nt, nn = len(parts[0][1]), len(parts[1][1])
        key = np.concatenate([T["frag"][parts[0][1]] + 10 ** 9, N["frag"][parts[1][1]]])   # 'N%07d' < 'T%07d'
        _, inv = np.unique(key, return_inverse=True)
        ranks.append(inv.astype(np.uint32))
        sel_idx.append(parts[0][1]); sel_lab.append(np.full(nt, TMR, dtype=np.uint8))
        sel_idx.append(parts[1][1]); sel_lab.append(np.full(nn, NML, dtype=np.uint8))
        read_begin[w + 1] = read_begin[w] + nt + nn
    labels = np.concatenate(sel_lab)
    seqs, quals, strand, mate, bxs, hps = [], [], [], [], [], []
    for i, k in enumerate(sel_idx):
        S = T if (i % 2 == 0) else N
        seqs.append(S["seq"][k]); quals.append(S["qual"][k]); strand.append(S["strand"][k]); mate.append(S["mate"][k])
        if linked:
            c = S["bx"][k]
            bxs.append(np.where(c >= 0, 2 * c + (i % 2), -1)); hps.append(S["hp"][k])      # (the two samples' barcodes: disjoint names)
    lr = {}
    if linked:
        code = np.concatenate(bxs) if bxs else np.zeros(0, dtype=np.int64)
        uniq, inv = np.unique(code[code >= 0], return_inverse=True)                          # names "B%012d-1": string order = numeric order
        rank = np.full(code.shape, 0xFFFFFFFF, dtype=np.uint32)
        rank[code >= 0] = inv.astype(np.uint32)
        lr = dict(bx_rank=rank, hp=np.concatenate(hps).astype(np.uint8) if hps else np.zeros(0, dtype=np.uint8),
                  bx_names=[f"B{int(u):012d}-1" for u in uniq])
    seq = np.concatenate(seqs).reshape(-1)
    qual = np.concatenate(quals).reshape(-1)
    R = int(read_begin[-1])
    refs = [ref_b[s - 1: s - 1 + window] for s in starts]
    hdr = [f"{chrom}:{int(s)}-{int(s) + window}" for s in starts]
    return WindowBatch(
        n_windows=n_windows, hdr=hdr, chrom=[chrom] * n_windows, chr_id=np.zeros(n_windows, dtype=np.int32),
        ref_start=starts.astype(np.int32), ref_off=(window * np.arange(n_windows + 1)).astype(np.uint32),
        ref_bases=np.concatenate(refs).copy(), read_begin=read_begin,
        seq_off=(read_len * np.arange(R + 1, dtype=np.uint64)).astype(np.uint32), seq=np.ascontiguousarray(seq),
        qual=np.ascontiguousarray(qual), label=labels, strand=np.concatenate(strand), mate=np.concatenate(mate),
        mapped=np.ones(R, dtype=np.uint8), name_rank=np.concatenate(ranks) if ranks else np.zeros(0, dtype=np.uint32), **lr)


def sub_batch(b: WindowBatch, w0: int, w1: int) -> WindowBatch:
    """Windows [w0, w1) of a batch as a new batch."""
    r0, r1 = int(b.read_begin[w0]), int(b.read_begin[w1])
    s0, s1 = int(b.seq_off[r0]), int(b.seq_off[r1])
    f0, f1 = int(b.ref_off[w0]), int(b.ref_off[w1])
    return WindowBatch(
        n_windows=w1 - w0, hdr=b.hdr[w0:w1], chrom=b.chrom[w0:w1], chr_id=b.chr_id[w0:w1].copy(),
        ref_start=b.ref_start[w0:w1].copy(), ref_off=(b.ref_off[w0:w1 + 1] - f0).astype(np.uint32),
        ref_bases=b.ref_bases[f0:f1].copy(), read_begin=(b.read_begin[w0:w1 + 1] - r0).astype(np.uint32),
        seq_off=(b.seq_off[r0:r1 + 1] - s0).astype(np.uint32), seq=b.seq[s0:s1].copy(), qual=b.qual[s0:s1].copy(),
        label=b.label[r0:r1].copy(), strand=b.strand[r0:r1].copy(), mate=b.mate[r0:r1].copy(),
        mapped=b.mapped[r0:r1].copy(), name_rank=b.name_rank[r0:r1].copy(),
        bx_rank=None if b.bx_rank is None else b.bx_rank[r0:r1].copy(), hp=None if b.hp is None else b.hp[r0:r1].copy(),
        bx_names=b.bx_names)      # (barcode ranks stay ranks among the parent batch's barcodes)


def concat_batches(parts) -> WindowBatch:
    """Several batches (short reads, same contig table) as one: the windows of parts[0], then parts[1], ..."""
    parts = list(parts)
    if len(parts) == 1:
        return parts[0]
    def offs(name, base_of):
        out, base = [np.zeros(1, dtype=np.int64)], 0
        for p in parts:
            a = getattr(p, name).astype(np.int64)
            out.append(a[1:] + base); base += int(a[-1])
        return np.concatenate(out).astype(np.uint32)
    cat = lambda name: np.concatenate([getattr(p, name) for p in parts])
    return WindowBatch(
        n_windows=sum(p.n_windows for p in parts), hdr=[h for p in parts for h in p.hdr], chrom=[c for p in parts for c in p.chrom],
        chr_id=cat("chr_id"), ref_start=cat("ref_start"), ref_off=offs("ref_off", None), ref_bases=cat("ref_bases"),
        read_begin=offs("read_begin", None), seq_off=offs("seq_off", None), seq=cat("seq"), qual=cat("qual"), label=cat("label"),
        strand=cat("strand"), mate=cat("mate"), mapped=cat("mapped"), name_rank=cat("name_rank"))


def algorithmic_bytes_split(b: WindowBatch, stats, n_variants: int):
    """The same bytes by the kernel whose work they are: (graph construction -- reads, reference, 16 B per k-mer occurrence: the LDS build
    kernel; graph passes, paths, alignment, records -- 40 B per node, 128 B per variant: the window kernel).  Their sum is algorithmic_bytes."""
    build = win = 0
    lens = np.diff(b.seq_off.astype(np.int64))
    per_read = (lens + 3) // 4 + lens
    csum = np.concatenate([[0], np.cumsum(per_read)])
    for w in range(b.n_windows):
        st = stats[w]
        W = int(b.ref_off[w + 1] - b.ref_off[w])
        reads_b = int(csum[int(b.read_begin[w + 1])] - csum[int(b.read_begin[w])])
        build += st["n_builds"] * (reads_b + (W + 3) // 4 + W // 8) + 16 * st["n_kmers"]
        win += 40 * st.get("sum_nodes", st["n_builds"] * st["max_nodes"])      # (every build its own node table)
    return int(build), int(win + 128 * n_variants)


def algorithmic_bytes(b: WindowBatch, stats, n_variants: int) -> int:
    """per (window, k-attempt that reaches buildgraph)
         sum_reads(ceil(len/4) + len) + ceil(W/4) + W/8 + 16*kmers + 40*nodes      (+128 B per emitted variant)."""
    total = 0
    lens = np.diff(b.seq_off.astype(np.int64))
    per_read = (lens + 3) // 4 + lens
    csum = np.concatenate([[0], np.cumsum(per_read)])
    for w in range(b.n_windows):
        st = stats[w]
        W = int(b.ref_off[w + 1] - b.ref_off[w])
        reads_b = int(csum[int(b.read_begin[w + 1])] - csum[int(b.read_begin[w])])
        total += st["n_builds"] * (reads_b + (W + 3) // 4 + W // 8) + 40 * st.get("sum_nodes", st["n_builds"] * st["max_nodes"]) + 16 * st["n_kmers"]
    return int(total + 128 * n_variants)


def kernel_fingerprint() -> str:
    """sha1 over everything that decides what the device code is: the device headers of lancet_amd/csrc (kernels.h, build_lds.h,
    build_lds_impl.h, wave.h, layout.h), the two .hip files (launch bounds, waves per SIMD, the order kernels) and the compile commands of
    lancet_amd/build.py (tuning builds pass -D knobs through HIPCC_EXTRA, which is part of them).  A measurement that was not taken by this
    very process (the PMC traffic passes, profiles/*_traffic.json) must have been taken on the same fingerprint for bench.py to quote it."""
    import hashlib
    import os
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    h = hashlib.sha1()
    for f in ("build_lds.h", "build_lds_impl.h", "kernels.h", "layout.h", "wave.h", "engine.hip", "window_fat.hip"):
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    from . import build as _b
    h.update(" ".join(_b.device_flags()).encode())
    return h.hexdigest()[:16]
