"""Deterministic synthetic tumor/normal read simulator (SURVEY.md §8(d) "Synthetic inputs").

This is *input generation* for tests, fixtures and bench.py -- it is not part of the hot path.
It produces, for a random reference contig:
  * planted germline (both samples, VAF 0.5) and somatic (tumor only, VAF 0.25) variants,
  * paired 2x150 bp reads, insert ~ N(400, 40), MAPQ 60, with CIGAR / MD / NM / AS / XS / RG,
  * SAM text (coordinate sorted) that the reference binary can consume after SAM->BAM conversion,
  * the same reads as a list of `SamRead` tuples that `lancet_amd.frontend` buckets into windows.

Everything is seeded; the same arguments always give byte-identical output.
"""
from __future__ import annotations

import dataclasses
from typing import List, Tuple

import numpy as np

BASES = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
for a, b in zip(b"ACGTN", b"TGCAN"):
    _COMP[a] = b


@dataclasses.dataclass
class SamRead:
    qname: str
    flag: int
    rname: str
    pos: int          # 1-based leftmost
    mapq: int
    cigar: str
    seq: str          # as stored in SAM (forward reference strand)
    qual: str         # phred+33, SAM orientation
    tags: dict        # e.g. {"AS": 150, "XS": 0, "NM": 1, "MD": "75A74", "RG": "tumor"}

    def sam_line(self, mate_pos: int, tlen: int) -> str:
        t = []
        for k, v in self.tags.items():
            if isinstance(v, (int, np.integer)):
                t.append(f"{k}:i:{int(v)}")
            else:
                t.append(f"{k}:Z:{v}")
        return "\t".join([self.qname, str(self.flag), self.rname, str(self.pos), str(self.mapq), self.cigar,
                          "=", str(mate_pos), str(tlen), self.seq, self.qual] + t)


def random_reference(length: int, seed: int = 22, str_fraction: float = 0.0,
                     lowcomplex_fraction: float = 0.0) -> str:
    """i.i.d. uniform ACGT; optionally with STR blocks / 2-letter low-complexity stretches (config 4)."""
    rng = np.random.default_rng(seed)
    ref = BASES[rng.integers(0, 4, size=length)].copy()
    if str_fraction > 0 or lowcomplex_fraction > 0:
        pos = 0
        while pos < length:
            r = rng.random()
            if r < str_fraction / 25.0:      # mean block length ~25 bp -> fraction of sequence
                unit = int(rng.integers(1, 5))
                copies = int(rng.integers(10, 41))
                u = BASES[rng.integers(0, 4, size=unit)]
                blk = np.tile(u, copies)[: max(0, length - pos)]
                ref[pos:pos + len(blk)] = blk
                pos += len(blk)
            elif r < (str_fraction / 25.0) + lowcomplex_fraction / 60.0:
                two = BASES[rng.choice(4, size=2, replace=False)]
                ln = int(rng.integers(30, 90))
                blk = two[rng.integers(0, 2, size=ln)][: max(0, length - pos)]
                ref[pos:pos + len(blk)] = blk
                pos += len(blk)
            else:
                pos += 1
    return ref.tobytes().decode()


@dataclasses.dataclass
class PlantedVariant:
    pos: int      # 0-based reference position of first affected base
    ref: str      # reference allele ('' for pure insertion before pos)
    alt: str      # alternative allele ('' for pure deletion)
    somatic: bool


def plant_variants(ref: str, seed: int, somatic_every: int = 2000, germline_every: int = 1000,
                   margin: int = 300, dup_prob: float = 0.0) -> List[PlantedVariant]:
    rng = np.random.default_rng(seed)
    out: List[PlantedVariant] = []

    def one(p: int, somatic: bool) -> PlantedVariant:
        r = rng.random()
        if r < 0.40:   # SNV
            alts = [b for b in "ACGT" if b != ref[p]]
            return PlantedVariant(p, ref[p], alts[int(rng.integers(0, 3))], somatic)
        if r < 0.65:   # deletion 1-30
            ln = int(rng.integers(1, 31))
            return PlantedVariant(p, ref[p:p + ln], "", somatic)
        if r < 0.90:   # insertion 1-30
            ln = int(rng.integers(1, 31))
            ins = BASES[rng.integers(0, 4, size=ln)].tobytes().decode()
            if r < 0.65 + 0.25 * dup_prob:   # tandem duplication of the preceding 12+ln bases (graph cycles)
                ins = ref[p - (12 + ln):p]
            return PlantedVariant(p, "", ins, somatic)
        ln = int(rng.integers(2, 5))  # MNV / complex
        alt = "".join([b for b in "ACGT" if b != c][int(rng.integers(0, 3))] for c in ref[p:p + ln])
        return PlantedVariant(p, ref[p:p + ln], alt, somatic)

    n = len(ref)
    p = margin
    while p < n - margin:
        out.append(one(p + int(rng.integers(0, germline_every // 4)), False))
        p += germline_every
    p = margin + somatic_every // 2
    while p < n - margin:
        out.append(one(p + int(rng.integers(0, somatic_every // 4)), True))
        p += somatic_every
    out.sort(key=lambda v: v.pos)
    # drop overlapping ones
    keep: List[PlantedVariant] = []
    last_end = -100
    for v in out:
        if v.pos - last_end < 50:
            continue
        keep.append(v)
        last_end = v.pos + len(v.ref)
    return keep


def build_haplotype(ref: str, variants: List[PlantedVariant]) -> Tuple[np.ndarray, np.ndarray]:
    """Returns (hap bases uint8[], refpos int32[]) ; refpos[i] = 0-based ref coordinate of hap base i,
    or -1 for inserted bases."""
    rb = np.frombuffer(ref.encode(), dtype=np.uint8)
    seqs = []
    poss = []
    cur = 0
    for v in variants:
        if v.pos < cur:
            continue
        seqs.append(rb[cur:v.pos])
        poss.append(np.arange(cur, v.pos, dtype=np.int32))
        alt = np.frombuffer(v.alt.encode(), dtype=np.uint8)
        ap = np.full(len(alt), -1, dtype=np.int32)
        if len(v.ref) == len(v.alt):           # substitution block keeps coordinates
            ap = np.arange(v.pos, v.pos + len(alt), dtype=np.int32)
        seqs.append(alt)
        poss.append(ap)
        cur = v.pos + len(v.ref)
    seqs.append(rb[cur:])
    poss.append(np.arange(cur, len(rb), dtype=np.int32))
    return np.concatenate(seqs), np.concatenate(poss)


def _cigar_md(ref_b: np.ndarray, seq: np.ndarray, rpos: np.ndarray):
    """Builds (pos1, cigar, md, nm) for a read whose bases `seq` carry ref coordinates `rpos`
    (-1 = inserted).  Leading/trailing inserted bases are soft-clipped."""
    n = len(seq)
    lo = 0
    while lo < n and rpos[lo] < 0:
        lo += 1
    hi = n
    while hi > lo and rpos[hi - 1] < 0:
        hi -= 1
    if lo >= hi:
        return None
    ops = []
    if lo:
        ops.append((lo, "S"))
    md = []
    run = 0
    nm = 0
    mlen = 0
    prev = rpos[lo] - 1
    i = lo
    while i < hi:
        if rpos[i] < 0:
            j = i
            while j < hi and rpos[j] < 0:
                j += 1
            if mlen:
                ops.append((mlen, "M")); mlen = 0
            ops.append((j - i, "I"))
            nm += j - i
            i = j
            continue
        gap = rpos[i] - prev - 1
        if gap > 0:
            if mlen:
                ops.append((mlen, "M")); mlen = 0
            ops.append((gap, "D"))
            nm += gap
            md.append(str(run)); run = 0
            md.append("^" + ref_b[prev + 1:prev + 1 + gap].tobytes().decode())
        if seq[i] == ref_b[rpos[i]]:
            run += 1
        else:
            md.append(str(run)); run = 0
            md.append(chr(ref_b[rpos[i]]))
            nm += 1
        mlen += 1
        prev = rpos[i]
        i += 1
    if mlen:
        ops.append((mlen, "M"))
    if hi < n:
        ops.append((n - hi, "S"))
    md.append(str(run))
    # MD must alternate number / token; collapse "x" "^..." adjacency is already number-separated
    mdstr = "".join(md)
    cigar = "".join(f"{l}{o}" for l, o in ops)
    return int(rpos[lo]) + 1, cigar, mdstr, int(nm)


def simulate_sample(ref: str, rname: str, haps: List[Tuple[np.ndarray, np.ndarray]], hap_probs: List[float],
                    coverage: float, seed: int, prefix: str, rg: str, read_len: int = 150,
                    insert_mean: float = 400.0, insert_sd: float = 40.0, error_rate: float = 0.005,
                    region: Tuple[int, int] | None = None, linked: bool = False, monotone_starts: bool = False) -> List[Tuple[SamRead, SamRead]]:
    """Simulates fragments; returns list of (read1, read2) with read1.pos <= read2.pos not guaranteed.
    linked=True adds 10x-style tags from a separate random stream (the reads themselves do not change): BX:Z barcode
    shared by a few fragments (some fragments have none), HP:i haplotype 0 (unassigned) / 1 / 2."""
    rng = np.random.default_rng(seed)
    rng_lr = np.random.default_rng(seed + 7777)
    ref_b = np.frombuffer(ref.encode(), dtype=np.uint8)
    lo, hi = (0, len(ref)) if region is None else region
    span = hi - lo
    n_frag = int(coverage * span / (2.0 * read_len))
    qual_levels = np.array([37, 30, 25, 12], dtype=np.uint8)
    qual_p = np.array([0.70, 0.15, 0.08, 0.07])
    pairs = []
    # monotone_starts (tools/make_scan_bams.py with worker processes): the fragment start is looked up in the haplotype's reference
    # coordinates with the inserted bases (-1) filled forward, so that the binary search is over a sorted array.  Without it (the generator
    # of every fixture and of the bench workload: unchanged) a long insertion makes the search land on the same haplotype position for many
    # starts -- a handful of windows per megabase then hold thousands of reads (the "pile-ups" of the earlier rounds' scans).
    hsearch = [np.maximum.accumulate(hp_) for _, hp_ in haps] if monotone_starts else None
    for f in range(n_frag):
        h = int(rng.choice(len(haps), p=hap_probs))
        hb, hp = haps[h]
        ins = max(read_len, int(round(rng.normal(insert_mean, insert_sd))))
        # fragment start in haplotype coordinates, chosen through a reference coordinate inside region
        s_ref = int(rng.integers(lo, max(lo + 1, hi - ins)))
        s = int(np.searchsorted(hsearch[h] if monotone_starts else hp, np.int32(s_ref)))  # hp is non-decreasing except -1 runs; good enough (int32 like hp: a Python int makes numpy convert the whole array per call)
        while s < len(hp) and hp[s] < 0:
            s += 1
        e = s + ins
        if e > len(hb):
            continue
        first_fwd = bool(rng.integers(0, 2))
        reads = []
        for which, (a, b) in enumerate(((s, s + read_len), (e - read_len, e))):
            seq = hb[a:b].copy()
            rp = hp[a:b]
            q = qual_levels[rng.choice(4, size=read_len, p=qual_p)].copy()
            err = rng.random(read_len) < error_rate
            ne = int(err.sum())
            if ne:
                idx = np.nonzero(err)[0]
                for i in idx:
                    alts = [c for c in b"ACGT" if c != seq[i]]
                    seq[i] = alts[int(rng.integers(0, 3))]
                q[idx] = rng.integers(8, 21, size=ne).astype(np.uint8)
            cm = _cigar_md(ref_b, seq, rp)
            if cm is None:
                reads = []
                break
            pos1, cigar, md, nm = cm
            reverse = (which == 1)
            is_first = (which == 0) == first_fwd
            flag = 0x1 | 0x2 | (0x10 if reverse else 0x20) | (0x40 if is_first else 0x80)
            reads.append(SamRead(f"{prefix}{f:07d}", flag, rname, pos1, 60, cigar,
                                 seq.tobytes().decode(), (q + 33).tobytes().decode(),
                                 {"AS": read_len - 5 * nm, "XS": 0, "NM": nm, "MD": md, "RG": rg}))
        if len(reads) == 2:
            if linked:
                pool = max(4, n_frag // 3)
                code = int(rng_lr.integers(0, pool))
                bx = "".join("ACGT"[(code * 2654435761 >> (2 * j)) & 3] for j in range(16)) + "-1"
                has_bx = rng_lr.random() >= 0.1
                u = rng_lr.random()
                hp = 0 if u < 0.3 else (1 if h == 0 else 2)
                if rng_lr.random() < 0.03:
                    hp = 3 - hp if hp else 1              # phasing error
                for rd in reads:
                    if has_bx:
                        rd.tags["BX"] = bx
                    if rng_lr.random() >= 0.05:           # HP tag occasionally missing on one mate
                        rd.tags["HP"] = hp
            pairs.append((reads[0], reads[1]))
    return pairs


def pairs_to_sorted_reads(pairs) -> List[SamRead]:
    reads = []
    for a, b in pairs:
        reads.append(a)
        reads.append(b)
    reads.sort(key=lambda r: (r.pos, r.qname, r.flag))
    return reads


def write_sam(path: str, rname: str, rlen: int, sample: str, rg: str, pairs) -> None:
    lines = ["@HD\tVN:1.6\tSO:coordinate", f"@SQ\tSN:{rname}\tLN:{rlen}",
             f"@RG\tID:{rg}\tSM:{sample}\tPL:ILLUMINA"]
    recs = []
    for a, b in pairs:
        tl = max(_ref_end(a), _ref_end(b)) - min(a.pos, b.pos) + 1
        recs.append((a.pos, a.qname, a.flag, a.sam_line(b.pos, tl if a.pos <= b.pos else -tl)))
        recs.append((b.pos, b.qname, b.flag, b.sam_line(a.pos, tl if b.pos < a.pos else -tl)))
    recs.sort(key=lambda r: (r[0], r[1], r[2]))
    with open(path, "w") as f:
        f.write("\n".join(lines + [r[3] for r in recs]) + "\n")


def _ref_end(r: SamRead) -> int:
    """1-based inclusive reference end from CIGAR."""
    return r.pos + cigar_ref_len(r.cigar) - 1


def cigar_ref_len(cigar: str) -> int:
    n = 0
    num = 0
    for c in cigar:
        if c.isdigit():
            num = num * 10 + ord(c) - 48
        else:
            if c in "MDN=X":
                n += num
            num = 0
    return n


def write_fasta(path: str, name: str, seq: str, width: int = 60) -> None:
    with open(path, "w") as f:
        f.write(f">{name}\n")
        for i in range(0, len(seq), width):
            f.write(seq[i:i + width] + "\n")
    nlines_full = len(seq) // width
    with open(path + ".fai", "w") as f:
        f.write(f"{name}\t{len(seq)}\t{len(name) + 2}\t{width}\t{width + 1}\n")


def make_tumor_normal(ref_len: int = 20000, cov_t: float = 30, cov_n: float = 30, ref_seed: int = 22,
                      tumor_seed: int = 101, normal_seed: int = 202, rname: str = "chr22",
                      str_fraction: float = 0.0, lowcomplex_fraction: float = 0.0,
                      read_len: int = 150, error_rate: float = 0.005,
                      somatic_every: int = 2000, germline_every: int = 1000,
                      region: Tuple[int, int] | None = None, dup_prob: float = 0.0,
                      insert_mean: float = 400.0, insert_sd: float = 40.0, n_runs=(), linked: bool = False,
                      palindromes=()):
    """Returns dict(ref, variants, tumor_pairs, normal_pairs).
    palindromes: (position, half length) pairs -- the reference gets s + revcomp(s) there (k-mers that are their own
    reverse complement exist for even k only: CanonicalMer_t::set ties, reference src/Mer.hh:57-71)."""
    ref = random_reference(ref_len, ref_seed, str_fraction, lowcomplex_fraction)
    if palindromes:
        rl = list(ref)
        comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
        for pos, half in palindromes:
            left = rl[pos:pos + half]
            rl[pos + half:pos + 2 * half] = [comp[b] for b in reversed(left)]
        ref = "".join(rl)
    variants = plant_variants(ref, ref_seed + 1, somatic_every, germline_every, dup_prob=dup_prob)
    germ = [v for v in variants if not v.somatic]
    h0 = build_haplotype(ref, [])
    h1 = build_haplotype(ref, germ)
    h2 = build_haplotype(ref, variants)
    if n_runs:   # reference gaps: the FASTA shows N, the sequenced genome has real bases there
        rl = list(ref)
        for pos, ln in n_runs:
            rl[pos:pos + ln] = "N" * ln
        ref = "".join(rl)
    tumor = simulate_sample(ref, rname, [h0, h1, h2], [0.5, 0.25, 0.25], cov_t, tumor_seed, "T", "tumor",
                            read_len=read_len, error_rate=error_rate, region=region, insert_mean=insert_mean,
                            insert_sd=insert_sd, linked=linked)
    normal = simulate_sample(ref, rname, [h0, h1], [0.5, 0.5], cov_n, normal_seed, "N", "normal",
                             read_len=read_len, error_rate=error_rate, region=region, insert_mean=insert_mean,
                             insert_sd=insert_sd, linked=linked)
    return {"ref": ref, "rname": rname, "variants": variants, "tumor": tumor, "normal": normal}
