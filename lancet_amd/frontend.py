"""Host-side callers of the hot path, restated for SAM-level inputs.

* `tile_region`     -- window tiler, follows loadRefs (reference src/Lancet.cc:189-316):
                       padding, clipping, 600 bp windows with stride 100, last window LEN = len-offset-1,
                       uppercase + IUPAC->N, windows keyed/ordered by the "chr:start-end" string
                       (std::map order, src/Microassembler.hh:148 / src/Microassembler.cc:779).
* `extract_reads`   -- per-window read selection, follows Microassembler::extractReads
                       (reference src/Microassembler.cc:436-655) on already-decoded SAM fields.
* `build_batch`     -- packs the selected reads of many windows into the SoA `WindowBatch` that crosses the
                       C-ABI (include/lancet_engine.h : lancet_window_batch).

BAM/BGZF decoding itself is out of scope for this round; inputs here are `SamRead` records.
"""
from __future__ import annotations

import dataclasses
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .synth import SamRead, cigar_ref_len

TMR = 4   # reference src/Ref.hh:36
NML = 5   # reference src/Ref.hh:37
FWD = 1   # reference src/ReadInfo.hh:30
REV = 2

_AMBIG = set("MRWSYKVHDBXmrwsykvhdbx")  # reference src/util.cc:167-187 isAmbiguos


@dataclasses.dataclass
class Window:
    hdr: str
    chrom: str
    start: int      # Ref_t::refstart (1-based)
    end: int        # Ref_t::refend = start + LEN
    seq: str


def tile_region(contig_seq: str, chrom: str, region: str, contig_len: Optional[int] = None,
                padding: int = 250, window_size: int = 600, delta: int = 100) -> List[Window]:
    """reference src/Lancet.cc:189-316 (loadRefs).  `contig_seq` is the full contig (1-based coords)."""
    contig_len = len(contig_seq) if contig_len is None else contig_len
    x = region.find(":")
    if x < 0:
        sp, ep = 1, contig_len
    else:
        y = region.find("-", x)
        sp = int(region[x + 1:y]) - padding
        ep = int(region[y + 1:]) + padding
        if sp < 1:
            sp = 1
        if ep > contig_len:
            ep = contig_len
    s = contig_seq[sp - 1:ep].upper()
    s = "".join("N" if c in _AMBIG else c for c in s)
    out: List[Window] = []
    end = len(s)
    offset = 0
    while offset < end:
        ln = window_size
        if offset + window_size >= len(s):
            ln = len(s) - offset - 1
            end = offset
        ss = s[offset:offset + ln] if ln > 0 else ""
        start = sp + offset
        out.append(Window(f"{chrom}:{start}-{start + ln}", chrom, start, start + ln, ss))
        offset += delta
    return out


def windows_in_processing_order(windows: Sequence[Window]) -> List[Window]:
    """std::map<string, Ref_t*> iteration order (reference src/Microassembler.cc:779)."""
    return sorted(windows, key=lambda w: w.hdr.encode())


@dataclasses.dataclass
class ReadFilterParams:
    min_map_qual: int = 15          # reference src/Lancet.hh:49
    max_delta_as_xs: int = 5        # reference src/Lancet.hh:50
    primary_alignment_only: bool = False
    xa_filter: bool = False
    max_avg_cov: int = 10000


def extract_reads(reads: Sequence[SamRead], starts: np.ndarray, win: Window, code: int,
                  p: ReadFilterParams) -> Tuple[List[Tuple[SamRead, int, int, bool]], bool]:
    """reference src/Microassembler.cc:436-655.  `reads` must be in BAM (coordinate) order and
    `starts` = their 0-based positions (for the region seek).  Returns ([(read, mate, strand, mapped)], skip)."""
    out = []
    mq = p.min_map_qual
    min_delta = p.max_delta_as_xs
    if code == NML:
        mq = 0
        min_delta = -1
    totalbp = 0
    rawlen = len(win.seq)
    # BamTools region = overlap semantics; anything that can pass the containment test starts >= refstart
    lo = int(np.searchsorted(starts, win.start, side="left"))
    for i in range(lo, len(reads)):
        r = reads[i]
        alstart = r.pos - 1
        if alstart > win.end:
            break
        if rawlen > 0 and (totalbp / rawlen) > p.max_avg_cov:       # :491-496
            return out, True
        alend = alstart + cigar_ref_len(r.cigar)                     # GetEndPosition: 0-based half-open
        if alstart < win.start or alend > win.end:                   # :498-500 (1-based vs 0-based, kept)
            continue
        if p.primary_alignment_only and (r.flag & 0x100):
            continue
        if not (r.mapq >= mq and not (r.flag & 0x400)):              # :504
            continue
        mate = 1 if (r.flag & 0x40) else (2 if (r.flag & 0x80) else 0)
        if (r.flag & 0x40) and (r.flag & 0x80):
            mate = 2                                                 # :510-511 second assignment wins
        strand = REV if (r.flag & 0x10) else FWD
        a_s = float(r.tags["AS"]) if "AS" in r.tags else -1.0
        x_s = float(r.tags["XS"]) if "XS" in r.tags else -1.0
        if abs(a_s - x_s) <= min_delta and a_s != -1 and x_s != -1:   # :535
            continue
        xt = r.tags.get("XT", "")
        if xt == "R" and code != NML:                                # :554-559
            continue
        xa = r.tags.get("XA", "")
        if xa != "" and code != NML and p.xa_filter:                 # :573-579
            continue
        mapped = not (r.flag & 0x4)
        out.append((r, mate, strand, mapped))
        totalbp += len(r.seq)
    return out, False


_MD_VALID = set("acgtumrwsykvhdbxnACGTUMRWSYKVHDBXN^")


def _parse_md(md: str, M: Dict[int, int], start: int, qual: str, min_qv: int) -> None:
    """reference src/util.cc:428-483 (parseMD), including its quirks: the quality looked at is the one of the base AFTER
    the mismatch in MD coordinates (rpos is incremented first; std::string::operator[] at size() is '\\0')."""
    n = len(md)
    p = next((i for i in range(n) if md[i] in _MD_VALID), -1)
    p_old = -1
    pos, rpos = start, 0
    while p != -1:
        num = md[p_old + 1:p]
        step = _atoi(num)
        pos += step; rpos += step
        if md[p] == "^":
            p2 = next((i for i in range(p + 1, n) if md[i] not in _MD_VALID), -1)
            dele = md[p + 1:p2] if p2 != -1 else md[p + 1:]
            pos += len(dele)
            if p2 == -1:           # find_first_of(valid, npos) == npos ; p_old = npos - 1
                p = -1
                p_old = n          # md.substr(npos) would throw in the reference; not reachable with a valid MD
                break
            p = next((i for i in range(p2, n) if md[i] in _MD_VALID), -1)
            p_old = p2 - 1
        else:
            pos += 1; rpos += 1
            q = ord(qual[rpos]) if rpos < len(qual) else 0
            if q >= min_qv:
                M[pos] = M.get(pos, 0) + 1
            p_old = p
            p = next((i for i in range(p_old + 1, n) if md[i] in _MD_VALID), -1)


def _atoi(s: str) -> int:
    i, n, v = 0, len(s), 0
    while i < n and s[i] in " \t\n\r\f\v":
        i += 1
    neg = False
    if i < n and s[i] in "+-":
        neg = s[i] == "-"; i += 1
    while i < n and s[i].isdigit():
        v = v * 10 + ord(s[i]) - 48; i += 1
    return -v if neg else v


_CIGAR_RE = None


def _cigar_ops(cigar: str):
    global _CIGAR_RE
    if _CIGAR_RE is None:
        import re
        _CIGAR_RE = re.compile(r"(\d+)([MIDNSHP=X])")
    return [(int(n), t) for n, t in _CIGAR_RE.findall(cigar)]


def is_active_region(reads: Sequence[SamRead], starts: np.ndarray, win: Window, code: int, p: "ReadFilterParams",
                     min_evidence: int = 3, min_qual_call: int = 17 + 33) -> bool:
    """reference src/Microassembler.cc:253-432 (isActiveRegion): any locus with >= MIN_EVIDENCE (= filters.minAltCntTumor)
    mismatches (MD, quality filtered), insertions, deletions or soft-clip starts among the reads contained in the
    window (MQ >= MIN_MAP_QUAL in the tumor, every MQ in the normal; duplicates skipped)."""
    mq = 0 if code == NML else p.min_map_qual
    mapX: Dict[int, int] = {}
    mapI: Dict[int, int] = {}
    mapD: Dict[int, int] = {}
    mapSC: Dict[int, int] = {}
    lo = int(np.searchsorted(starts, win.start, side="left"))
    for i in range(lo, len(reads)):
        r = reads[i]
        alstart = r.pos - 1
        if alstart > win.end:
            break
        ops = _cigar_ops(r.cigar)
        alend = alstart + sum(n for n, t in ops if t in "MDN=X")
        if alstart < win.start or alend > win.end:
            continue
        if not (r.mapq >= mq and not (r.flag & 0x400)):
            continue
        if not r.seq or not r.qual or r.seq == "*" or r.qual == "*":
            continue
        md = r.tags.get("MD")
        if md is not None:
            _parse_md(str(md), mapX, alstart, r.qual, min_qual_call)
        pos = alstart
        for n, t in ops:
            if t != "I":
                pos += n
            if t == "X":
                mapX[pos] = mapX.get(pos, 0) + 1
            if t == "I":
                mapI[pos] = mapI.get(pos, 0) + 1
            if t == "D":
                mapD[pos] = mapD.get(pos, 0) + 1
        # BamAlignment::GetSoftClips (bamtools 2.5.2, src/api/BamAlignment.cpp:536-606): genome position of every S op
        refpos = alstart
        for n, t in ops:
            if t in "DMXN=":
                refpos += n
            elif t == "S":
                mapSC[refpos] = mapSC.get(refpos, 0) + 1
    return any(any(v >= min_evidence for v in m.values()) for m in (mapX, mapI, mapD, mapSC))


@dataclasses.dataclass
class WindowBatch:
    """SoA batch that crosses the C-ABI (include/lancet_engine.h: lancet_window_batch)."""
    n_windows: int
    hdr: List[str]
    chrom: List[str]
    chr_id: np.ndarray        # int32[n]
    ref_start: np.ndarray     # int32[n]
    ref_off: np.ndarray       # uint32[n+1]
    ref_bases: np.ndarray     # uint8[]
    read_begin: np.ndarray    # uint32[n+1]
    seq_off: np.ndarray       # uint32[R+1]
    seq: np.ndarray           # uint8[]
    qual: np.ndarray          # uint8[]
    label: np.ndarray         # uint8[R]
    strand: np.ndarray        # uint8[R]
    mate: np.ndarray          # uint8[R]
    mapped: np.ndarray        # uint8[R]
    name_rank: np.ndarray     # uint32[R]
    # --linked-reads only (None otherwise): ReadInfo_t::BX as the dense rank of the barcode among the batch's barcodes
    # under std::string operator< (NO_BX = "null"), ReadInfo_t::HP (0 | 1 | 2), and the barcode strings by rank
    bx_rank: Optional[np.ndarray] = None    # uint32[R]
    hp: Optional[np.ndarray] = None         # uint8[R]
    bx_names: Optional[List[str]] = None

    @property
    def n_reads(self) -> int:
        return int(self.read_begin[-1])


NO_BX = 0xFFFFFFFF


def build_batch(windows: Sequence[Window], per_window_reads: Sequence[Sequence[Tuple[str, str, str, int, int, int, bool]]],
                chrom_ids: Optional[Dict[str, int]] = None, linked: bool = False) -> WindowBatch:
    """per_window_reads[w] = [(name, seq, qual, label, strand, mate, mapped)], tumor reads then normal reads
    (the order Graph_t::readid2info gets filled, reference src/Microassembler.cc:833-834)."""
    n = len(windows)
    chrom_ids = chrom_ids if chrom_ids is not None else {}
    ref_off = np.zeros(n + 1, dtype=np.uint32)
    read_begin = np.zeros(n + 1, dtype=np.uint32)
    refs = []
    seqs: List[bytes] = []
    quals: List[bytes] = []
    seq_len: List[int] = []
    label: List[int] = []
    strand: List[int] = []
    mate: List[int] = []
    mapped: List[int] = []
    ranks: List[int] = []
    chr_id = np.zeros(n, dtype=np.int32)
    ref_start = np.zeros(n, dtype=np.int32)
    bxs: List[str] = []
    hps: List[int] = []
    for w, win in enumerate(windows):
        refs.append(win.seq.encode())
        ref_off[w + 1] = ref_off[w] + len(win.seq)
        chr_id[w] = chrom_ids.setdefault(win.chrom, len(chrom_ids))
        ref_start[w] = win.start
        rs = per_window_reads[w]
        names = sorted({r[0].encode() for r in rs})
        rank = {nm: i for i, nm in enumerate(names)}
        for rec in rs:
            (name, s, q, lab, st, mt, mp) = rec[:7]
            if linked:
                bxs.append(rec[7]); hps.append(rec[8])
            seqs.append(s.encode())
            quals.append(q.encode())
            seq_len.append(len(s))
            label.append(lab)
            strand.append(st)
            mate.append(mt)
            mapped.append(1 if mp else 0)
            ranks.append(rank[name.encode()])
        read_begin[w + 1] = read_begin[w] + len(rs)
    seq_off = np.zeros(len(seq_len) + 1, dtype=np.uint32)
    if seq_len:
        seq_off[1:] = np.cumsum(np.asarray(seq_len, dtype=np.uint64)).astype(np.uint32)
    lr = {}
    if linked:
        names = sorted({b.encode() for b in bxs if b != "null"})
        rk = {nm: i for i, nm in enumerate(names)}
        lr = dict(bx_rank=np.asarray([rk[b.encode()] if b != "null" else NO_BX for b in bxs], dtype=np.uint32),
                  hp=np.asarray(hps, dtype=np.uint8), bx_names=[nm.decode() for nm in names])
    return WindowBatch(
        **lr,
        n_windows=n, hdr=[w.hdr for w in windows], chrom=[w.chrom for w in windows],
        chr_id=chr_id, ref_start=ref_start, ref_off=ref_off,
        ref_bases=np.frombuffer(b"".join(refs), dtype=np.uint8).copy(),
        read_begin=read_begin, seq_off=seq_off,
        seq=np.frombuffer(b"".join(seqs), dtype=np.uint8).copy(),
        qual=np.frombuffer(b"".join(quals), dtype=np.uint8).copy(),
        label=np.asarray(label, dtype=np.uint8), strand=np.asarray(strand, dtype=np.uint8),
        mate=np.asarray(mate, dtype=np.uint8), mapped=np.asarray(mapped, dtype=np.uint8),
        name_rank=np.asarray(ranks, dtype=np.uint32))


def batch_from_sam(windows: Sequence[Window], tumor: Sequence[SamRead], normal: Sequence[SamRead],
                   p: Optional[ReadFilterParams] = None, max_k: int = 101, linked: bool = False,
                   active_region: bool = False, min_evidence: int = 3, min_qual_call: int = 17 + 33,
                   leak: Optional[list] = None):
    """Runs the per-window part of processReads (reference src/Microassembler.cc:779-842) up to the
    processGraph call: returns (batch, kept_windows) for windows that are not skipped.
    Active-region prefilter is not applied here (== --active-region-off)."""
    p = p or ReadFilterParams()
    # several contigs (--bed): tumor / normal may be {contig: reads}; the windows of all contigs share one processing order
    by_chrom = isinstance(tumor, dict)
    tumor_by = tumor if by_chrom else None
    normal_by = normal if by_chrom else None
    starts_of = lambda reads: np.asarray([r.pos - 1 for r in reads], dtype=np.int64)
    if by_chrom:
        t_starts_by = {c: starts_of(v) for c, v in tumor_by.items()}
        n_starts_by = {c: starts_of(v) for c, v in normal_by.items()}
    else:
        t_starts = starts_of(tumor)
        n_starts = starts_of(normal)
    kept: List[Window] = []
    per: List[list] = []
    # `leak`: reads of a window without a mapped read.  processGraph returns there before g.clear() (reference
    # src/Microassembler.cc:83), so they are still in the graph when the next window is loaded; pass the same list object
    # again to carry them across calls (chunks of one scan).
    leak = leak if leak is not None else []
    for win in windows_in_processing_order(windows):
        if by_chrom:
            tumor, normal = tumor_by.get(win.chrom, []), normal_by.get(win.chrom, [])
            t_starts, n_starts = t_starts_by.get(win.chrom, starts_of([])), n_starts_by.get(win.chrom, starts_of([]))
        if not win.seq:                 # isNseq (:799, src/util.cc:259-273): `!= 'N' || != 'n'` holds for every character,
            continue                    # so only an EMPTY window counts as all-N (long all-N windows fall to isRepeat)
        if _is_repeat(win.seq, max_k):                                  # :800
            continue
        if active_region and not (is_active_region(tumor, t_starts, win, TMR, p, min_evidence, min_qual_call)        # :817-820
                                  or is_active_region(normal, n_starts, win, NML, p, min_evidence, min_qual_call)):
            continue
        tr, skip_t = extract_reads(tumor, t_starts, win, TMR, p)
        nr, skip_n = extract_reads(normal, n_starts, win, NML, p)
        if skip_t or skip_n:
            leak.clear()                                                # g.clear(true), :841
            continue
        # BX / HP as extractReads reads them (reference src/Microassembler.cc:581-593): missing BX -> "null", missing HP -> 0
        lr = (lambda r: (r.tags.get("BX", "") or "null", max(0, int(r.tags.get("HP", 0))))) if linked else (lambda r: ())
        rs = list(leak)
        rs += [(r.qname, r.seq, r.qual, TMR, st, mt, mp) + lr(r) for (r, mt, st, mp) in tr]
        rs += [(r.qname, r.seq, r.qual, NML, st, mt, mp) + lr(r) for (r, mt, st, mp) in nr]
        kept.append(win)
        per.append(rs)
        if any(rec[6] for rec in rs):
            leak.clear()
        else:
            leak[:] = rs                                                # countMappedReads() <= 0: nothing is cleared
    return build_batch(kept, per, linked=linked), kept


def _is_repeat(seq: str, k: int) -> bool:
    """reference src/util.cc:295-315 (offsets [0, len-K))."""
    seen = set()
    for off in range(0, len(seq) - k):
        s = seq[off:off + k]
        if s in seen:
            return True
        seen.add(s)
    return False
