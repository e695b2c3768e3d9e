"""Minimal BGZF / BAM / FASTA readers for the host side of the scan.

The reference reads its inputs through bamtools 2.5.2 (BamReader + .bai jump) and htslib's faidx
(reference src/Microassembler.cc:436-655, src/Lancet.cc:189-316).  Neither library is part of the hot path nor
of /root/reference; this module restates the published file formats (SAM/BAM specification v1, sections 4.1-4.2)
far enough to feed `frontend.batch_from_sam`: sequential decode of a coordinate-sorted BAM into `SamRead`
records (the region "jump" becomes a binary search over the decoded start positions, which selects the same
alignments as bamtools' overlap region), and whole-file FASTA load.  Pure host code, no GPU involvement."""
from __future__ import annotations

import struct
import zlib
from typing import Dict, List, Tuple

from .synth import SamRead

_SEQ = "=ACMGRSVTWYHKDBN"
_CIG = "MIDNSHP=X"
_SEQ2 = [a + b for a in _SEQ for b in _SEQ]


def bgzf_decompress(raw: bytes) -> bytes:
    """A BGZF file is a series of gzip members (each <= 64 KiB of payload); the empty EOF block is harmless."""
    out = []
    data = raw
    while data:
        d = zlib.decompressobj(31)
        out.append(d.decompress(data))
        if not d.eof:
            raise ValueError("truncated BGZF block")
        data = d.unused_data
    return b"".join(out)


def _tags(buf: bytes, p: int, end: int) -> dict:
    tags = {}
    while p < end:
        key = buf[p:p + 2].decode()
        t = chr(buf[p + 2])
        p += 3
        if t == "A":
            # What BamAlignment::GetTag(tag, std::string&) of bamtools 2.5.2 hands the reference for a one-character tag
            # (src/api/BamAlignment.cpp: strlen over the raw tag block): the character AND every byte after it up to the
            # next NUL -- i.e. the bare character only when the tag is the record's last one.  extractReads compares that
            # string with "R" (XT:A:R, src/Microassembler.cc:549-559), so the repeat filter fires only then.
            q = buf.find(b"\0", p, end)
            tags[key] = buf[p:(q if q >= 0 else end)].decode("latin-1"); p += 1
        elif t in "cCsSiI":
            fmt, n = {"c": ("<b", 1), "C": ("<B", 1), "s": ("<h", 2), "S": ("<H", 2), "i": ("<i", 4), "I": ("<I", 4)}[t]
            tags[key] = struct.unpack_from(fmt, buf, p)[0]; p += n
        elif t == "f":
            tags[key] = struct.unpack_from("<f", buf, p)[0]; p += 4
        elif t in "ZH":
            q = buf.index(b"\0", p)
            tags[key] = buf[p:q].decode(); p = q + 1
        elif t == "B":
            st = chr(buf[p]); n = struct.unpack_from("<i", buf, p + 1)[0]
            sz = {"c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4}[st]
            tags[key] = list(struct.unpack_from("<" + str(n) + {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[st], buf, p + 5))
            p += 5 + n * sz
        else:
            raise ValueError(f"unknown BAM tag type {t!r}")
    return tags


def read_bam(path: str) -> Tuple[dict, List[SamRead]]:
    """Returns (header, reads).  header = {"text": str, "refs": [(name, length)], "samples": [SM of the @RG lines]}.
    reads keep file order (coordinate order for the inputs the reference accepts); unplaced reads have pos 0."""
    buf = bgzf_decompress(open(path, "rb").read())
    if buf[:4] != b"BAM\1":
        raise ValueError(f"{path}: not a BAM file")
    l_text = struct.unpack_from("<i", buf, 4)[0]
    text = buf[8:8 + l_text].split(b"\0")[0].decode()
    p = 8 + l_text
    n_ref = struct.unpack_from("<i", buf, p)[0]; p += 4
    refs = []
    for _ in range(n_ref):
        ln = struct.unpack_from("<i", buf, p)[0]; p += 4
        name = buf[p:p + ln - 1].decode(); p += ln
        refs.append((name, struct.unpack_from("<i", buf, p)[0])); p += 4
    samples = []
    for line in text.splitlines():
        if line.startswith("@RG"):
            for f in line.split("\t")[1:]:
                if f.startswith("SM:"):
                    samples.append(f[3:])
    reads: List[SamRead] = []
    n = len(buf)
    while p + 4 <= n:
        bs = struct.unpack_from("<i", buf, p)[0]; p += 4
        end = p + bs
        ref_id, pos, l_name, mapq, _bin, n_cig, flag, l_seq, _nref, _npos, _tlen = struct.unpack_from("<iiBBHHHiiii", buf, p)
        q = p + 32
        qname = buf[q:q + l_name - 1].decode(); q += l_name
        cig = struct.unpack_from("<" + str(n_cig) + "I", buf, q); q += 4 * n_cig
        cigar = "".join(f"{c >> 4}{_CIG[c & 15]}" for c in cig) or "*"
        nb = (l_seq + 1) // 2
        sb = buf[q:q + nb]; q += nb
        seq = "".join(_SEQ2[b] for b in sb)[:l_seq]
        qb = buf[q:q + l_seq]; q += l_seq
        # unstored qualities (0xFF): bamtools fills (char)0xFF, a negative char below every threshold; NUL bytes here
        qual = "\0" * l_seq if (l_seq and qb[0] == 0xFF) else bytes(c + 33 for c in qb).decode()
        tags = _tags(buf, q, end)
        reads.append(SamRead(qname, flag, refs[ref_id][0] if ref_id >= 0 else "*", pos + 1, mapq, cigar, seq, qual, tags))
        p = end
    return {"text": text, "refs": refs, "samples": samples}, reads


def read_fasta(path: str) -> Dict[str, str]:
    """name (first word of the header) -> sequence, as stored (case and ambiguity codes are handled by tile_region)."""
    seqs: Dict[str, List[str]] = {}
    cur = None
    with open(path) as fh:
        for line in fh:
            line = line.rstrip("\n\r")
            if line.startswith(">"):
                cur = line[1:].split()[0] if len(line) > 1 else ""
                seqs[cur] = []
            elif cur is not None:
                seqs[cur].append(line)
    return {k: "".join(v) for k, v in seqs.items()}
