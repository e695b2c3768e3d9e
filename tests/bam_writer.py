"""Test-side BAM writer (SAM specification v1, sections 4.1-4.2): SamRead records -> BGZF-compressed BAM, so that the
native BAM reader can be exercised on randomized inputs.  Test infrastructure only."""
import struct
import zlib

_CIG = "MIDNSHP=X"
_SEQ = "=ACMGRSVTWYHKDBN"


def _bgzf_block(data: bytes) -> bytes:
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = co.compress(data) + co.flush()
    bsize = len(comp) + 25
    return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", bsize) + comp
            + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))


def _cigar(cigar: str):
    out, num = [], 0
    for c in cigar:
        if c.isdigit():
            num = num * 10 + ord(c) - 48
        else:
            out.append((num << 4) | _CIG.index(c)); num = 0
    return out


def _tag(key: str, val) -> bytes:
    k = key.encode()
    if isinstance(val, tuple):                       # ("A", "R") : explicit type
        t, v = val
        if t == "A":
            return k + b"A" + v.encode()
        raise ValueError(t)
    if isinstance(val, int):
        if -128 <= val < 128:
            return k + b"c" + struct.pack("<b", val)
        if 0 <= val < 65536:
            return k + b"S" + struct.pack("<H", val)
        return k + b"i" + struct.pack("<i", val)
    if isinstance(val, float):
        return k + b"f" + struct.pack("<f", val)
    return k + b"Z" + str(val).encode() + b"\0"


def _reg2bin(beg: int, end: int) -> int:
    """SAM spec 5.3"""
    end -= 1
    if beg >> 14 == end >> 14:
        return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17:
        return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20:
        return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23:
        return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26:
        return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def write_bam(path: str, refs, reads, sample: str = "S", rg: str = "rg1", index: bool = False, linear: bool = True, block: int = 60000) -> None:
    """refs = [(name, length)]; reads = SamRead list in file order.  index=True also writes <path>.bai (SAM spec 5.2: bins with
    their chunks and, unless linear=False -- what `bamtools index` leaves out --, the 16 kb linear index)."""
    text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join(f"@SQ\tSN:{n}\tLN:{l}\n" for n, l in refs) + f"@RG\tID:{rg}\tSM:{sample}\n"
    body = bytearray(b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(refs)))
    for n, l in refs:
        body += struct.pack("<i", len(n) + 1) + n.encode() + b"\0" + struct.pack("<i", l)
    rid = {n: i for i, (n, _) in enumerate(refs)}
    spans = []          # (ref id, start, end, offsets of the record in the inflated stream)
    for r in reads:
        cig = [] if r.cigar == "*" else _cigar(r.cigar)
        seq = "" if r.seq == "*" else r.seq
        nib = [_SEQ.index(c) if c in _SEQ else 15 for c in seq]
        if len(nib) % 2:
            nib.append(0)
        sb = bytes((nib[i] << 4) | nib[i + 1] for i in range(0, len(nib), 2))
        qb = bytes([0xFF] * len(seq)) if r.qual == "*" else bytes(ord(c) - 33 for c in r.qual)
        tags = b"".join(_tag(k, v) for k, v in r.tags.items())
        name = r.qname.encode() + b"\0"
        rec = struct.pack("<iiBBHHHiiii", rid.get(r.rname, -1), r.pos - 1, len(name), r.mapq, 4680, len(cig), r.flag, len(seq), -1, -1, 0)
        rec += name + struct.pack("<" + str(len(cig)) + "I", *cig) + sb + qb + tags
        ref_len = sum(c >> 4 for c in cig if (c & 15) in (0, 2, 3, 7, 8)) or 1
        spans.append((rid.get(r.rname, -1), r.pos - 1, r.pos - 1 + ref_len, len(body), len(body) + 4 + len(rec)))
        body += struct.pack("<i", len(rec)) + rec
    coff = []
    with open(path, "wb") as fh:
        for i in range(0, len(body), block):
            coff.append(fh.tell())
            fh.write(_bgzf_block(bytes(body[i:i + block])))
        coff.append(fh.tell())
        fh.write(_bgzf_block(b""))
    if not index:
        return
    voff = lambda u: (coff[u // block] << 16) | (u % block)
    out = bytearray(b"BAI\1" + struct.pack("<i", len(refs)))
    for t in range(len(refs)):
        bins, lin = {}, {}
        for (tid, beg, end, u0, u1) in spans:
            if tid != t:
                continue
            b = _reg2bin(beg, end)
            ch = bins.setdefault(b, [])
            if ch and ch[-1][1] == voff(u0):
                ch[-1][1] = voff(u1)
            else:
                ch.append([voff(u0), voff(u1)])
            for w in range(beg >> 14, ((end - 1) >> 14) + 1):
                lin[w] = min(lin.get(w, voff(u0)), voff(u0))
        out += struct.pack("<i", len(bins))
        for b in sorted(bins):
            out += struct.pack("<Ii", b, len(bins[b]))
            for c0, c1 in bins[b]:
                out += struct.pack("<QQ", c0, c1)
        if linear and lin:
            n = max(lin) + 1
            out += struct.pack("<i", n)
            prev = 0
            for w in range(n):
                prev = lin.get(w, prev)          # (samtools fills the gaps with the preceding offset)
                out += struct.pack("<Q", prev)
        else:
            out += struct.pack("<i", 0)
    with open(path + ".bai", "wb") as fh:
        fh.write(bytes(out))
