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


def write_bam(path: str, refs, reads, sample: str = "S", rg: str = "rg1") -> None:
    """refs = [(name, length)]; reads = SamRead list in file order."""
    text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join(f"@SQ\tSN:{n}\tLN:{l}\n" for n, l in refs) + f"@RG\tID:{rg}\tSM:{sample}\n"
    body = bytearray(b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(refs)))
    for n, l in refs:
        body += struct.pack("<i", len(n) + 1) + n.encode() + b"\0" + struct.pack("<i", l)
    rid = {n: i for i, (n, _) in enumerate(refs)}
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
        body += struct.pack("<i", len(rec)) + rec
    with open(path, "wb") as fh:
        for i in range(0, len(body), 60000):
            fh.write(_bgzf_block(bytes(body[i:i + 60000])))
        fh.write(_bgzf_block(b""))
