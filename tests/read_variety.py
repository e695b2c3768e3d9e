"""Tag / flag / CIGAR variety for synthetic reads: everything the read filters of the reference look at
(src/Microassembler.cc:498-579 extractReads, :253-432 isActiveRegion).  Shared by the randomized native-host tests and
by tools/make_filter_golden.py (which runs the reference on such reads)."""
import numpy as np

from lancet_amd import synth


def decorate(reads, rng, linked=False):
    out = []
    for r in reads:
        tags = dict(r.tags)
        u = rng.random()
        flag, mapq, cigar, pos = r.flag, r.mapq, r.cigar, r.pos
        if u < 0.04:
            tags["XT"] = ("A", "R")
        elif u < 0.08:
            tags["XT"] = ("A", "U")
        elif u < 0.12:
            tags["XA"] = "chr1,+100,100M,1;"
        elif u < 0.16:
            tags["XS"] = int(tags.get("AS", 90)) - int(rng.integers(0, 8))
        elif u < 0.19:
            flag |= 0x400
        elif u < 0.22:
            flag |= 0x100
        elif u < 0.26:
            mapq = int(rng.integers(0, 20))
        elif u < 0.30 and cigar.endswith("M") and cigar[:-1].isdigit() and int(cigar[:-1]) > 20:     # soft clip the head or the tail
            n, k = int(cigar[:-1]), int(rng.integers(3, 12))
            if rng.random() < 0.5:
                cigar, pos = f"{k}S{n - k}M", pos + k
            else:
                cigar = f"{n - k}M{k}S"
            tags.pop("MD", None)
        elif u < 0.32:
            tags.pop("AS", None)
        elif u < 0.33:
            flag |= 0x4
        if linked and rng.random() < 0.9:
            tags["BX"] = "ACGT"[int(rng.integers(0, 4))] * 4 + f"{int(rng.integers(0, 40)):04d}-1"
            if rng.random() < 0.8:
                tags["HP"] = int(rng.integers(1, 3))
        out.append(synth.SamRead(r.qname, flag, r.rname, pos, mapq, cigar, r.seq, r.qual, tags))
    return sorted(out, key=lambda x: x.pos)


def sam_line(r) -> str:
    t = []
    for k, v in r.tags.items():
        if isinstance(v, tuple):
            t.append(f"{k}:{v[0]}:{v[1]}")
        elif isinstance(v, (int, np.integer)):
            t.append(f"{k}:i:{int(v)}")
        else:
            t.append(f"{k}:Z:{v}")
    return "\t".join([r.qname, str(r.flag), r.rname, str(r.pos), str(r.mapq), r.cigar, "=", str(r.pos), "0", r.seq, r.qual] + t)
