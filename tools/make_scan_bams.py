"""Synthetic tumor/normal BAM + FASTA for end-to-end runs of the command-line programs (test-side BAM writer).

    python tools/make_scan_bams.py OUTDIR [ref_len=500000] [cov_t=30] [cov_n=30] [procs=1] [contigs=1]

contigs > 1 (with procs > 1): that many contigs of ref_len bases each (chr1 .. chr22, chrX, chrY; every contig its own reference, variants and
read seeds), one BAM pair + .bai over all of them, a multi-contig FASTA + .fai and OUTDIR/regions.bed (the whole genome minus 1 kb at the
contig ends): the one-GPU proxy of BASELINE config 3.

procs > 1: the contig is dealt out in stretches to worker processes (same reference, same planted variants; every stretch its own read
seed; a fragment starts inside its stretch and may end in the next one, so the coverage has no seams), which simulate AND encode their
reads; the parent sorts the records by (position, name, flag) and writes the BAM + .bai.  Fragment starts are looked up in a sorted copy of
the haplotype's coordinates (synth.simulate_sample monotone_starts): uniform coverage.  procs = 1 is the generator of the earlier rounds
(build/scan500k was made with it): its binary search over coordinates with inserted bases piles thousands of fragments onto a few loci per
megabase -- the coverage pile-ups of the earlier rounds' scans, kept there as they are a useful stress."""
import multiprocessing as mp
import os, struct, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bam_writer
from lancet_amd import synth

_G = {}


def _encode(r, rid):
    cig = [] if r.cigar == "*" else bam_writer._cigar(r.cigar)
    seq = r.seq
    nib = [bam_writer._SEQ.index(c) if c in bam_writer._SEQ else 15 for c in seq]
    if len(nib) % 2:
        nib.append(0)
    sb = bytes((nib[i] << 4) | nib[i + 1] for i in range(0, len(nib), 2))
    qb = bytes(ord(c) - 33 for c in r.qual)
    tags = b"".join(bam_writer._tag(k, v) for k, v in r.tags.items())
    name = r.qname.encode() + b"\0"
    rec = struct.pack("<iiBBHHHiiii", rid, r.pos - 1, len(name), r.mapq, 4680, len(cig), r.flag, len(seq), -1, -1, 0)
    rec += name + struct.pack("<" + str(len(cig)) + "I", *cig) + sb + qb + tags
    ref_len = sum(c >> 4 for c in cig if (c & 15) in (0, 2, 3, 7, 8)) or 1
    return rec, ref_len


def _stretch(job):
    which, i, lo, hi = job
    g = _G
    haps, probs, cov, seed, prefix, rg = (g["haps_t"], [0.5, 0.25, 0.25], g["cov_t"], 101, "T", "tumor") if which == "T" else (g["haps_n"], [0.5, 0.5], g["cov_n"], 202, "N", "normal")
    tid = g.get("tid", 0)
    tag = f"{prefix}{i:02d}x" if not g.get("multi") else f"{prefix}{tid:02d}_{i:02d}x"
    pairs = synth.simulate_sample(g["ref"], g["rname"], haps, probs, cov, seed + 1000 * i + 100003 * tid, tag, rg, read_len=150, error_rate=0.005,
                                  region=(lo, min(hi + 400, len(g["ref"]))), insert_mean=400.0, insert_sd=40.0, monotone_starts=True)
    keys, blob = [], bytearray()
    for a, b in pairs:
        for r in (a, b):
            rec, rl = _encode(r, tid)
            keys.append((r.pos, r.qname, r.flag, rl, len(blob), len(rec)))
            blob += rec
    return which, i, keys, bytes(blob)


def _write(path, refs, sample, rg, parts, block=60000):
    text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join(f"@SQ\tSN:{n}\tLN:{l}\n" for n, l in refs) + f"@RG\tID:{rg}\tSM:{sample}\n"
    body = bytearray(b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(refs)))
    for n, l in refs:
        body += struct.pack("<i", len(n) + 1) + n.encode() + b"\0" + struct.pack("<i", l)
    items = []
    for pi, part in enumerate(parts):                               # (keys, blob) of contig 0, or (tid, keys, blob)
        tid, keys = (0, part[0]) if len(part) == 2 else (part[0], part[1])
        items += [(tid, k[0], k[1], k[2], pi, k[3], k[4], k[5]) for k in keys]
    items.sort()
    spans = []
    for tid, pos, _, _, pi, rl, off, ln in items:
        spans.append((tid, pos - 1, pos - 1 + rl, len(body), len(body) + 4 + ln))
        body += struct.pack("<i", ln) + parts[pi][-1][off:off + ln]
    coff = []
    with open(path, "wb") as fh:
        for i in range(0, len(body), block):
            coff.append(fh.tell())
            fh.write(bam_writer._bgzf_block(bytes(body[i:i + block])))
        coff.append(fh.tell())
        fh.write(bam_writer._bgzf_block(b""))
    voff = lambda u: (coff[u // block] << 16) | (u % block)
    out = bytearray(b"BAI\1" + struct.pack("<i", len(refs)))
    per = [({}, {}) for _ in refs]
    for (tid, beg, end, u0, u1) in spans:
        bins, lin = per[tid]
        b = bam_writer._reg2bin(beg, end)
        ch = bins.setdefault(b, [])
        if ch and ch[-1][1] == voff(u0):
            ch[-1][1] = voff(u1)
        else:
            ch.append([voff(u0), voff(u1)])
        for w in range(beg >> 14, ((end - 1) >> 14) + 1):
            lin[w] = min(lin.get(w, voff(u0)), voff(u0))
    for bins, lin in per:
        out += struct.pack("<i", len(bins))
        for b in sorted(bins):
            out += struct.pack("<Ii", b, len(bins[b]))
            for c0, c1 in bins[b]:
                out += struct.pack("<QQ", c0, c1)
        n = (max(lin) + 1) if lin else 0
        out += struct.pack("<i", n)
        prev = 0
        for w in range(n):
            prev = lin.get(w, prev)
            out += struct.pack("<Q", prev)
    with open(path + ".bai", "wb") as fh:
        fh.write(bytes(out))
    return len(items)


def main():
    out = sys.argv[1]
    ref_len = int(sys.argv[2]) if len(sys.argv) > 2 else 500000
    cov_t = float(sys.argv[3]) if len(sys.argv) > 3 else 30.0
    cov_n = float(sys.argv[4]) if len(sys.argv) > 4 else 30.0
    procs = int(sys.argv[5]) if len(sys.argv) > 5 else 1
    os.makedirs(out, exist_ok=True)
    t = time.time()
    if procs <= 1:
        data = synth.make_tumor_normal(ref_len=ref_len, cov_t=cov_t, cov_n=cov_n, ref_seed=22, tumor_seed=101, normal_seed=202,
                                       read_len=150, insert_mean=400.0, insert_sd=40.0, somatic_every=2000, germline_every=1000)
        refs = [(data["rname"], len(data["ref"]))]
        bam_writer.write_bam(os.path.join(out, "tumor.bam"), refs, synth.pairs_to_sorted_reads(data["tumor"]), sample="TUMOR", index=True)
        bam_writer.write_bam(os.path.join(out, "normal.bam"), refs, synth.pairs_to_sorted_reads(data["normal"]), sample="NORMAL", index=True)
        synth.write_fasta(os.path.join(out, "ref.fa"), data["rname"], data["ref"])
        print(f"{out}: {ref_len} bp, {cov_t}x/{cov_n}x, {time.time() - t:.1f} s; region {data['rname']}:1000-{ref_len - 1000}")
        return
    contigs = int(sys.argv[6]) if len(sys.argv) > 6 else 1
    if contigs > 1:
        names = ([f"chr{i}" for i in range(1, 23)] + ["chrX", "chrY"])[:contigs]
        refs, allp, seqs = [], {"T": [], "N": []}, []
        for tid, name in enumerate(names):
            ref = synth.random_reference(ref_len, 22 + tid, 0.0, 0.0)
            variants = synth.plant_variants(ref, 23 + tid, 2000, 1000, dup_prob=0.0)
            germ = [v for v in variants if not v.somatic]
            h0, h1, h2 = synth.build_haplotype(ref, []), synth.build_haplotype(ref, germ), synth.build_haplotype(ref, variants)
            _G.update(ref=ref, rname=name, haps_t=[h0, h1, h2], haps_n=[h0, h1], cov_t=cov_t, cov_n=cov_n, tid=tid, multi=True)
            nst = max(procs, (ref_len + 249999) // 250000)
            bounds = [ref_len * i // nst for i in range(nst + 1)]
            jobs = [(w, i, bounds[i], bounds[i + 1]) for w in ("T", "N") for i in range(nst)]
            with mp.get_context("fork").Pool(procs) as pool:
                res = pool.map(_stretch, jobs, chunksize=1)
            for w in ("T", "N"):
                allp[w] += [(tid, k, b) for ww, i, k, b in sorted((r for r in res if r[0] == w), key=lambda r: r[1])]
            refs.append((name, ref_len)); seqs.append(ref)
            print(f"{name}: simulated, {time.time() - t:.0f} s", flush=True)
        for w, fname, sample in (("T", "tumor.bam", "TUMOR"), ("N", "normal.bam", "NORMAL")):
            n = _write(os.path.join(out, fname), refs, sample, "tumor" if w == "T" else "normal", allp[w])
            print(f"{fname}: {n} alignments, {time.time() - t:.0f} s")
        width, off = 60, 0
        with open(os.path.join(out, "ref.fa"), "w") as f, open(os.path.join(out, "ref.fa.fai"), "w") as fi, open(os.path.join(out, "regions.bed"), "w") as fb:
            for (name, ln), seq in zip(refs, seqs):
                hdr = f">{name}\n"
                f.write(hdr); off += len(hdr)
                fi.write(f"{name}\t{ln}\t{off}\t{width}\t{width + 1}\n")
                for i in range(0, ln, width):
                    f.write(seq[i:i + width] + "\n")
                off += ln + (ln + width - 1) // width
                fb.write(f"{name}\t1000\t{ln - 1000}\n")
        print(f"{out}: {contigs} contigs x {ref_len} bp, {cov_t}x/{cov_n}x, {procs} processes, {time.time() - t:.1f} s; --bed {out}/regions.bed")
        return
    ref = synth.random_reference(ref_len, 22, 0.0, 0.0)
    variants = synth.plant_variants(ref, 23, 2000, 1000, dup_prob=0.0)
    germ = [v for v in variants if not v.somatic]
    h0, h1, h2 = synth.build_haplotype(ref, []), synth.build_haplotype(ref, germ), synth.build_haplotype(ref, variants)
    _G.update(ref=ref, rname="chr22", haps_t=[h0, h1, h2], haps_n=[h0, h1], cov_t=cov_t, cov_n=cov_n)
    nst = max(procs, (ref_len + 249999) // 250000)                  # stretches of <= 250 kb
    bounds = [ref_len * i // nst for i in range(nst + 1)]
    jobs = [(w, i, bounds[i], bounds[i + 1]) for w in ("T", "N") for i in range(nst)]
    with mp.get_context("fork").Pool(procs) as pool:
        res = pool.map(_stretch, jobs, chunksize=1)
    refs = [("chr22", ref_len)]
    for w, name, sample in (("T", "tumor.bam", "TUMOR"), ("N", "normal.bam", "NORMAL")):
        parts = [(k, b) for ww, i, k, b in sorted((r for r in res if r[0] == w), key=lambda r: r[1])]
        n = _write(os.path.join(out, name), refs, sample, "tumor" if w == "T" else "normal", parts)
        print(f"{name}: {n} alignments, {time.time() - t:.0f} s")
    synth.write_fasta(os.path.join(out, "ref.fa"), "chr22", ref)
    print(f"{out}: {ref_len} bp, {cov_t}x/{cov_n}x, {procs} processes, {time.time() - t:.1f} s; region chr22:1000-{ref_len - 1000}")


if __name__ == "__main__":
    main()
