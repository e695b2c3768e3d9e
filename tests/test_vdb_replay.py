"""Rank 0's replay of a large batch (host_vdb.cc: keys hashed on several threads, one map shard per thread, the x86 SHA
extensions when the CPU has them) against (1) the same records added in small single-threaded batches with the scalar
SHA-256 and (2) a Python model of VariantDB_t::addVar (reference src/VariantDB.cc:28-91) keyed with hashlib's sha256.
The map's iteration order feeds an unstable std::sort, so the VCF bytes themselves are compared across configurations."""
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, ctypes as C
sys.path.insert(0, %(root)r)
import numpy as np
from lancet_amd import abi, engine
sys.path.insert(0, %(root)r + "/tests")
import test_vdb_replay as T
recs, blob, lr, ids, bxn = T.make_records(int(sys.argv[1]), sys.argv[3] == "lr")
chunk = int(sys.argv[2])
db = engine.VariantDB()
vdt = np.dtype(abi.LancetVariant); ldt = np.dtype(abi.LancetVariantLR)
for o in range(0, len(recs), chunk):
    r = np.ascontiguousarray(recs[o:o + chunk])
    vp = r.ctypes.data_as(C.POINTER(abi.LancetVariant))
    if lr is not None:
        l = np.ascontiguousarray(lr[o:o + chunk])
        db.add_raw_lr(vp, l.ctypes.data_as(C.POINTER(abi.LancetVariantLR)), len(r), blob, ids.ctypes.data_as(C.POINTER(C.c_uint32)), bxn, ["chr1", "chr10", "chr2"])
    else:
        db.add_raw(vp, len(r), blob, ["chr1", "chr10", "chr2"])
sys.stdout.write(db.vcf(date_line="##fileDate=x\n"))
"""


def make_records(n, with_lr, seed=5):
    """n records over few positions (many repeated keys, ties and larger-coverage replacements) of every type."""
    sys.path.insert(0, ROOT)
    from lancet_amd import abi
    rng = np.random.default_rng(seed)
    vdt = np.dtype(abi.LancetVariant)
    recs = np.zeros(n, dtype=vdt)
    alleles = [b"A", b"C", b"G", b"T", b"AC", b"GGT", b"A-C", b"--T", b"AC-GT-A", b"T" * 40, b"ACGT" * 30]
    blob = bytearray()
    offs = []
    for a in alleles:
        offs.append((len(blob), len(a))); blob += a
    strs = [b"", b"3A", b"12AC"]
    soffs = []
    for a in strs:
        soffs.append((len(blob), len(a))); blob += a
    recs["window"] = np.arange(n) // 3
    recs["seq_in_window"] = np.arange(n) % 3
    recs["chr_id"] = rng.integers(0, 3, n)
    recs["pos"] = rng.integers(100, 160, n)
    recs["code"] = np.frombuffer(b"x^vc", dtype=np.uint8)[rng.integers(0, 4, n)]
    recs["prev_bp_ref"] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n)]
    recs["prev_bp_alt"] = recs["prev_bp_ref"]
    recs["kmer"] = rng.integers(11, 101, n)
    recs["cov"] = rng.integers(0, 6, (n, 8))
    ra = rng.integers(0, len(alleles), n); aa = rng.integers(0, len(alleles), n); sa = rng.integers(0, len(strs), n)
    o = np.array(offs, dtype=np.uint32); so = np.array(soffs, dtype=np.uint32)
    recs["ref_off"], recs["ref_len"] = o[ra, 0], o[ra, 1]
    recs["alt_off"], recs["alt_len"] = o[aa, 0], o[aa, 1]
    recs["str_off"], recs["str_len"] = so[sa, 0], so[sa, 1]
    lr = ids = bxn = None
    if with_lr:
        lr = np.zeros(n, dtype=np.dtype(abi.LancetVariantLR))
        lr["hp"] = rng.integers(0, 4, (n, 12))
        bxn = ["BX%03d" % i for i in range(50)]
        ids = np.sort(rng.integers(0, 50, 4096)).astype(np.uint32)
        lr["bx_off"] = rng.integers(0, 4000, (n, 4))
        lr["bx_len"] = rng.integers(0, 4, (n, 4))
    return recs, bytes(blob) + b"\0", lr, ids, bxn


def model(recs, blob):
    """addVar on the normalised records: {hex key: (pos, ref, alt, kmer, cov)}"""
    chrs = ["chr1", "chr10", "chr2"]
    db = {}
    for r in recs:
        ref = blob[r["ref_off"]:r["ref_off"] + r["ref_len"]].decode(); alt = blob[r["alt_off"]:r["alt_off"] + r["alt_len"]].decode()
        code = chr(r["code"]); pos = int(r["pos"]); typ = "?"; ln = 0
        if code == "^": typ = "I"; ref = ""; ln = len(alt)
        if code == "v": typ = "D"; alt = ""; ln = len(ref)
        if code == "x": typ = "S"; pos += 1
        if code == "c":
            typ = "C"; ref = ref.replace("-", ""); alt = alt.replace("-", ""); ln = abs(len(ref) - len(alt)) or len(alt)
        if typ != "S":
            ref = chr(r["prev_bp_alt"]) + ref; alt = chr(r["prev_bp_alt"]) + alt
        else:
            ln = 1
        sig = "%s:%d:%s:%d:%s:%s" % (chrs[r["chr_id"]], pos, typ, ln, ref, alt)
        key = hashlib.sha256(sig.encode()).hexdigest()
        cov = [int(c) for c in r["cov"]]
        if key in db:
            if sum(db[key][4]) < sum(cov):
                db[key] = db[key][:3] + (int(r["kmer"]), cov)
        else:
            db[key] = (chrs[r["chr_id"]], pos, (ref, alt), int(r["kmer"]), cov)
    return db


def run_child(n, chunk, mode, env_extra):
    env = dict(os.environ); env.update(env_extra)
    r = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT), str(n), str(chunk), mode], capture_output=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    return r.stdout.decode()


@pytest.mark.parametrize("mode", ["plain", "lr"])
def test_large_batch_replay_equals_small_batches(mode):
    n = 70000
    base = run_child(n, 1000, mode, {"LANCET_NO_SHA_NI": "1", "LANCET_VDB_THREADS": "1"})      # scalar sha, one thread, small adds
    for env in ({"LANCET_VDB_THREADS": "7"}, {"LANCET_VDB_THREADS": "16", "LANCET_NO_SHA_NI": "1"}, {}):
        assert run_child(n, n, mode, env) == base, env
    assert run_child(n, 40000, mode, {"LANCET_VDB_THREADS": "3"}) == base                           # threaded add onto a filled map
    if mode == "plain":
        recs, blob, _, _, _ = make_records(n, False)
        want = model(recs, blob)
        lines = [l.split("\t") for l in base.splitlines() if not l.startswith("#")]
        want = {k: v for k, v in want.items() if sum(v[4][4:]) > 0}       # printVCF prints nothing without alt support (src/Variant.cc:59-63)
        assert len(lines) == len(want)
        got = set()
        for f in lines:
            k = [x for x in f[7].split(";") if x.startswith("KMERSIZE=")][0][9:]
            nrm, tum = f[9].split(":"), f[10].split(":")
            cov = [int(x) for x in (nrm[2].split(",") + tum[2].split(",") + nrm[3].split(",") + tum[3].split(","))]
            got.add((f[0], int(f[1]), f[3], f[4], int(k), tuple(cov)))
        exp = set((c, p, ra[0], ra[1], k, tuple(cov)) for (c, p, ra, k, cov) in want.values())
        assert got == exp


@pytest.mark.parametrize("mode", ["plain", "lr"])
def test_keys_made_by_the_senders_and_reduced_streams_give_the_same_database(mode):
    """Rank 0 of an N-rank step only inserts: the records arrive with their addVar keys (lancet_vdb_keys on the rank that made them) and
    without the records that cannot change the database (lancet_vdb_reduce: per key the first record and the first of the largest total
    coverage).  70 000 records with many repeated keys, ties and replacements, dealt out over 8 ranks in chunks of windows (so that a
    key's records lie on several ranks, out of order at rank 0): the VCF bytes equal those of one plain lancet_vdb_add, with and
    without the reduction; the keys equal hashlib's sha256 of the reference's signature string."""
    import ctypes as C
    sys.path.insert(0, ROOT)
    from lancet_amd import abi, dist as ldist, engine
    n = 70000
    recs, blob, lr, ids, bxn = make_records(n, mode == "lr")
    chrs = ["chr1", "chr10", "chr2"]
    vp = lambda a: a.ctypes.data_as(C.POINTER(abi.LancetVariant))
    lp = lambda a: a.ctypes.data_as(C.POINTER(abi.LancetVariantLR))
    base = engine.VariantDB()
    if lr is not None:
        base.add_raw_lr(vp(recs), lp(lr), n, blob, ids.ctypes.data_as(C.POINTER(C.c_uint32)), bxn, chrs)
    else:
        base.add_raw(vp(recs), n, blob, chrs)
    want = base.vcf(date_line="##fileDate=x\n")
    keys = engine.record_keys(vp(recs), n, blob, chrs)
    for i in (0, 1, 777, n - 1):                                   # the key IS sha256(getSignature())
        r = recs[i]
        ref = blob[r["ref_off"]:r["ref_off"] + r["ref_len"]].decode(); alt = blob[r["alt_off"]:r["alt_off"] + r["alt_len"]].decode()
        code = chr(r["code"]); pos = int(r["pos"]); typ = "?"; ln = 0
        if code == "^": typ = "I"; ref = ""; ln = len(alt)
        if code == "v": typ = "D"; alt = ""; ln = len(ref)
        if code == "x": typ = "S"; pos += 1
        if code == "c":
            typ = "C"; ref = ref.replace("-", ""); alt = alt.replace("-", ""); ln = abs(len(ref) - len(alt)) or len(alt)
        if typ != "S":
            ref = chr(r["prev_bp_alt"]) + ref; alt = chr(r["prev_bp_alt"]) + alt
        else:
            ln = 1
        assert bytes(keys[i]) == hashlib.sha256(("%s:%d:%s:%d:%s:%s" % (chrs[r["chr_id"]], pos, typ, ln, ref, alt)).encode()).digest()
    keep = engine.records_that_matter(vp(recs), keys, n)
    assert 0 < keep.sum() < n - 5000                               # (repeated keys: ties and smaller totals drop out)
    world = 8
    for reduce in (False, True):
        parts = []
        for rank in range(world):
            mine = np.array(ldist.shard_windows(int(recs["window"].max()) + 1, rank, world, chunk=5), dtype=np.int64)
            sel = np.isin(recs["window"], mine)
            sub = np.ascontiguousarray(recs[sel])
            local = np.searchsorted(mine, sub["window"])           # the rank's own window numbering
            sub["window"] = local
            kw = {}
            if lr is not None:
                sl = np.ascontiguousarray(lr[sel])
                parts.append(ldist.pack_records(vp(sub), len(sub), blob, lp(sl), ids.ctypes.data_as(C.POINTER(C.c_uint32)), bxn,
                                                chr_names=chrs, window_index=mine, reduce=reduce))
            else:
                parts.append(ldist.pack_records(vp(sub), len(sub), blob, chr_names=chrs, window_index=mine, reduce=reduce))
        db = engine.VariantDB()
        added = ldist.merge_into_vdb(parts, db)
        assert (added < n - 3000) if reduce else (added == n)
        assert db.vcf(date_line="##fileDate=x\n") == want, reduce
