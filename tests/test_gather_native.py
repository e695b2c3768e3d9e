"""The native N-process record gather (include/lancet_gather.h, lancet_amd/csrc/host_gather.cc) on CPU: pack -> gather -> merge with
8 "ranks", against the Python harness's implementation of the same byte format (lancet_amd/dist.py) and against the reference's
single-process VCF (what is replaced: the merge of per-thread databases, reference src/Lancet.cc:940-959).
Records come from the oracle (no GPU here).  The RCCL transport itself needs GPUs: the `files` transport (LANCET_COMM_TEST_FILES=1,
test use only) carries the payloads of separate processes here; tests/test_host_native.py runs `lancet_gpu --ranks` on the GPU box."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import golden_util as gu
from test_dist_gloo import records_to_c

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    from lancet_amd import engine
    L = engine.lib()
    L.lancet_records_pack.restype = C.c_int
    L.lancet_records_pack.argtypes = [C.c_void_p, C.c_uint32, C.c_char_p, C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(C.c_char_p), C.c_uint32,
                                      C.POINTER(C.c_char_p), C.c_int32, C.POINTER(C.c_int64), C.c_uint32, C.c_int, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
    L.lancet_records_merge.restype = C.c_int
    L.lancet_records_merge.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_int, C.POINTER(C.c_uint32)]
    L.lancet_comm_create.restype = C.c_void_p
    L.lancet_comm_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_double, C.c_char_p, C.c_size_t]
    L.lancet_comm_gather.restype = C.c_int
    L.lancet_comm_gather.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
    L.lancet_comm_destroy.argtypes = [C.c_void_p]
    L.lancet_comm_transport.restype = C.c_char_p
    L.lancet_comm_transport.argtypes = [C.c_void_p]
    L.lancet_free.argtypes = [C.c_void_p]
    return L


def _strs(names):
    return (C.c_char_p * max(1, len(names)))(*[n.encode() for n in names])


def native_pack(L, recs, lr, bx_names, chr_names, window_index, reduce=True):
    arr, blob, lra, bxb = records_to_c(recs, lr)
    out = C.POINTER(C.c_uint8)(); n = C.c_size_t()
    wi = (C.c_int64 * max(1, len(window_index)))(*window_index)
    rc = L.lancet_records_pack(arr, len(recs), blob, len(blob), lra if lr else None, bxb if lr else None, _strs(bx_names or []) if lr else None,
                               len(bx_names or []) if lr else 0, _strs(chr_names), len(chr_names), wi, len(window_index), 1 if reduce else 0, C.byref(out), C.byref(n))
    assert rc == 0
    b = C.string_at(out, n.value)
    L.lancet_free(out)
    return b, (arr, blob, lra, bxb)


def native_merge(L, parts):
    from lancet_amd import engine
    db = engine.VariantDB()
    pa = (C.c_char_p * max(1, len(parts)))(*parts)
    la = (C.c_size_t * max(1, len(parts)))(*[len(p) for p in parts])
    added = C.c_uint32()
    rc = L.lancet_records_merge(db.h, pa, la, len(parts), C.byref(added))
    assert rc == 0
    return db, added.value


def _rank_records(case, world, chunk=3):
    """the windows of a golden dealt out `chunk` at a time; per rank: its records (local window numbers) and its windows' global numbers"""
    from lancet_amd import dist as ldist, workload
    from oracle import oracle
    meta, batch, kept, _ = gu.case_batch(case)
    p = gu.params(meta)
    out = []
    for rank in range(world):
        mine = ldist.shard_windows(batch.n_windows, rank, world, chunk=chunk)
        recs = []
        for li, w in enumerate(mine):
            v, _, _ = oracle.run(workload.sub_batch(batch, w, w + 1), p)
            for r in v:
                r["window"] = li
                recs.append(r)
        out.append((recs, mine))
    return meta, batch, out


@pytest.mark.parametrize("case", ["tile30", "lr30"])
def test_native_pack_is_the_harness_s_byte_format_and_eight_ranks_merge_to_the_reference_vcf(case):
    from lancet_amd import build, dist as ldist, engine
    build.build()
    L = _lib()
    meta, batch, ranks = _rank_records(case, 8)
    lr = gu.case_lr(meta)
    parts = []
    for recs, mine in ranks:
        b, (arr, blob, lra, bxb) = native_pack(L, recs, lr, batch.bx_names, ["chr22"], mine)
        # the harness packs the same arrays into the same bytes (and unpacks what the native side packed)
        assert b == ldist.pack_records(arr, len(recs), blob, lra, bxb, batch.bx_names, chr_names=["chr22"], window_index=mine)
        u = ldist.unpack_records(b)
        assert u["n"] <= len(recs) and u["chr_names"] == ["chr22"] and (u["keys"] is None) == (u["n"] == 0)
        parts.append(b)
    assert sum(len(r) for r, _ in ranks) > 0
    # rank order is NOT window order here (interleaved runs of three windows): the merge sorts, keyed and reduced
    db, added = native_merge(L, parts)
    assert added > 0 and db.vcf() == gu.golden_vcf(case)
    # any arrival order of the parts, and unreduced parts, give the same database
    db2, _ = native_merge(L, parts[::-1])
    assert db2.vcf() == gu.golden_vcf(case)
    whole = [native_pack(L, recs, lr, batch.bx_names, ["chr22"], mine, reduce=False)[0] for recs, mine in ranks]
    db3, added3 = native_merge(L, whole)
    assert db3.vcf() == gu.golden_vcf(case) and added3 >= added
    # the harness's merge of the natively packed parts: same VCF
    db4 = engine.VariantDB()
    ldist.merge_into_vdb(parts, db4)
    assert db4.vcf() == gu.golden_vcf(case)


def test_native_merge_refuses_damaged_parts():
    from lancet_amd import build
    build.build()
    L = _lib()
    meta, batch, ranks = _rank_records("tile30", 2)
    b, _ = native_pack(L, ranks[0][0], False, None, ["chr22"], ranks[0][1])
    from lancet_amd import engine
    for bad in (b[:40], b[:len(b) - 7], b[:8] + b"\xff" * 8 + b[16:]):
        db = engine.VariantDB()
        pa = (C.c_char_p * 1)(bad); la = (C.c_size_t * 1)(len(bad))
        assert L.lancet_records_merge(db.h, pa, la, 1, None) != 0


_RANK = r"""
import ctypes as C, os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import test_gather_native as T
L = T._lib()
rank, world, path = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
err = C.create_string_buffer(512)
c = L.lancet_comm_create(rank, world, 0, path.encode(), 60.0, err, 512)
assert c, err.value
assert L.lancet_comm_transport(c) == b"files"
payload = (b"rank%d:" % rank) * (1000 * rank + 1) if rank != 2 else b""
for rnd in range(2):                               # two gathers over one communicator
    out = C.POINTER(C.c_uint8)(); lens = (C.c_size_t * world)()
    assert L.lancet_comm_gather(c, payload, len(payload), C.byref(out), lens) == 0
    if rank == 0:
        tot = sum(lens)
        got = C.string_at(out, tot)
        want = b"".join(((b"rank%d:" % r) * (1000 * r + 1) if r != 2 else b"") for r in range(world))
        assert got == want and list(lens) == [len((b"rank%d:" % r) * (1000 * r + 1)) if r != 2 else 0 for r in range(world)]
        L.lancet_free(out)
    else:
        assert not out
L.lancet_comm_destroy(c)
print("ok", rank)
"""


def test_gather_between_processes_over_the_test_transport(tmp_path):
    """4 processes, variable sizes (one empty), two rounds: what rank 0 receives is every rank's payload in rank order."""
    from lancet_amd import build
    build.build()
    path = str(tmp_path / "rv.id")
    env = dict(os.environ, LANCET_COMM_TEST_FILES="1")
    procs = [subprocess.Popen([sys.executable, "-c", _RANK.format(root=ROOT), str(r), "4", path], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(4)]
    for r, p in enumerate(procs):
        out, err = p.communicate(timeout=120)
        assert p.returncode == 0 and out.strip() == f"ok {r}", err[-2000:]
