"""TEST INFRASTRUCTURE ONLY -- ctypes loader for oracle/liblancet_oracle.so and oracle/_ref/libalign_ref.so.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

from lancet_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None


def build(quiet: bool = True) -> None:
    subprocess.run(["make", "-C", _HERE] + (["-s"] if quiet else []), check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liblancet_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.lancet_oracle_run.restype = C.c_void_p
        L.lancet_oracle_run.argtypes = [C.POINTER(abi.LancetParams), C.POINTER(abi.LancetWindowBatch),
                                        C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_int]
        L.lancet_oracle_n_variants.restype = C.c_uint32
        L.lancet_oracle_n_variants.argtypes = [C.c_void_p]
        L.lancet_oracle_variants.restype = C.POINTER(abi.LancetVariant)
        L.lancet_oracle_variants.argtypes = [C.c_void_p]
        L.lancet_oracle_blob.restype = C.c_void_p
        L.lancet_oracle_blob.argtypes = [C.c_void_p]
        L.lancet_oracle_blob_len.restype = C.c_uint32
        L.lancet_oracle_blob_len.argtypes = [C.c_void_p]
        L.lancet_oracle_variants_lr.restype = C.POINTER(abi.LancetVariantLR)
        L.lancet_oracle_variants_lr.argtypes = [C.c_void_p]
        L.lancet_oracle_bx_blob.restype = C.POINTER(C.c_uint32)
        L.lancet_oracle_bx_blob.argtypes = [C.c_void_p]
        L.lancet_oracle_stats.restype = C.POINTER(abi.LancetWindowStats)
        L.lancet_oracle_stats.argtypes = [C.c_void_p]
        L.lancet_oracle_trace.restype = C.c_char_p
        L.lancet_oracle_trace.argtypes = [C.c_void_p]
        L.lancet_oracle_free.argtypes = [C.c_void_p]
        L.lancet_oracle_align.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
        L.lancet_oracle_std_hash.restype = C.c_uint64
        L.lancet_oracle_std_hash.argtypes = [C.c_char_p]
        L.lancet_oracle_find_tandems.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                 C.POINTER(C.c_int), C.c_char_p, C.c_int]
        _LIB = L
    return _LIB


def run(batch, params=None, verbose: bool = False):
    """Returns (variants [dict], stats [dict], trace str)."""
    L = lib()
    p = params or abi.default_params()
    cb = abi.batch_to_c(batch)
    chroms = sorted(set(batch.chrom), key=lambda c: batch.chrom.index(c))
    id2chr = {}
    for c, i in zip(batch.chrom, batch.chr_id):
        id2chr[int(i)] = c
    names = [id2chr.get(i, str(i)) for i in range(max(id2chr) + 1)] if id2chr else []
    cn = abi.c_string_array(names)
    hd = abi.c_string_array(batch.hdr)
    h = L.lancet_oracle_run(C.byref(p), C.byref(cb), cn, hd, 1 if verbose else 0)
    try:
        n = L.lancet_oracle_n_variants(h)
        blob = C.string_at(L.lancet_oracle_blob(h), L.lancet_oracle_blob_len(h)) if L.lancet_oracle_blob_len(h) else b""
        variants = abi.variants_to_py(L.lancet_oracle_variants(h), n, blob)
        if p.lr_mode:
            abi.variants_lr_to_py(variants, L.lancet_oracle_variants_lr(h), L.lancet_oracle_bx_blob(h))
        sp = L.lancet_oracle_stats(h)
        stats = [dict(status=sp[i].status, final_k=sp[i].final_k, n_builds=sp[i].n_builds, n_variants=sp[i].n_variants,
                      n_kmers=sp[i].n_kmers, max_nodes=sp[i].max_nodes, sum_nodes=sp[i].sum_nodes) for i in range(batch.n_windows)]
        trace = L.lancet_oracle_trace(h).decode()
    finally:
        L.lancet_oracle_free(h)
    return variants, stats, trace


def align(S: str, T: str):
    L = lib()
    cap = len(S) + len(T) + 8
    a = C.create_string_buffer(cap)
    b = C.create_string_buffer(cap)
    n = L.lancet_oracle_align(S.encode(), T.encode(), a, b, cap)
    assert n >= 0
    return a.value.decode(), b.value.decode()


def ref_align_available() -> bool:
    return os.path.exists(os.path.join(_HERE, "_ref", "libalign_ref.so"))


def ref_align(S: str, T: str):
    """The reference's own global_align_aff (oracle/_ref/libalign_ref.so, built from reference src/align.cc)."""
    global _REF
    if _REF is None:
        _REF = C.CDLL(os.path.join(_HERE, "_ref", "libalign_ref.so"))
        _REF.lancet_ref_align.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
    cap = len(S) + len(T) + 8
    a = C.create_string_buffer(cap)
    b = C.create_string_buffer(cap)
    n = _REF.lancet_ref_align(S.encode(), T.encode(), a, b, cap)
    assert n >= 0
    return a.value.decode(), b.value.decode()


def std_hash(s: str) -> int:
    return int(lib().lancet_oracle_std_hash(s.encode()))


def find_tandems(seq: str, pos: int, max_unit_len=4, min_report_units=3, min_report_len=7, dist_from_str=1):
    L = lib()
    ln = C.c_int(0)
    motif = C.create_string_buffer(256)
    a = L.lancet_oracle_find_tandems(seq.encode(), max_unit_len, min_report_units, min_report_len, dist_from_str, pos,
                                     C.byref(ln), motif, 256)
    return bool(a), ln.value, motif.value.decode()
