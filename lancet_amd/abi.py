"""ctypes mirror of include/lancet_engine.h (struct layouts only; no logic)."""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence

import numpy as np


class LancetParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "min_k", "max_k", "max_tip_len", "cov_threshold", "low_cov_threshold", "dfs_limit", "max_indel_len",
        "max_mismatch", "min_qual_trim", "min_qual_call", "max_unit_len", "min_report_units", "min_report_len",
        "dist_from_str", "lr_mode", "reserved")] + [("min_cov_ratio", C.c_double)]


def default_params(**over) -> LancetParams:
    """Reference defaults, src/Lancet.hh:33-81."""
    p = LancetParams(min_k=11, max_k=101, max_tip_len=11, cov_threshold=5, low_cov_threshold=1, dfs_limit=1000000,
                     max_indel_len=500, max_mismatch=2, min_qual_trim=10 + 33, min_qual_call=17 + 33,
                     max_unit_len=4, min_report_units=3, min_report_len=7, dist_from_str=1, lr_mode=0, reserved=0,
                     min_cov_ratio=0.01)
    for k, v in over.items():
        setattr(p, k, v)
    return p


class LancetWindowBatch(C.Structure):
    _fields_ = [
        ("n_windows", C.c_int32),
        ("chr_id", C.POINTER(C.c_int32)), ("ref_start", C.POINTER(C.c_int32)),
        ("ref_off", C.POINTER(C.c_uint32)), ("ref_bases", C.c_char_p),
        ("read_begin", C.POINTER(C.c_uint32)), ("seq_off", C.POINTER(C.c_uint32)),
        ("seq", C.c_char_p), ("qual", C.c_char_p),
        ("label", C.POINTER(C.c_uint8)), ("strand", C.POINTER(C.c_uint8)), ("mate", C.POINTER(C.c_uint8)),
        ("mapped", C.POINTER(C.c_uint8)), ("name_rank", C.POINTER(C.c_uint32)),
        ("bx_rank", C.POINTER(C.c_uint32)), ("hp", C.POINTER(C.c_uint8)),
    ]


class LancetPackedReads(C.Structure):
    """include/lancet_engine.h lancet_packed_reads: the reads of a batch trimmed and packed by the caller."""
    _fields_ = [("struct_size", C.c_uint32), ("reserved", C.c_uint32),
                ("rinfo", C.POINTER(C.c_uint32)), ("base_woff", C.POINTER(C.c_uint32)), ("good_woff", C.POINTER(C.c_uint32)),
                ("bases", C.POINTER(C.c_uint32)), ("good", C.POINTER(C.c_uint32)), ("min_qual_trim", C.c_int32), ("min_qual_call", C.c_int32),
                ("read_index", C.POINTER(C.c_uint32)), ("n_distinct", C.c_uint32), ("reserved2", C.c_uint32)]


class LancetVariantLR(C.Structure):
    _fields_ = [("hp", C.c_uint16 * 12), ("bx_off", C.c_uint32 * 4), ("bx_len", C.c_uint32 * 4), ("reserved", C.c_uint32 * 2)]


class LancetVariant(C.Structure):
    _fields_ = [
        ("window", C.c_int32), ("seq_in_window", C.c_int32), ("chr_id", C.c_int32), ("pos", C.c_int32),
        ("code", C.c_uint8), ("prev_bp_ref", C.c_uint8), ("prev_bp_alt", C.c_uint8), ("reserved", C.c_uint8),
        ("kmer", C.c_uint16), ("cov", C.c_uint16 * 8), ("reserved2", C.c_uint16),
        ("ref_off", C.c_uint32), ("ref_len", C.c_uint32), ("alt_off", C.c_uint32), ("alt_len", C.c_uint32),
        ("str_off", C.c_uint32), ("str_len", C.c_uint32),
    ]


class LancetWindowStats(C.Structure):
    _fields_ = [("status", C.c_int32), ("final_k", C.c_int32), ("n_builds", C.c_int32), ("n_variants", C.c_int32),
                ("n_kmers", C.c_uint64), ("max_nodes", C.c_uint32), ("sum_nodes", C.c_uint32)]


class LancetFilters(C.Structure):
    _fields_ = [("min_phred_fisher_str", C.c_double), ("min_phred_fisher", C.c_double),
                ("max_vaf_normal", C.c_double), ("min_vaf_tumor", C.c_double),
                ("min_cov_normal", C.c_int32), ("max_cov_normal", C.c_int32), ("min_cov_tumor", C.c_int32),
                ("max_cov_tumor", C.c_int32), ("min_alt_cnt_tumor", C.c_int32), ("max_alt_cnt_normal", C.c_int32),
                ("min_strand_bias", C.c_int32), ("reserved", C.c_int32)]


def _p(a: np.ndarray, t):
    return a.ctypes.data_as(C.POINTER(t))


def batch_to_c(b) -> LancetWindowBatch:
    """b: lancet_amd.frontend.WindowBatch (numpy SoA).  The returned struct borrows b's arrays."""
    cb = LancetWindowBatch()
    cb.n_windows = b.n_windows
    cb.chr_id = _p(b.chr_id, C.c_int32)
    cb.ref_start = _p(b.ref_start, C.c_int32)
    cb.ref_off = _p(b.ref_off, C.c_uint32)
    cb.ref_bases = b.ref_bases.ctypes.data_as(C.c_char_p)
    cb.read_begin = _p(b.read_begin, C.c_uint32)
    cb.seq_off = _p(b.seq_off, C.c_uint32)
    cb.seq = b.seq.ctypes.data_as(C.c_char_p)
    cb.qual = b.qual.ctypes.data_as(C.c_char_p)
    cb.label = _p(b.label, C.c_uint8)
    cb.strand = _p(b.strand, C.c_uint8)
    cb.mate = _p(b.mate, C.c_uint8)
    cb.mapped = _p(b.mapped, C.c_uint8)
    cb.name_rank = _p(b.name_rank, C.c_uint32)
    if getattr(b, "bx_rank", None) is not None:
        cb.bx_rank = _p(b.bx_rank, C.c_uint32)
        cb.hp = _p(b.hp, C.c_uint8)
    cb._keep = b
    return cb


def variants_to_py(vptr, n: int, blob: bytes) -> List[dict]:
    out = []
    for i in range(n):
        v = vptr[i]
        out.append(dict(
            window=v.window, seq=v.seq_in_window, chr_id=v.chr_id, pos=v.pos, code=chr(v.code),
            prev_bp_ref=chr(v.prev_bp_ref), prev_bp_alt=chr(v.prev_bp_alt), kmer=v.kmer, cov=tuple(v.cov),
            ref=blob[v.ref_off:v.ref_off + v.ref_len].decode(), alt=blob[v.alt_off:v.alt_off + v.alt_len].decode(),
            str=blob[v.str_off:v.str_off + v.str_len].decode()))
    return out


def variants_lr_to_py(out: List[dict], lrptr, bx_blob) -> List[dict]:
    """Adds the linked-read annotations (hp: 12 counts, bx: 4 tuples of barcode ranks) to variants_to_py's dicts."""
    for i, d in enumerate(out):
        l = lrptr[i]
        d["hp"] = tuple(l.hp)
        d["bx"] = tuple(tuple(int(bx_blob[l.bx_off[q] + j]) for j in range(l.bx_len[q])) for q in range(4))
    return out


def c_string_array(strs: Sequence[str]):
    arr = (C.c_char_p * len(strs))(*[s.encode() for s in strs])
    return arr
