"""ctypes binding of include/lancet_host.h: the native host side (BAM / FASTA input, window tiling, per-window filters
and read selection, batch assembly) that `lancet_amd/bin/lancet_gpu` runs in front of the engine.

`NativeHost.batch()` returns the same `frontend.WindowBatch` as `frontend.batch_from_sam` (tests/test_host_native.py
holds the two against each other); it is pure CPU code and needs neither a GPU nor the engine."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Tuple

import numpy as np

from . import abi, engine, frontend


class LancetHostOpts(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("padding", "window_size", "min_map_qual", "max_delta_as_xs", "primary_alignment_only",
                                          "xa_filter", "max_avg_cov", "max_k", "linked", "active_region", "min_evidence",
                                          "min_qual_call")]


_BOUND = False


def _lib():
    global _BOUND
    L = engine.lib()
    if not _BOUND:
        L.lancet_host_opts_default.argtypes = [C.POINTER(LancetHostOpts)]
        L.lancet_host_open.restype = C.c_void_p
        L.lancet_host_open.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]
        L.lancet_host_close.argtypes = [C.c_void_p]
        L.lancet_host_last_error.restype = C.c_char_p
        L.lancet_host_last_error.argtypes = [C.c_void_p]
        L.lancet_host_sample.restype = C.c_char_p
        L.lancet_host_sample.argtypes = [C.c_void_p, C.c_int]
        L.lancet_host_tile.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(LancetHostOpts)]
        L.lancet_host_tile_regions.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_char_p), C.c_int, C.POINTER(LancetHostOpts)]
        L.lancet_host_chrom.restype = C.c_char_p
        L.lancet_host_chrom.argtypes = [C.c_void_p]
        L.lancet_host_chroms.restype = C.POINTER(C.c_char_p)
        L.lancet_host_chroms.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.lancet_host_first_has_md.argtypes = [C.c_void_p, C.c_int]
        L.lancet_host_set_rg_file.argtypes = [C.c_void_p, C.c_char_p]
        L.lancet_host_window_chrom.argtypes = [C.c_void_p, C.c_int]
        L.lancet_host_window_span.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.lancet_host_window_hdr.restype = C.c_char_p
        L.lancet_host_window_hdr.argtypes = [C.c_void_p, C.c_int]
        L.lancet_host_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(LancetHostOpts), C.POINTER(abi.LancetWindowBatch),
                                        C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.lancet_host_batch_packed.restype = C.c_int
        L.lancet_host_batch_packed.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(LancetHostOpts), C.POINTER(abi.LancetParams), C.POINTER(abi.LancetWindowBatch),
                                               C.POINTER(abi.LancetPackedReads), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.lancet_host_bx_names.restype = C.POINTER(C.c_char_p)
        L.lancet_host_bx_names.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        _BOUND = True
    return L


def default_opts(**over) -> LancetHostOpts:
    o = LancetHostOpts()
    _lib().lancet_host_opts_default(C.byref(o))
    for k, v in over.items():
        setattr(o, k, int(v))
    return o


class NativeHost:
    def __init__(self, tumor_bam: str, normal_bam: str, ref_fasta: str):
        self.L = _lib()
        err = C.create_string_buffer(512)
        self.h = self.L.lancet_host_open(tumor_bam.encode(), normal_bam.encode(), ref_fasta.encode(), err, 512)
        if not self.h:
            raise engine.EngineError(err.value.decode())
        self.n_windows = 0

    def close(self):
        if getattr(self, "h", None):
            self.L.lancet_host_close(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def sample(self, tumor: bool) -> str:
        return self.L.lancet_host_sample(self.h, 1 if tumor else 0).decode()

    def set_rg_file(self, path: Optional[str]) -> None:
        """--rg-file: keep only the alignments of the read groups named in the file (None: all)."""
        if self.L.lancet_host_set_rg_file(self.h, path.encode() if path else None) != 0:
            raise engine.EngineError(self.L.lancet_host_last_error(self.h).decode())

    def tile(self, region: str, opts: LancetHostOpts) -> List[str]:
        n = self.L.lancet_host_tile(self.h, region.encode(), C.byref(opts))
        if n < 0:
            raise engine.EngineError(self.L.lancet_host_last_error(self.h).decode())
        self.n_windows = n
        return [self.L.lancet_host_window_hdr(self.h, w).decode() for w in range(n)]

    def tile_regions(self, regions: List[str], opts: LancetHostOpts, bed: Optional[str] = None) -> List[str]:
        """--bed and/or several regions into one table of windows (processing order = header order, across contigs)."""
        arr = (C.c_char_p * max(1, len(regions)))(*[r.encode() for r in regions])
        n = self.L.lancet_host_tile_regions(self.h, bed.encode() if bed else None, arr, len(regions), C.byref(opts))
        if n < 0:
            raise engine.EngineError(self.L.lancet_host_last_error(self.h).decode())
        self.n_windows = n
        return [self.L.lancet_host_window_hdr(self.h, w).decode() for w in range(n)]

    def chroms(self) -> List[str]:
        n = C.c_int()
        p = self.L.lancet_host_chroms(self.h, C.byref(n))
        return [p[i].decode() for i in range(n.value)]

    def first_has_md(self, tumor: bool) -> bool:
        return bool(self.L.lancet_host_first_has_md(self.h, 1 if tumor else 0))

    def batch(self, w_begin: int, w_end: int, opts: LancetHostOpts, pack_params=None):
        """Windows [w_begin, w_end) of the tiling -> (batch, tiled indices of the windows kept).  Arrays are copied out.
        pack_params (abi.LancetParams): lancet_host_batch_packed -- the batch comes without seq / qual (empty arrays) and a third value is
        returned, the packed reads as a dict of arrays (rinfo, base_woff, good_woff, bases, good) for Engine.upload_packed."""
        cb = abi.LancetWindowBatch()
        kept = (C.c_int32 * max(1, w_end - w_begin))()
        nk = C.c_int32()
        pk = abi.LancetPackedReads()
        if pack_params is not None:
            rc = self.L.lancet_host_batch_packed(self.h, w_begin, w_end, C.byref(opts), C.byref(pack_params), C.byref(cb), C.byref(pk), kept, C.byref(nk))
        else:
            rc = self.L.lancet_host_batch(self.h, w_begin, w_end, C.byref(opts), C.byref(cb), kept, C.byref(nk))
        if rc != 0:
            raise engine.EngineError(self.L.lancet_host_last_error(self.h).decode())
        n = cb.n_windows

        def arr(ptr, count, dt):
            if count == 0:
                return np.zeros(0, dtype=dt)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(count * np.dtype(dt).itemsize,)).view(dt).copy()
        read_begin = arr(cb.read_begin, n + 1, np.uint32)
        R = int(read_begin[-1])
        seq_off = arr(cb.seq_off, R + 1, np.uint32)
        ref_off = arr(cb.ref_off, n + 1, np.uint32)
        nb = int(seq_off[-1])
        idx = [int(kept[i]) for i in range(nk.value)]
        hdr = [self.L.lancet_host_window_hdr(self.h, w).decode() for w in idx]
        names = self.chroms()
        chrom = [names[self.L.lancet_host_window_chrom(self.h, w)] for w in idx]
        lr = {}
        if opts.linked:
            nbx = C.c_uint32()
            names = self.L.lancet_host_bx_names(self.h, C.byref(nbx))
            lr = dict(bx_rank=arr(cb.bx_rank, R, np.uint32), hp=arr(cb.hp, R, np.uint8), bx_names=[names[i].decode() for i in range(nbx.value)])
        b = frontend.WindowBatch(
            n_windows=n, hdr=hdr, chrom=chrom, chr_id=arr(cb.chr_id, n, np.int32), ref_start=arr(cb.ref_start, n, np.int32),
            ref_off=ref_off, ref_bases=arr(cb.ref_bases, int(ref_off[-1]), np.uint8), read_begin=read_begin, seq_off=seq_off,
            seq=arr(cb.seq, 0 if pack_params is not None else nb, np.uint8), qual=arr(cb.qual, 0 if pack_params is not None else nb, np.uint8), label=arr(cb.label, R, np.uint8),
            strand=arr(cb.strand, R, np.uint8), mate=arr(cb.mate, R, np.uint8), mapped=arr(cb.mapped, R, np.uint8),
            name_rank=arr(cb.name_rank, R, np.uint32), **lr)
        if pack_params is not None:
            shared = bool(pk.read_index)                      # the reads stored once: the arrays describe pk.n_distinct distinct reads
            U = int(pk.n_distinct) if shared else R
            bw = arr(pk.base_woff, U + 1, np.uint32); gw = arr(pk.good_woff, U + 1, np.uint32)
            packed = dict(rinfo=arr(pk.rinfo, U + 1, np.uint32), base_woff=bw, good_woff=gw,
                          bases=arr(pk.bases, (int(bw[-1]) if U else 0) + 4, np.uint32), good=arr(pk.good, (int(gw[-1]) if U else 0) + 1, np.uint32),
                          min_qual_trim=int(pk.min_qual_trim), min_qual_call=int(pk.min_qual_call))
            if shared:
                packed["read_index"] = arr(pk.read_index, R, np.uint32); packed["n_distinct"] = U
            return b, idx, packed
        return b, idx
