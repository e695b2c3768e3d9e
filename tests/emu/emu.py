"""TEST INFRASTRUCTURE ONLY -- loader for tests/emu/libemu.so (host emulation build of the kernels)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from lancet_amd import abi, trace

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_LIBS = {}
FAT = [os.environ.get("LANCET_EMU_FAT") == "1"]                # True: the build with LANCET_FAT (the re-run tier's source, lanes still one after the other)


def lib():
    global _LIB
    _LIB = _LIBS.get(FAT[0])
    if _LIB is None:
        subprocess.run(["make", "-s", "-C", _HERE, "all"], check=True)
        L = C.CDLL(os.path.join(_HERE, "libemu_fat.so" if FAT[0] else "libemu.so"))
        L.lancet_emu_run.restype = C.c_void_p
        L.lancet_emu_run.argtypes = [C.POINTER(abi.LancetParams), C.POINTER(abi.LancetWindowBatch), C.c_uint32]
        for n, rt in (("n_variants", C.c_uint32), ("variants", C.POINTER(abi.LancetVariant)), ("blob", C.c_void_p),
                      ("blob_len", C.c_uint32), ("stats", C.POINTER(abi.LancetWindowStats)),
                      ("evt_len", C.POINTER(C.c_uint32)), ("evt", C.POINTER(C.c_uint32))):
            f = getattr(L, "lancet_emu_" + n)
            f.restype = rt
            f.argtypes = [C.c_void_p]
        L.lancet_emu_free.argtypes = [C.c_void_p]
        L.lancet_emu_n_prebuilt.restype = C.c_uint32
        L.lancet_emu_n_prebuilt.argtypes = [C.c_void_p]
        for f in ("lancet_emu_n_ahead_built", "lancet_emu_n_ahead_used", "lancet_emu_n_biglist", "lancet_emu_n_svc_posted", "lancet_emu_n_svc_built", "lancet_emu_n_svc_stolen", "lancet_emu_n_cmp_done"):
            getattr(L, f).restype = C.c_uint32
            getattr(L, f).argtypes = [C.c_void_p]
        _LIB = L
        _LIBS[FAT[0]] = L
    return _LIB


LAST_EVENTS = []
LAST_BIGLIST = [0]           # windows handed to the 1024-lane configuration of the LDS build kernel, last run
LAST_AHEAD = [0, 0]          # graphs built ahead at a later k / taken by the window kernel, last run
LAST_PREBUILT = [0]          # windows of the last run whose first graph came from the LDS build kernel
LAST_CMP = [0]               # windows of the last run whose first compress (and markRefEnds) the build kernel did
LAST_SVC = [0, 0, 0]         # build service, last run: requests posted by suspended windows / served in LDS / taken back (general build)


def run(batch, params=None, evt_cap: int = 0):
    """Returns (variants sorted by (window, seq), stats, trace text)."""
    L = lib()
    p = params or abi.default_params()
    cb = abi.batch_to_c(batch)
    h = L.lancet_emu_run(C.byref(p), C.byref(cb), evt_cap)
    try:
        n = L.lancet_emu_n_variants(h)
        LAST_PREBUILT[0] = int(L.lancet_emu_n_prebuilt(h))
        LAST_BIGLIST[0] = int(L.lancet_emu_n_biglist(h))
        LAST_AHEAD[0], LAST_AHEAD[1] = int(L.lancet_emu_n_ahead_built(h)), int(L.lancet_emu_n_ahead_used(h))
        LAST_CMP[0] = int(L.lancet_emu_n_cmp_done(h))
        LAST_SVC[0], LAST_SVC[1], LAST_SVC[2] = int(L.lancet_emu_n_svc_posted(h)), int(L.lancet_emu_n_svc_built(h)), int(L.lancet_emu_n_svc_stolen(h))
        bl = L.lancet_emu_blob_len(h)
        blob = C.string_at(L.lancet_emu_blob(h), bl) if bl else b""
        variants = abi.variants_to_py(L.lancet_emu_variants(h), n, blob)
        if p.lr_mode:
            L.lancet_emu_variants_lr.restype = C.POINTER(abi.LancetVariantLR); L.lancet_emu_variants_lr.argtypes = [C.c_void_p]
            L.lancet_emu_bx_blob.restype = C.POINTER(C.c_uint32); L.lancet_emu_bx_blob.argtypes = [C.c_void_p]
            abi.variants_lr_to_py(variants, L.lancet_emu_variants_lr(h), L.lancet_emu_bx_blob(h))
        sp = L.lancet_emu_stats(h)
        stats = [dict(status=sp[i].status, final_k=sp[i].final_k, n_builds=sp[i].n_builds, n_variants=sp[i].n_variants,
                      n_kmers=sp[i].n_kmers, max_nodes=sp[i].max_nodes) for i in range(batch.n_windows)]
        text = ""
        if evt_cap:
            lens = np.ctypeslib.as_array(L.lancet_emu_evt_len(h), shape=(batch.n_windows,))
            ev = np.ctypeslib.as_array(L.lancet_emu_evt(h), shape=(batch.n_windows * evt_cap,))
            parts = []
            LAST_EVENTS.clear()
            for w in range(batch.n_windows):
                words = ev[w * evt_cap: w * evt_cap + int(lens[w])]
                LAST_EVENTS.append(words.copy())                       # raw event words per window (for the formatter tests)
                end = int(batch.ref_start[w]) + int(batch.ref_off[w + 1] - batch.ref_off[w])
                parts.append(trace.format_window(words, w + 1, batch.hdr[w], batch.chrom[w], int(batch.ref_start[w]), end, dfs_limit=int(p.dfs_limit)))
            text = "".join(parts)
    finally:
        L.lancet_emu_free(h)
    ok = {w for w, s in enumerate(stats) if s["status"] >= 0}
    variants = sorted((v for v in variants if v["window"] in ok), key=lambda v: (v["window"], v["seq"]))
    return variants, stats, text
