"""Python binding of the C-ABI in include/lancet_engine.h (ctypes; plain pointers only).

`Engine` is the drop-in for the reference seam `Microassembler::processGraph` (reference
src/Microassembler.cc:837) applied to a whole batch of windows; `VariantDB` is the host side below it
(reference src/VariantDB.cc).  There is no CPU fallback: a missing library or a missing GPU raises."""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import numpy as np

from . import abi, trace as _trace

_LIBPATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "liblancet_engine.so")
_LIB = None

ERRORS = {0: "OK", -1: "bad argument", -2: "no HIP device", -3: "HIP error", -4: "unsupported", -5: "out of memory", -6: "bad state"}


class EngineError(RuntimeError):
    pass


def lib():
    global _LIB
    if _LIB is None:
        path = _LIBPATH
        if not os.path.exists(path):
            raise EngineError(f"{_LIBPATH} is missing: build it with `python -m lancet_amd.build` (there is no CPU fallback)")
        L = C.CDLL(path)
        L.lancet_params_default.argtypes = [C.POINTER(abi.LancetParams)]
        L.lancet_engine_create.argtypes = [C.POINTER(abi.LancetParams), C.c_int, C.POINTER(C.c_void_p)]
        L.lancet_engine_destroy.argtypes = [C.c_void_p]
        L.lancet_engine_last_error.restype = C.c_char_p
        L.lancet_engine_last_error.argtypes = [C.c_void_p]
        L.lancet_engine_upload.argtypes = [C.c_void_p, C.POINTER(abi.LancetWindowBatch)]
        L.lancet_engine_run.argtypes = [C.c_void_p]
        L.lancet_engine_submit.argtypes = [C.c_void_p]
        L.lancet_engine_submit_after.argtypes = [C.c_void_p, C.c_void_p]
        L.lancet_engine_wait.argtypes = [C.c_void_p]
        L.lancet_engine_process.argtypes = [C.c_void_p, C.POINTER(abi.LancetWindowBatch)]
        L.lancet_engine_results.argtypes = [C.c_void_p, C.POINTER(C.POINTER(abi.LancetVariant)), C.POINTER(C.c_uint32),
                                            C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.POINTER(C.POINTER(abi.LancetWindowStats))]
        L.lancet_engine_last_timing.argtypes = [C.c_void_p, C.POINTER(C.c_float * 2)]
        L.lancet_engine_results_lr.argtypes = [C.c_void_p, C.POINTER(C.POINTER(abi.LancetVariantLR)), C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.c_uint32)]
        L.lancet_engine_set_trace.argtypes = [C.c_void_p, C.c_uint32]
        L.lancet_engine_trace.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.c_uint32)]
        L.lancet_engine_rerun_count.argtypes = [C.c_void_p]
        L.lancet_engine_prebuilt_count.argtypes = [C.c_void_p]
        L.lancet_engine_ahead_counts.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.lancet_engine_svc_counts.argtypes = [C.c_void_p, C.POINTER(C.c_uint32 * 4)]
        L.lancet_engine_build_phase_times.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_uint64))]
        L.lancet_engine_kernel_times.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int]
        L.lancet_engine_kernel_name.restype = C.c_char_p
        L.lancet_engine_kernel_name.argtypes = [C.c_int]
        L.lancet_engine_geometry.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_uint64)]
        L.lancet_debug_align.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
        L.lancet_debug_align_mode.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int]
        L.lancet_engine_phase_times.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_uint64))]
        L.lancet_filters_default.argtypes = [C.POINTER(abi.LancetFilters)]
        L.lancet_vdb_create.restype = C.c_void_p
        L.lancet_vdb_create.argtypes = [C.POINTER(abi.LancetFilters)]
        L.lancet_vdb_destroy.argtypes = [C.c_void_p]
        L.lancet_vdb_add.argtypes = [C.c_void_p, C.POINTER(abi.LancetVariant), C.c_uint32, C.c_char_p, C.POINTER(C.c_char_p), C.c_int32]
        L.lancet_vdb_add_lr.argtypes = [C.c_void_p, C.POINTER(abi.LancetVariant), C.POINTER(abi.LancetVariantLR), C.c_uint32, C.c_char_p,
                                        C.POINTER(C.c_uint32), C.POINTER(C.c_char_p), C.c_uint32, C.POINTER(C.c_char_p), C.c_int32]
        L.lancet_vdb_keys.argtypes = [C.POINTER(abi.LancetVariant), C.c_uint32, C.c_char_p, C.POINTER(C.c_char_p), C.c_int32, C.c_void_p]
        L.lancet_vdb_add_keyed.argtypes = [C.c_void_p, C.POINTER(abi.LancetVariant), C.POINTER(abi.LancetVariantLR), C.c_void_p, C.c_uint32, C.c_char_p,
                                           C.POINTER(C.c_uint32), C.POINTER(C.c_char_p), C.c_uint32, C.POINTER(C.c_char_p), C.c_int32]
        L.lancet_vdb_reduce.argtypes = [C.POINTER(abi.LancetVariant), C.c_void_p, C.c_uint32, C.c_void_p]
        L.lancet_vdb_size.restype = C.c_uint32
        L.lancet_vdb_size.argtypes = [C.c_void_p]
        L.lancet_vdb_vcf.restype = C.c_void_p
        L.lancet_vdb_vcf.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p]
        L.lancet_free.argtypes = [C.c_void_p]
        L.lancet_trace_format.restype = C.c_void_p
        L.lancet_trace_format.argtypes = [C.POINTER(C.c_uint32), C.c_uint32, C.c_int32, C.c_char_p, C.c_char_p, C.c_int32, C.c_int32, C.c_int32]
        _LIB = L
    return _LIB


class Engine:
    def __init__(self, params: Optional[abi.LancetParams] = None, device: int = 0, trace_words: int = 0):
        self.L = lib()
        self.params = params or abi.default_params()
        h = C.c_void_p()
        rc = self.L.lancet_engine_create(C.byref(self.params), device, C.byref(h))
        if rc != 0:
            raise EngineError(f"lancet_engine_create failed: {ERRORS.get(rc, rc)}")
        self.h = h
        self._batch = None
        if trace_words:
            self.L.lancet_engine_set_trace(self.h, trace_words)

    def _chk(self, rc):
        if rc != 0:
            raise EngineError(f"{ERRORS.get(rc, rc)}: {self.L.lancet_engine_last_error(self.h).decode()}")

    def upload(self, batch) -> None:
        self._batch = batch
        cb = abi.batch_to_c(batch)
        self._chk(self.L.lancet_engine_upload(self.h, C.byref(cb)))

    def upload_packed(self, batch, packed) -> None:
        """lancet_engine_upload_packed: `batch` without bases / qualities (host.NativeHost.batch(..., pack_params=...)), `packed` the dict of
        arrays that call returned (packed with THIS engine's parameters)."""
        self._batch = batch
        cb = abi.batch_to_c(batch)
        keep = {k: np.ascontiguousarray(packed[k], dtype=np.uint32) for k in ("rinfo", "base_woff", "good_woff", "bases", "good")}
        pk = abi.LancetPackedReads(C.sizeof(abi.LancetPackedReads), 0, *[keep[k].ctypes.data_as(C.POINTER(C.c_uint32)) for k in ("rinfo", "base_woff", "good_woff", "bases", "good")])
        # the thresholds the producer packed with (host.NativeHost.batch records them): required -- a dict without them would claim this
        # engine's own and walk past the C side's check against reads packed with other thresholds
        for k in ("min_qual_trim", "min_qual_call"):
            if k not in packed:
                raise EngineError(f"upload_packed: the packed dict has no '{k}' (the thresholds the reads were trimmed / masked with)")
        pk.min_qual_trim = int(packed["min_qual_trim"]); pk.min_qual_call = int(packed["min_qual_call"])
        if packed.get("read_index") is not None:              # the reads stored once per batch (include/lancet_engine.h)
            keep["read_index"] = np.ascontiguousarray(packed["read_index"], dtype=np.uint32)
            pk.read_index = keep["read_index"].ctypes.data_as(C.POINTER(C.c_uint32)); pk.n_distinct = int(packed["n_distinct"])
        self._packed_keep = keep
        self.L.lancet_engine_upload_packed.restype = C.c_int
        self.L.lancet_engine_upload_packed.argtypes = [C.c_void_p, C.POINTER(abi.LancetWindowBatch), C.POINTER(abi.LancetPackedReads)]
        self._chk(self.L.lancet_engine_upload_packed(self.h, C.byref(cb), C.byref(pk)))

    def run(self) -> None:
        self._chk(self.L.lancet_engine_run(self.h))

    def submit(self, after: "Engine | None" = None) -> None:
        """Launch the kernels of the uploaded batch and return at once (collect with wait()).  `after`: another engine on the same
        device whose batch is in flight -- this batch's kernels then start when that one's build kernel is through."""
        if after is not None and after is not self:
            self._chk(self.L.lancet_engine_submit_after(self.h, after.h))
        else:
            self._chk(self.L.lancet_engine_submit(self.h))

    def wait(self) -> None:
        self._chk(self.L.lancet_engine_wait(self.h))

    def process(self, batch):
        self.upload(batch)
        self.run()
        return self.results()

    def raw_results(self):
        vp = C.POINTER(abi.LancetVariant)()
        n = C.c_uint32()
        blob = C.c_void_p()
        bl = C.c_uint32()
        sp = C.POINTER(abi.LancetWindowStats)()
        self._chk(self.L.lancet_engine_results(self.h, C.byref(vp), C.byref(n), C.byref(blob), C.byref(bl), C.byref(sp)))
        return vp, n.value, (C.string_at(blob, bl.value) if bl.value else b""), sp

    def raw_results_lr(self):
        """(lancet_variant_lr*, bx_blob*, bx_blob_len) of the last run; the first is NULL unless lr_mode."""
        lp = C.POINTER(abi.LancetVariantLR)()
        bp = C.POINTER(C.c_uint32)()
        bl = C.c_uint32()
        self._chk(self.L.lancet_engine_results_lr(self.h, C.byref(lp), C.byref(bp), C.byref(bl)))
        return lp, bp, bl.value

    def results(self):
        vp, n, blob, sp = self.raw_results()
        variants = abi.variants_to_py(vp, n, blob)
        if self.params.lr_mode:
            lp, bp, _ = self.raw_results_lr()
            abi.variants_lr_to_py(variants, lp, bp)
        nw = self._batch.n_windows
        stats = [dict(status=sp[i].status, final_k=sp[i].final_k, n_builds=sp[i].n_builds, n_variants=sp[i].n_variants,
                      n_kmers=sp[i].n_kmers, max_nodes=sp[i].max_nodes, sum_nodes=sp[i].sum_nodes) for i in range(nw)]
        return variants, stats

    def timing_ms(self):
        t = (C.c_float * 2)()
        self._chk(self.L.lancet_engine_last_timing(self.h, C.byref(t)))
        return float(t[0]), float(t[1])

    def kernel_names(self):
        out = []
        i = 0
        while True:
            n = self.L.lancet_engine_kernel_name(i)
            if not n:
                return out
            out.append(n.decode()); i += 1

    def kernel_times(self):
        """HIP-event duration (ms) of every kernel of the last run, in kernel_names() order."""
        buf = (C.c_float * 8)()
        n = self.L.lancet_engine_kernel_times(self.h, buf, 8)
        if n < 0:
            self._chk(n)
        return [float(buf[i]) for i in range(n)]

    def build_phase_times(self):
        """Workgroup-seconds of the LDS build kernel per phase (profiling aid)."""
        p = C.POINTER(C.c_uint64)()
        self._chk(self.L.lancet_engine_build_phase_times(self.h, C.byref(p)))
        return [p[i] * 1e-8 for i in range(16)]

    def prebuilt_count(self) -> int:
        return int(self.L.lancet_engine_prebuilt_count(self.h))

    def pre_headers(self):
        """test hook: per window of the last run (built in LDS, K, nodes, table order + components came along) from the hand-off headers"""
        import numpy as np
        nw = self._batch.n_windows
        out = np.zeros(8 * max(1, nw), dtype=np.uint32)
        self.L.lancet_debug_pre_headers.argtypes = [C.c_void_p, C.c_void_p]
        if self.L.lancet_debug_pre_headers(self.h, out.ctypes.data) != 0:
            return []
        o = out.reshape(-1, 8)[:nw]
        return [dict(built=int(r[0] & 0xFF) == 1, K=int(r[1]), nodes=int(r[3]), order=bool(r[0] >> 16 & 1)) for r in o]

    def ahead_counts(self):
        """(graphs the build kernel built ahead at a later k, how many the window kernel took)"""
        b, u = C.c_int32(), C.c_int32()
        self._chk(self.L.lancet_engine_ahead_counts(self.h, C.byref(b), C.byref(u)))
        return b.value, u.value

    def svc_counts(self):
        """Build service of the last run: (requests posted, served in LDS, not buildable there, taken back by the window kernel)"""
        a = (C.c_uint32 * 4)()
        self._chk(self.L.lancet_engine_svc_counts(self.h, C.byref(a)))
        return tuple(int(x) for x in a)

    def rerun_count(self) -> int:
        return int(self.L.lancet_engine_rerun_count(self.h))

    def geometry(self):
        s = C.c_int32()
        b = C.c_uint64()
        self._chk(self.L.lancet_engine_geometry(self.h, C.byref(s), C.byref(b)))
        return s.value, b.value

    def trace_text(self) -> str:
        lp = C.POINTER(C.c_uint32)()
        ep = C.POINTER(C.c_uint32)()
        wpw = C.c_uint32()
        self._chk(self.L.lancet_engine_trace(self.h, C.byref(lp), C.byref(ep), C.byref(wpw)))
        if not wpw.value:
            return ""
        b = self._batch
        lens = np.ctypeslib.as_array(lp, shape=(b.n_windows,))
        ev = np.ctypeslib.as_array(ep, shape=(b.n_windows * wpw.value,))
        parts = []
        for w in range(b.n_windows):
            words = ev[w * wpw.value: w * wpw.value + int(lens[w])]
            end = int(b.ref_start[w]) + int(b.ref_off[w + 1] - b.ref_off[w])
            parts.append(_trace.format_window(words, w + 1, b.hdr[w], b.chrom[w], int(b.ref_start[w]), end, dfs_limit=int(self.params.dfs_limit)))
        return "".join(parts)

    def phase_times(self):
        """[n_windows, 16] array of per-phase times in seconds (profiling aid)."""
        p = C.POINTER(C.c_uint64)()
        self._chk(self.L.lancet_engine_phase_times(self.h, C.byref(p)))
        return np.ctypeslib.as_array(p, shape=(self._batch.n_windows, 16)).astype(np.float64) * 1e-8

    def debug_align(self, s: str, t: str, mode: int = 0):
        """Test hook: the device global_align_aff on one pair of strings.  mode 0: banded matrix with the full one as fall-back (what
        the window kernel runs), 1: full matrix only, 2: band only -- returns None when the band could not be certified."""
        cap = len(s) + len(t) + 8
        a = C.create_string_buffer(cap)
        b = C.create_string_buffer(cap)
        rc = self.L.lancet_debug_align_mode(self.h, s.encode(), t.encode(), a, b, cap, mode)
        if rc == -6 and mode == 2:
            return None
        self._chk(rc)
        return a.value.decode(), b.value.decode()

    def close(self):
        if getattr(self, "h", None):
            self.L.lancet_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def record_keys(vptr, n: int, blob: bytes, chr_names: Sequence[str]):
    """lancet_vdb_keys: the 32-byte addVar key (sha256 of Variant_t::getSignature) of each of n records -> uint8[n, 32]."""
    import numpy as np
    keys = np.zeros((max(1, n), 32), dtype=np.uint8)
    names = abi.c_string_array(list(chr_names))
    rc = lib().lancet_vdb_keys(vptr, n, blob, names, len(chr_names), keys.ctypes.data)
    if rc != 0:
        raise EngineError(f"lancet_vdb_keys: {ERRORS.get(rc, rc)}")
    return keys[:n]


def records_that_matter(vptr, keys, n: int):
    """lancet_vdb_reduce: bool[n], the records of a keyed stream (in adding order) that can change a VariantDB."""
    import numpy as np
    keep = np.zeros(max(1, n), dtype=np.uint8)
    rc = lib().lancet_vdb_reduce(vptr, keys.ctypes.data, n, keep.ctypes.data)
    if rc != 0:
        raise EngineError(f"lancet_vdb_reduce: {ERRORS.get(rc, rc)}")
    return keep[:n].astype(bool)


class VariantDB:
    """reference src/VariantDB.cc: addVar + printToVCF (host)."""

    def __init__(self, filters: Optional[abi.LancetFilters] = None):
        self.L = lib()
        self.h = C.c_void_p(self.L.lancet_vdb_create(C.byref(filters) if filters is not None else None))

    def add_raw(self, vptr, n: int, blob: bytes, chr_names: Sequence[str]) -> None:
        names = abi.c_string_array(list(chr_names))
        rc = self.L.lancet_vdb_add(self.h, vptr, n, blob, names, len(chr_names))
        if rc != 0:
            raise EngineError(f"lancet_vdb_add: {ERRORS.get(rc, rc)}")

    def add_raw_lr(self, vptr, lrptr, n: int, blob: bytes, bx_blob, bx_names: Sequence[str], chr_names: Sequence[str]) -> None:
        names = abi.c_string_array(list(chr_names))
        bxn = abi.c_string_array(list(bx_names))
        rc = self.L.lancet_vdb_add_lr(self.h, vptr, lrptr, n, blob, bx_blob, bxn, len(bx_names), names, len(chr_names))
        if rc != 0:
            raise EngineError(f"lancet_vdb_add_lr: {ERRORS.get(rc, rc)}")

    def add_raw_keyed(self, vptr, lrptr, keys, n: int, blob: bytes, bx_blob, bx_names: Optional[Sequence[str]], chr_names: Sequence[str]) -> None:
        """The records with their addVar keys (record_keys, computed where the records were made): rank 0 of a multi-GPU run only inserts.
        keys: numpy uint8 array of 32 * n bytes; lrptr / bx_blob / bx_names None without --linked-reads."""
        names = abi.c_string_array(list(chr_names))
        bxn = abi.c_string_array(list(bx_names)) if bx_names is not None else None
        rc = self.L.lancet_vdb_add_keyed(self.h, vptr, lrptr, keys.ctypes.data, n, blob, bx_blob, bxn, len(bx_names) if bx_names is not None else 0,
                                         names, len(chr_names))
        if rc != 0:
            raise EngineError(f"lancet_vdb_add_keyed: {ERRORS.get(rc, rc)}")

    def add_records(self, records: List[dict], chr_names: Sequence[str], bx_names: Optional[Sequence[str]] = None) -> None:
        """records: dicts as produced by abi.variants_to_py (any source); with bx_names: linked-read records
        (abi.variants_lr_to_py) for a --linked-reads database."""
        arr = (abi.LancetVariant * len(records))()
        blob = bytearray()
        if bx_names is not None:
            lr = (abi.LancetVariantLR * max(1, len(records)))()
            ids: List[int] = []
            for i, r in enumerate(records):
                for q in range(12):
                    lr[i].hp[q] = r["hp"][q]
                for q in range(4):
                    lr[i].bx_off[q], lr[i].bx_len[q] = len(ids), len(r["bx"][q])
                    ids += list(r["bx"][q])
            bxb = (C.c_uint32 * max(1, len(ids)))(*ids)
        for i, r in enumerate(records):
            v = arr[i]
            v.window, v.seq_in_window, v.chr_id, v.pos = r["window"], r["seq"], r["chr_id"], r["pos"]
            v.code, v.prev_bp_ref, v.prev_bp_alt, v.kmer = ord(r["code"]), ord(r["prev_bp_ref"]), ord(r["prev_bp_alt"]), r["kmer"]
            for q in range(8):
                v.cov[q] = r["cov"][q]
            v.ref_off, v.ref_len = len(blob), len(r["ref"]); blob += r["ref"].encode()
            v.alt_off, v.alt_len = len(blob), len(r["alt"]); blob += r["alt"].encode()
            v.str_off, v.str_len = len(blob), len(r["str"]); blob += r["str"].encode()
        if bx_names is not None:
            self.add_raw_lr(arr, lr, len(records), bytes(blob) + b"\0", bxb, bx_names, chr_names)
        else:
            self.add_raw(arr, len(records), bytes(blob) + b"\0", chr_names)

    def size(self) -> int:
        return int(self.L.lancet_vdb_size(self.h))

    def vcf(self, version: Optional[str] = None, cmdline: Optional[str] = None, reference: Optional[str] = None,
            date_line: Optional[str] = None, sample_normal: str = "NORMAL", sample_tumor: str = "TUMOR") -> str:
        enc = lambda s: s.encode() if s is not None else None
        p = self.L.lancet_vdb_vcf(self.h, enc(version), enc(cmdline), enc(reference), enc(date_line), enc(sample_normal), enc(sample_tumor))
        if not p:
            raise EngineError("lancet_vdb_vcf failed")
        try:
            return C.string_at(p).decode()
        finally:
            self.L.lancet_free(p)

    def close(self):
        if getattr(self, "h", None):
            self.L.lancet_vdb_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
