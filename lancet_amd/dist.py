"""Multi-GPU: windows are sharded across ranks (independent units, no data-path collective); the only exchange is the
gather of the variant records into the VariantDB on rank 0.

  sizes    one all_gather of 8 bytes per rank
  payload  point-to-point send / recv to rank 0 only (RCCL over xGMI on the GPUs, gloo in the CPU tests): a gatherv

A payload carries everything `Variant_t`'s constructor gets (reference src/Graph.cc:1166-1188): the packed
`lancet_variant` records, their string blob and -- with --linked-reads -- the `lancet_variant_lr` records, the barcode ids
of their four barcode sets and the names of the barcodes used (ids are ranks inside ONE batch, so they travel as names).
Rank 0 replays the union in (global window index, emission order): `addVar` keeps the first record on ties
(src/VariantDB.cc:51), so the VCF must not depend on how the windows were dealt out.
Every rank also sends the 32-byte addVar key of each of its records (lancet_vdb_keys: Variant_t's normalisation, getSignature and
sha256, src/VariantDB.cc:36-40), so that rank 0 -- the serial part of an N-rank step -- only inserts (lancet_vdb_add_keyed)."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist

from . import abi

_VDT = np.dtype(abi.LancetVariant)
_LDT = np.dtype(abi.LancetVariantLR)
_HDR = 8          # u64 words


def shard_windows(n_windows: int, rank: int, world: int, chunk: int = 4096) -> List[int]:
    """Window i -> rank (i // chunk) % world : contiguous chunks keep a rank's reads sequential."""
    return [i for i in range(n_windows) if (i // chunk) % world == rank]


def records_in_replay_order(recs: np.ndarray, window_index: Optional[np.ndarray]) -> bool:
    """The per-key reduction (lancet_vdb_reduce) keeps, per key, the FIRST record and the first of the largest total coverage -- "first"
    in the order rank 0 will replay them (global window, emission).  It may only run on a rank whose records are already in that order:
    (window, seq_in_window) non-decreasing in the array, and the local -> global window map strictly increasing.  Otherwise the caller
    sends every record (rank 0 sorts; reference src/VariantDB.cc:28-91 then sees what one process would have seen)."""
    if len(recs) > 1:
        w, q = recs["window"], recs["seq_in_window"]
        if not bool(np.all((w[1:] > w[:-1]) | ((w[1:] == w[:-1]) & (q[1:] >= q[:-1])))):
            return False
    if window_index is not None:
        wi = np.asarray(window_index, dtype=np.int64)
        if len(wi) > 1 and not bool(np.all(wi[1:] > wi[:-1])):
            return False
    return True


def pack_records(vptr, n: int, blob: bytes, lrptr=None, bx_blob=None, bx_names: Optional[Sequence[str]] = None, *,
                 chr_names: Sequence[str], window_index: Optional[np.ndarray] = None, with_keys: bool = True, reduce: bool = True) -> bytes:
    """One rank's records as bytes.  chr_names[chr_id] = the contig names of the rank's batch (required: records carry ids);
    window_index[w] = global index of the rank's window w (default: identity); with_keys: the records' addVar keys travel along;
    reduce (with keys): only the records that can change the VariantDB travel (lancet_vdb_reduce: per key the first record and the
    first of the largest total coverage) -- the rank's records must be in (window, emission) order, as the engine returns them."""
    if not chr_names:
        raise ValueError("pack_records: chr_names is required")
    keys = b""; keep = None
    recs = np.frombuffer(C.string_at(vptr, n * _VDT.itemsize), dtype=_VDT).copy() if n else np.zeros(0, dtype=_VDT)
    if with_keys and n:
        from . import engine
        k = engine.record_keys(vptr, n, blob, chr_names)
        if reduce and records_in_replay_order(recs, window_index):
            keep = engine.records_that_matter(vptr, k, n)
            k = np.ascontiguousarray(k[keep])
        keys = k.tobytes()
    n_all = n
    if keep is not None:
        recs = np.ascontiguousarray(recs[keep]); n = len(recs)
    if window_index is not None and n:
        recs["window"] = np.asarray(window_index, dtype=np.int64)[recs["window"]]
    lr = b""; ids = np.zeros(0, dtype=np.uint32); names = b""
    has_lr = lrptr is not None and bool(lrptr)
    if has_lr:
        l = np.frombuffer(C.string_at(lrptr, n_all * _LDT.itemsize), dtype=_LDT).copy() if n_all else np.zeros(0, dtype=_LDT)
        nbx = int((l["bx_off"] + l["bx_len"]).max()) if n_all else 0
        raw = np.ctypeslib.as_array(bx_blob, shape=(nbx,)).astype(np.uint32) if nbx else np.zeros(0, dtype=np.uint32)
        used, inv = np.unique(raw, return_inverse=True)                       # only the barcodes that occur travel
        ids = inv.astype(np.uint32)
        names = "\0".join(bx_names[int(u)] for u in used).encode()
        if keep is not None:
            l = np.ascontiguousarray(l[keep])
        lr = l.tobytes()
    chrs = "\0".join(chr_names).encode()
    hdr = np.array([n, len(blob), 1 if has_lr else 0, len(ids), len(names), len(chrs), len(keys), 0], dtype=np.uint64)
    return hdr.tobytes() + recs.tobytes() + blob + lr + ids.tobytes() + names + chrs + keys


def unpack_records(buf: bytes, copy: bool = True) -> dict:
    """copy=False returns read-only views into `buf` for the record arrays (merge_into_vdb copies them once, into place)."""
    n, bl, has_lr, nbx, nl, cl, kl, _ = (int(x) for x in np.frombuffer(buf[:8 * _HDR], dtype=np.uint64))
    o = 8 * _HDR
    own = (lambda a: a.copy()) if copy else (lambda a: a)
    recs = own(np.frombuffer(buf, dtype=_VDT, count=n, offset=o)); o += n * _VDT.itemsize
    blob = buf[o:o + bl]; o += bl
    lr = None; ids = None; names: List[str] = []
    if has_lr:
        lr = own(np.frombuffer(buf, dtype=_LDT, count=n, offset=o)); o += n * _LDT.itemsize
        ids = own(np.frombuffer(buf, dtype=np.uint32, count=nbx, offset=o)); o += 4 * nbx
        names = buf[o:o + nl].decode().split("\0") if nl else []; o += nl
    chrs = buf[o:o + cl].decode().split("\0") if cl else []; o += cl
    keys = own(np.frombuffer(buf, dtype=np.uint8, count=kl, offset=o).reshape(-1, 32)) if kl else None
    return dict(n=n, recs=recs, blob=blob, lr=lr, bx_ids=ids, bx_names=names, chr_names=chrs, keys=keys)


def gather_bytes(payload: bytes, device: torch.device, dst: int = 0) -> List[bytes]:
    """Variable-size gather to `dst`: sizes by all_gather, payloads by send / recv to `dst` only.  Returns [] elsewhere."""
    world = dist.get_world_size()
    rank = dist.get_rank()
    size = torch.tensor([len(payload)], dtype=torch.int64, device=device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, size)
    sizes = [int(s.item()) for s in sizes]
    if rank != dst:
        if len(payload):
            dist.send(torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(device), dst=dst)
        return []
    # one receive buffer for all ranks, every irecv posted before any wait, ONE device -> host copy at the end
    offs = [0] * (world + 1)
    for r in range(world):
        offs[r + 1] = offs[r] + (0 if r == dst else sizes[r])
    buf = torch.empty(max(1, offs[world]), dtype=torch.uint8, device=device)
    reqs = [dist.irecv(buf[offs[r]:offs[r + 1]], src=r) for r in range(world) if r != dst and sizes[r]]
    for q in reqs:
        q.wait()
    host = buf.cpu().numpy() if offs[world] else np.zeros(0, dtype=np.uint8)
    return [payload if r == dst else host[offs[r]:offs[r + 1]].tobytes() for r in range(world)]


def merge_into_vdb(parts: Sequence[bytes], db) -> int:
    """Rank 0: replays the records of every rank into `db` (lancet_amd.engine.VariantDB) in (global window, emission)
    order -- the order a single process would have produced them in.  Returns the number of records added."""
    ps = [unpack_records(b, copy=False) for b in parts if b]
    ps = [p for p in ps if p["n"]]
    if not ps:
        return 0
    chr_names: List[str] = []
    bx_names: List[str] = []
    bx_index = {}
    ids_all, blobs = [], []
    blob_base = 0; id_base = 0; o = 0
    lr_mode = any(p["lr"] is not None for p in ps)
    if lr_mode and not all(p["lr"] is not None for p in ps):
        raise ValueError("merge_into_vdb: some ranks sent linked-read records and some did not (one run is either --linked-reads or not)")
    total = sum(p["n"] for p in ps)
    keyed = all(p["keys"] is not None for p in ps)
    keys = np.empty((total, 32), dtype=np.uint8) if keyed else None
    recs = np.empty(total, dtype=_VDT)                        # every part is copied once, straight into its place
    lr = np.empty(total, dtype=_LDT) if lr_mode else None
    for p in ps:
        r = recs[o:o + p["n"]]
        r[:] = p["recs"]
        if keyed:
            keys[o:o + p["n"]] = p["keys"]
        cmap = np.zeros(max(1, len(p["chr_names"])), dtype=np.int32)
        for i, c in enumerate(p["chr_names"]):
            if c not in chr_names:
                chr_names.append(c)
            cmap[i] = chr_names.index(c)
        r["chr_id"] = cmap[r["chr_id"]]
        for f in ("ref_off", "alt_off", "str_off"):
            r[f] += blob_base
        blob_base += len(p["blob"]); blobs.append(p["blob"])
        if lr_mode:
            l = lr[o:o + p["n"]]
            l[:] = p["lr"]
            gmap = np.zeros(max(1, len(p["bx_names"])), dtype=np.uint32)
            for i, nm in enumerate(p["bx_names"]):
                if nm not in bx_index:
                    bx_index[nm] = len(bx_names); bx_names.append(nm)
                gmap[i] = bx_index[nm]
            l["bx_off"] += id_base
            id_base += len(p["bx_ids"])
            ids_all.append(gmap[p["bx_ids"]] if len(p["bx_ids"]) else np.zeros(0, dtype=np.uint32))
        o += p["n"]
    # ranks that hold contiguous runs of windows arrive already in (window, emission) order: check before sorting
    w, q = recs["window"], recs["seq_in_window"]
    if total > 1 and not bool(np.all((w[1:] > w[:-1]) | ((w[1:] == w[:-1]) & (q[1:] >= q[:-1])))):
        order = np.lexsort((q, w))
        recs = np.ascontiguousarray(recs[order])
        if keyed:
            keys = np.ascontiguousarray(keys[order])
        if lr_mode:
            lr = np.ascontiguousarray(lr[order])
    blob = b"".join(blobs) + b"\0"
    vptr = recs.ctypes.data_as(C.POINTER(abi.LancetVariant))
    if lr_mode:
        # ids become ranks in the union's name order: a set stays sorted by name (std::set<string>) because every rank's ids
        # were ranks by name already
        rank_of = np.argsort(np.argsort(np.array(bx_names, dtype=object))) if bx_names else np.zeros(0, dtype=np.int64)
        sorted_names = sorted(bx_names)
        ids = rank_of[np.concatenate(ids_all)].astype(np.uint32) if id_base else np.zeros(1, dtype=np.uint32)
        ids = np.ascontiguousarray(ids)
        if keyed:
            db.add_raw_keyed(vptr, lr.ctypes.data_as(C.POINTER(abi.LancetVariantLR)), keys, len(recs), blob,
                             ids.ctypes.data_as(C.POINTER(C.c_uint32)), sorted_names, chr_names)
        else:
            db.add_raw_lr(vptr, lr.ctypes.data_as(C.POINTER(abi.LancetVariantLR)), len(recs), blob,
                          ids.ctypes.data_as(C.POINTER(C.c_uint32)), sorted_names, chr_names)
    elif keyed:
        db.add_raw_keyed(vptr, None, keys, len(recs), blob, None, None, chr_names)
    else:
        db.add_raw(vptr, len(recs), blob, chr_names)
    return len(recs)
