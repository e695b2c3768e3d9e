"""Multi-GPU: windows are sharded across ranks (no data-path collective); the only exchange is the gather of
variant records into the VariantDB on rank 0: sizes via all_gather, payload via padded
all_gather of one uint8 tensor per rank (NCCL == RCCL on ROCm; gloo on CPU for the tests).  Records are tiny
(tens of bytes per variant), so this is latency- not bandwidth-bound on xGMI."""
from __future__ import annotations

import ctypes as C
from typing import List, Tuple

import numpy as np
import torch
import torch.distributed as dist

from . import abi


def shard_windows(n_windows: int, rank: int, world: int, chunk: int = 4096) -> List[int]:
    """Window i -> rank (i // chunk) % world : contiguous chunks keep a rank's reads sequential."""
    return [i for i in range(n_windows) if (i // chunk) % world == rank]


def pack_records(vptr, n: int, blob: bytes) -> bytes:
    raw = C.string_at(vptr, n * C.sizeof(abi.LancetVariant)) if n else b""
    hdr = np.array([n, len(blob)], dtype=np.uint64).tobytes()
    return hdr + raw + blob


def unpack_records(buf: bytes):
    n, bl = (int(x) for x in np.frombuffer(buf[:16], dtype=np.uint64))
    sz = C.sizeof(abi.LancetVariant)
    arr = (abi.LancetVariant * n).from_buffer_copy(buf[16:16 + n * sz]) if n else (abi.LancetVariant * 0)()
    blob = buf[16 + n * sz:16 + n * sz + bl]
    return arr, n, blob


def gather_bytes(payload: bytes, device: torch.device, dst: int = 0) -> List[bytes]:
    """Variable-size gather to `dst` (returns [] on other ranks)."""
    world = dist.get_world_size()
    rank = dist.get_rank()
    size = torch.tensor([len(payload)], dtype=torch.int64, device=device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, size)
    mx = int(max(int(s.item()) for s in sizes))
    buf = torch.zeros(max(mx, 1), dtype=torch.uint8, device=device)
    if payload:
        buf[:len(payload)] = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(device)
    outs = [torch.zeros(max(mx, 1), dtype=torch.uint8, device=device) for _ in range(world)]
    dist.all_gather(outs, buf)
    if rank != dst:
        return []
    return [bytes(o[:int(s.item())].cpu().numpy().tobytes()) for o, s in zip(outs, sizes)]
