"""N>1 path on CPU: two gloo ranks shard the windows of a reference golden, pack their variant records (with the linked-read
annotations when the case has them) and send them to rank 0, which replays the union in window order into the VariantDB:
the VCF must equal the reference's single-process VCF."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import golden_util as gu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def records_to_c(recs, lr):
    """dict records (abi.variants_to_py / variants_lr_to_py) -> the arrays the engine hands out"""
    import ctypes as C
    from lancet_amd import abi
    arr = (abi.LancetVariant * max(1, len(recs)))()
    blob = bytearray()
    lra = (abi.LancetVariantLR * max(1, len(recs)))() if lr else None
    ids = []
    for i, r in enumerate(recs):
        x = arr[i]
        x.window, x.seq_in_window, x.chr_id, x.pos = r["window"], r["seq"], r["chr_id"], r["pos"]
        x.code, x.prev_bp_ref, x.prev_bp_alt, x.kmer = ord(r["code"]), ord(r["prev_bp_ref"]), ord(r["prev_bp_alt"]), r["kmer"]
        for k in range(8):
            x.cov[k] = r["cov"][k]
        x.ref_off, x.ref_len = len(blob), len(r["ref"]); blob += r["ref"].encode()
        x.alt_off, x.alt_len = len(blob), len(r["alt"]); blob += r["alt"].encode()
        x.str_off, x.str_len = len(blob), len(r["str"]); blob += r["str"].encode()
        if lr:
            for q in range(12):
                lra[i].hp[q] = r["hp"][q]
            for q in range(4):
                lra[i].bx_off[q], lra[i].bx_len[q] = len(ids), len(r["bx"][q])
                ids += list(r["bx"][q])
    bxb = (C.c_uint32 * max(1, len(ids)))(*ids) if lr else None
    return arr, bytes(blob), lra, bxb


def _worker(rank, world, port, case, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lancet_amd import dist as ldist, engine, workload
    from oracle import oracle
    meta, batch, kept, (min_k, max_k) = gu.case_batch(case)
    lr = gu.case_lr(meta)
    p = gu.params(meta)
    mine = ldist.shard_windows(batch.n_windows, rank, world, chunk=3)
    # this rank's windows as ONE batch of its own (local window indices 0..len(mine)-1); records by the oracle (no GPU here)
    recs = []
    for li, w in enumerate(mine):
        sub = workload.sub_batch(batch, w, w + 1)
        v, _, _ = oracle.run(sub, p)
        for r in v:
            r["window"] = li
            recs.append(r)
    arr, blob, lra, bxb = records_to_c(recs, lr)
    import time
    t0 = time.perf_counter()
    payload = ldist.pack_records(arr, len(recs), blob, lra, bxb, batch.bx_names, chr_names=["chr22"], window_index=mine)
    t1 = time.perf_counter()
    parts = ldist.gather_bytes(payload, torch.device("cpu"))
    t2 = time.perf_counter()
    if rank == 0:
        db = engine.VariantDB()
        n = ldist.merge_into_vdb(parts, db)
        # (a printed figure, pytest -s: what a step's gather costs the thread that runs it -- in bench.py a communication thread, not the one that submits kernels)
        print(f"[gloo x{world}] {case}: {len(recs)} records on rank 0, pack_records {1e3 * (t1 - t0):.2f} ms, gather_bytes {1e3 * (t2 - t1):.2f} ms, "
              f"merge_into_vdb {1e3 * (time.perf_counter() - t2):.2f} ms", file=sys.stderr)
        q.put((n, db.vcf()))
    else:
        assert parts == []
    dist.destroy_process_group()


def _run_ranks(world, case):
    from lancet_amd import build
    build.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + 7 * world
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    n, vcf = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return n, vcf


@pytest.mark.parametrize("case", ["tile30", "lr30"])
def test_two_rank_gather_reproduces_single_process_vcf(case):
    n, vcf = _run_ranks(2, case)
    assert n > 0
    assert vcf == gu.golden_vcf(case)


@pytest.mark.parametrize("case", ["tile30", "lr30"])
def test_eight_rank_gather_with_keys_made_on_the_sending_ranks(case):
    """World size 8, the windows dealt out three at a time (every rank holds interleaved runs: the records reach rank 0 out of window
    order, and a variant's overlapping windows lie on different ranks).  Every rank keys its own records (lancet_vdb_keys) and sends
    only those that can change the database (lancet_vdb_reduce); rank 0 re-sorts and inserts: the reference's single-process VCF."""
    n, vcf = _run_ranks(8, case)
    assert n > 0
    assert vcf == gu.golden_vcf(case)


def test_unordered_records_are_sent_whole():
    """pack_records' per-key reduction keeps "the first" record of a key -- first in the order rank 0 replays them.  A rank whose records are
    not in (window, emission) order, or whose local -> global window map is not increasing, must send every record."""
    import numpy as np
    from lancet_amd import dist as ldist
    dt = np.dtype(ldist.abi.LancetVariant)
    recs = np.zeros(4, dtype=dt)
    recs["window"] = [0, 0, 1, 2]; recs["seq_in_window"] = [0, 1, 0, 0]
    assert ldist.records_in_replay_order(recs, None) and ldist.records_in_replay_order(recs, np.array([5, 9, 12]))
    assert not ldist.records_in_replay_order(recs, np.array([5, 4, 12]))            # the window map goes backwards
    recs["window"] = [0, 1, 0, 2]
    assert not ldist.records_in_replay_order(recs, None)                              # the records do
    recs["window"] = [0, 0, 1, 2]; recs["seq_in_window"] = [1, 0, 0, 0]
    assert not ldist.records_in_replay_order(recs, None)


def test_shard_windows_partitions_everything():
    from lancet_amd import dist as ldist
    for world in (1, 2, 4, 8):
        seen = sorted(w for r in range(world) for w in ldist.shard_windows(1000, r, world, chunk=64))
        assert seen == list(range(1000))


def test_bench_spawns_its_ranks(monkeypatch):
    """`python bench.py --gpus N` without WORLD_SIZE must re-launch itself under torch.distributed.run with N ranks."""
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    seen = {}
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(os, "execvp", lambda prog, argv: seen.setdefault("argv", argv))
    bench.maybe_spawn(["--gpus", "4", "--steps", "2"], 4)
    a = seen["argv"]
    assert "torch.distributed.run" in a and "--nproc-per-node=4" in a and a[-4:] == ["--gpus", "4", "--steps", "2"]
    monkeypatch.setenv("WORLD_SIZE", "4")
    seen.clear()
    bench.maybe_spawn(["--gpus", "4"], 4)
    assert not seen


def test_strong_scaling_shards_reassemble_the_contig():
    """`bench.py --scaling strong`: one contig dealt out in chunks, every rank's interleaved runs of windows concatenated into ITS batch
    (workload.concat_batches); the ranks' records, mapped back through their window lists, are the one-process records."""
    import numpy as np
    from lancet_amd import abi, dist as ldist, workload
    from oracle import oracle
    full = workload.make_scan_batch(40, 20, 20, seed=3, read_len=100)
    p = abi.default_params()
    want, wst, _ = oracle.run(full, p)
    got, nwin = [], 0
    for rank in range(3):
        mine = np.array(ldist.shard_windows(40, rank, 3, chunk=8), dtype=np.int64)
        runs = np.split(mine, np.where(np.diff(mine) != 1)[0] + 1)
        batch = workload.concat_batches([workload.sub_batch(full, int(r[0]), int(r[-1]) + 1) for r in runs])
        assert batch.n_windows == len(mine)
        nwin += batch.n_windows
        v, st, _ = oracle.run(batch, p)
        assert [s["final_k"] for s in st] == [wst[int(w)]["final_k"] for w in mine]
        for r in v:
            r["window"] = int(mine[r["window"]])
            got.append(r)
    assert nwin == 40
    assert sorted(got, key=lambda r: (r["window"], r["seq"])) == want
