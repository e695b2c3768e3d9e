"""N>1 path on CPU: two gloo ranks shard windows, pack their variant records and gather them to rank 0, where the
VariantDB/VCF of the union must equal the single-process result."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import golden_util as gu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, case, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ctypes as C
    from lancet_amd import abi, dist as ldist, engine, workload
    from oracle import oracle
    meta, batch, kept, (min_k, max_k) = gu.case_batch(case)
    mine = ldist.shard_windows(batch.n_windows, rank, world, chunk=7)
    # contiguous runs of windows -> sub batches; records produced here by the oracle (no GPU in this test)
    recs = []
    for w in mine:
        sub = workload.sub_batch(batch, w, w + 1)
        v, _, _ = oracle.run(sub, abi.default_params(min_k=min_k, max_k=max_k))
        for r in v:
            r["window"] = w
            recs.append(r)
    arr = (abi.LancetVariant * len(recs))()
    blob = bytearray()
    for i, r in enumerate(recs):
        x = arr[i]
        x.window, x.seq_in_window, x.chr_id, x.pos = r["window"], r["seq"], r["chr_id"], r["pos"]
        x.code, x.prev_bp_ref, x.prev_bp_alt, x.kmer = ord(r["code"]), ord(r["prev_bp_ref"]), ord(r["prev_bp_alt"]), r["kmer"]
        for k in range(8):
            x.cov[k] = r["cov"][k]
        x.ref_off, x.ref_len = len(blob), len(r["ref"]); blob += r["ref"].encode()
        x.alt_off, x.alt_len = len(blob), len(r["alt"]); blob += r["alt"].encode()
        x.str_off, x.str_len = len(blob), len(r["str"]); blob += r["str"].encode()
    payload = ldist.pack_records(arr, len(recs), bytes(blob))
    parts = ldist.gather_bytes(payload, torch.device("cpu"))
    if rank == 0:
        allrecs = []
        for buf in parts:
            a, n, b = ldist.unpack_records(buf)
            allrecs += abi.variants_to_py(a, n, b)
        allrecs.sort(key=lambda r: (r["window"], r["seq"]))      # replay in window order, whatever the rank count
        db = engine.VariantDB()
        db.add_records(allrecs, ["chr22"])
        q.put(db.vcf())
    dist.destroy_process_group()


@pytest.mark.parametrize("case", ["tile30"])
def test_two_rank_gather_reproduces_single_process_vcf(case):
    from lancet_amd import build
    build.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, case, q)) for r in range(2)]
    for p in procs:
        p.start()
    vcf = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert vcf == gu.golden_vcf(case)


def test_shard_windows_partitions_everything():
    from lancet_amd import dist as ldist
    for world in (1, 2, 4, 8):
        seen = sorted(w for r in range(world) for w in ldist.shard_windows(1000, r, world, chunk=64))
        assert seen == list(range(1000))
