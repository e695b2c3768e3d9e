"""Random parameter / workload sweep on the device: the HIP engine against the oracle on synthetic scan batches drawn
the way tools/fuzz_reference.py draws them for the emulated kernels (which cannot show a device-only race).  Options
are the reference's (src/Lancet.cc:653-760); the records, the per-window status / final k / number of builds / k-mer
trip count and the node high-water mark must all be equal."""
from concurrent.futures import ThreadPoolExecutor
import os

import numpy as np
import pytest

from lancet_amd import abi, engine, workload
from oracle import oracle

pytestmark = pytest.mark.gpu


# LANCET_SWEEP_OFFSET=<n>: other draws than the committed 48 (n is added to every seed); LANCET_SWEEP_LINKED=1: every draw with BX / HP tags
# and --linked-reads.  For a longer run on the GPU box beside the suite (tools/fat_check.sh style), not set by the tests themselves.
OFFSET = int(os.environ.get("LANCET_SWEEP_OFFSET", "0"))
LINKED = bool(os.environ.get("LANCET_SWEEP_LINKED"))


def draw(seed):
    seed += OFFSET
    rng = np.random.default_rng(90000 + seed)
    pick = lambda xs: xs[int(rng.integers(0, len(xs)))]
    over = {}
    if rng.random() < 0.7:
        over["min_k"] = int(pick([10, 11, 12, 13, 14, 15, 16, 20]))
    if rng.random() < 0.4:
        over["max_k"] = int(pick([31, 35, 61, 85, 101]))
    if rng.random() < 0.4:
        over["max_tip_len"] = int(rng.integers(4, 20))
    if rng.random() < 0.4:
        over["cov_threshold"] = int(rng.integers(3, 11))
    if rng.random() < 0.3:
        over["low_cov_threshold"] = int(pick([1, 2]))
    if rng.random() < 0.4:
        over["max_mismatch"] = int(rng.integers(0, 4))
    if rng.random() < 0.3:
        over["min_qual_trim"] = 33 + int(rng.integers(5, 21))
    if rng.random() < 0.3:
        over["min_qual_call"] = 33 + int(rng.integers(5, 38))
    if rng.random() < 0.3:
        over["max_unit_len"] = int(rng.integers(1, 7))
    if rng.random() < 0.3:
        over["min_report_units"] = int(rng.integers(1, 5))
    if rng.random() < 0.3:
        over["min_report_len"] = int(rng.integers(2, 14))
    if rng.random() < 0.3:
        over["dist_from_str"] = int(rng.integers(0, 3))
    if rng.random() < 0.3:
        over["max_indel_len"] = int(rng.integers(20, 200))
    if rng.random() < 0.2:
        over["dfs_limit"] = int(pick([500, 5000]))
    if rng.random() < 0.3:
        over["min_cov_ratio"] = 0.02
    wl = dict(cov_t=float(pick([6, 12, 18, 30, 45, 70])), cov_n=float(pick([6, 12, 15, 30, 40])),
              read_len=int(pick([76, 100, 125, 150, 250])), error_rate=float(pick([0.0, 0.003, 0.005, 0.008, 0.015])),
              str_fraction=float(pick([0.0, 0.0, 0.1, 0.3])), lowcomplex_fraction=float(pick([0.0, 0.0, 0.05])),
              somatic_every=int(pick([400, 1000, 2000])), germline_every=int(pick([300, 1000])))
    if LINKED:
        over["lr_mode"] = 1; wl["linked"] = True
    return over, wl


def oracle_parallel(batch, p, per=16):
    chunks = [(a, min(batch.n_windows, a + per)) for a in range(0, batch.n_windows, per)]
    def one(ab):
        v, st, _ = oracle.run(workload.sub_batch(batch, ab[0], ab[1]), p)
        for r in v:
            r["window"] += ab[0]
        return v, st
    oracle.lib()
    with ThreadPoolExecutor(max_workers=min(64, os.cpu_count() or 1)) as ex:
        res = list(ex.map(one, chunks))
    return [r for v, _ in res for r in v], [s for _, st in res for s in st]


@pytest.mark.parametrize("seed", range(48))
def test_random_options_and_workloads_match_the_oracle(seed):
    over, wl = draw(seed)
    p = abi.default_params(**over)
    batch = workload.make_scan_batch(384, seed=700 + seed + OFFSET, **wl)
    eng = engine.Engine(p, device=0)
    variants, stats = eng.process(batch)
    ov, ostats = oracle_parallel(batch, p)
    assert all(s["status"] >= 0 for s in stats), (over, wl, [s for s in stats if s["status"] < 0][:3])
    key = lambda s: (s["status"], s["final_k"], s["n_builds"], s["n_variants"], s["n_kmers"], s["max_nodes"])
    bad = [w for w in range(batch.n_windows) if key(stats[w]) != key(ostats[w])]
    assert not bad, (over, wl, bad[:5], [(key(stats[w]), key(ostats[w])) for w in bad[:3]])
    assert variants == ov, (over, wl)
    variants2, _ = eng.process(batch)                       # and again on the same engine: same records
    assert variants2 == variants
    eng.close()


@pytest.mark.parametrize("seed", range(12))
def test_random_draws_through_the_rerun_tier(seed, monkeypatch):
    """The same draws with a tier-1 node table of 64 entries: practically every window overflows tier 1 and is assembled by the
    several-wave kernel of the re-run tier (window_fat.hip) -- its helper waves, split quality counts and grouped mate prefilter
    against the oracle, on ordinary windows rather than the rare pile-up."""
    monkeypatch.setenv("LANCET_NODE_CAP1", "64")
    over, wl = draw(seed)
    p = abi.default_params(**over)
    batch = workload.make_scan_batch(256, seed=700 + seed + OFFSET, **wl)
    eng = engine.Engine(p, device=0)
    variants, stats = eng.process(batch)
    if eng.rerun_count() == 0 and all(s["n_builds"] == 0 for s in stats):
        eng.close()
        pytest.skip("this draw builds no graph (every k of every window is turned down by the reference repeat test): nothing can overflow tier 1")
    assert eng.rerun_count() > 0
    ov, ostats = oracle_parallel(batch, p)
    key = lambda s: (s["status"], s["final_k"], s["n_builds"], s["n_variants"], s["n_kmers"], s["max_nodes"])
    assert all(s["status"] >= 0 for s in stats), (over, wl)
    assert [key(s) for s in stats] == [key(s) for s in ostats], (over, wl)
    assert variants == ov, (over, wl)
    eng.close()
