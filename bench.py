#!/usr/bin/env python3
"""bench.py -- assembled windows/sec of the per-window micro-assembly hot path on MI355X.

Workload (BASELINE.json configs[1] proxy): synthetic "chr22 scan", 600 bp windows with
stride 100, paired 2x150 bp reads at 30x tumor / 30x normal, 0.5 % substitution errors, planted germline +
somatic variants, self-tuning k = 11..101, every window assembled (== --active-region-off).
A step = one pass of the hot path (lancet_engine_run) over one batch of windows already resident in HBM.
N GPUs: each rank assembles its own batch (weak scaling) and the variant records are gathered to rank 0
over RCCL inside the timed step.

Prints ONE JSON line on rank 0."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--windows", type=int, default=32768, help="windows per GPU per step")
    ap.add_argument("--cov", type=float, default=30.0, help="coverage per sample")
    ap.add_argument("--cpu-sample", type=int, default=1536, help="windows timed on the CPU oracle (0 = skip)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from lancet_amd import abi, engine, workload
    from lancet_amd import dist as ldist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    batch = workload.make_scan_batch(args.windows, args.cov, args.cov, seed=22 + 1000 * rank)
    params = abi.default_params()
    eng = engine.Engine(params, device=local_rank)
    t_up = time.perf_counter()
    eng.upload(batch)                      # host -> HBM + trim/pack: outside the timed region
    upload_ms = 1000.0 * (time.perf_counter() - t_up)
    n_slots, slot_bytes = eng.geometry()

    def step():
        eng.run()
        if world > 1:
            vp, n, blob, _ = eng.raw_results()
            ldist.gather_bytes(ldist.pack_records(vp, n, blob), device)

    for _ in range(args.warmup):
        step()
    kernel_ms = []
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        kernel_ms.append(eng.timing_ms()[1])
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    variants, stats = eng.results()
    n_kmers = int(sum(s["n_kmers"] for s in stats))
    n_bad = sum(1 for s in stats if s["status"] < 0)
    alg_bytes = workload.algorithmic_bytes(batch, stats, len(variants))
    if world > 1:
        t = torch.tensor([n_kmers, n_bad], dtype=torch.int64, device=device)
        dist.all_reduce(t)
        n_kmers_all, n_bad_all = int(t[0].item()), int(t[1].item())
    else:
        n_kmers_all, n_bad_all = n_kmers, n_bad

    if rank == 0:
        ms_kernel = float(np.mean(kernel_ms))
        achieved = alg_bytes / (ms_kernel * 1e-3) / 1e9
        out = {
            "metric": "assembled windows/sec (whole node), 600bp windows, self-tuning k, synthetic T/N",
            "value": round(world * args.windows * args.steps / dt, 2),
            "unit": "windows/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1000.0 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "mkmers_per_s": round(n_kmers_all * args.steps / dt / 1e6, 2),
            "overflowed_windows": n_bad_all,
            "config": {"workload": f"chr22-scan proxy: {args.windows} windows/GPU x 600 bp, stride 100, "
                                   f"{args.cov:g}x tumor / {args.cov:g}x normal, 2x150 bp, k=11..101, active-region-off",
                       "windows_per_gpu": args.windows, "coverage": [args.cov, args.cov], "reads_per_gpu": int(batch.n_reads),
                       "variants_rank0": len(variants), "slots_in_flight": n_slots, "upload_ms": round(upload_ms, 1), "windows_rerun_tier2": eng.rerun_count(), "workspace_MB_per_slot": round(slot_bytes / 2 ** 20, 1)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(achieved / 8000.0, 6), "traffic": None,
                         "kernel": "window_kernel", "kernel_ms": round(ms_kernel, 3), "algorithmic_bytes_per_launch": alg_bytes},
        }
        # HBM-side bytes per launch from the committed PMC passes (tools/pmc_total.sh -> profiles/r1_traffic.json):
        # rocprofv3 cannot run inside this process, so the figure is looked up for the exact workload it was taken on
        try:
            with open(os.path.join(ROOT, "profiles", "r1_traffic.json")) as fh:
                for rec in json.load(fh)["measurements"]:
                    if rec["windows"] == args.windows and rec["coverage"] == args.cov and world == 1:
                        out["roofline"]["traffic"] = int((rec["FETCH_SIZE_KB"] + rec["WRITE_SIZE_KB"]) * 1024)
                        out["roofline"]["traffic_note"] = rec["note"]
                        break
        except (OSError, KeyError, ValueError):
            pass
        if args.cpu_sample and world == 1:
            from oracle import oracle
            ns = min(args.cpu_sample, batch.n_windows)
            sample = workload.sub_batch(batch, 0, ns)
            t1 = time.perf_counter()
            ov, ostats, _ = oracle.run(sample, params)
            cdt = time.perf_counter() - t1
            same = ov == [v for v in variants if v["window"] < ns]
            out["cpu_baseline"] = {"value": round(ns / cdt, 2), "unit": "windows/s", "cores": 1, "kind": "port",
                                   "sample": f"first {ns} windows of the same batch, oracle/liblancet_oracle.so, 1 thread, {cdt:.1f} s",
                                   "mkmers_per_s": round(sum(s["n_kmers"] for s in ostats) / cdt / 1e6, 3),
                                   "gpu_results_identical_on_sample": bool(same)}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
