#!/usr/bin/env python3
"""bench.py -- assembled windows/sec of the per-window micro-assembly hot path on MI355X.

Workload (BASELINE.json configs[1] proxy): synthetic "chr22 scan", 600 bp windows with
stride 100, paired 2x150 bp reads at 30x tumor / 30x normal, 0.5 % substitution errors, planted germline +
somatic variants, self-tuning k = 11..101, every window assembled (== --active-region-off).
A step = one pass of the hot path (lancet_engine_run) over one batch of windows already resident in HBM.

N GPUs (`--gpus N`; without WORLD_SIZE in the environment the script re-launches itself under torch.distributed.run with
N ranks): rank r assembles its own synthetic contig (weak scaling); inside the timed step the variant records of every
rank are sent to rank 0 over RCCL (sizes by all_gather, payloads point to point) and rank 0 replays them in window order
into a VariantDB.

Order of a run: upload (outside the timed region), `--settle` untimed steps (default 300: a device fresh from boot was seen to run its
first seconds of steps 7 % slower than ever after, with the kernels' own durations unchanged), `--warmup` untimed steps, barrier +
synchronize, EXACTLY `--steps` timed steps, barrier + synchronize; then, untimed, the kernel durations on launches that have the GPU to
themselves (roofline), the PCIe-inclusive loop, the CPU baseline, the native BAM -> VCF run and the side configurations.

Prints ONE JSON line on rank 0."""
import argparse
import hashlib
import json
import os
import queue
import threading
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def maybe_spawn(argv, gpus: int) -> None:
    """`python bench.py --gpus N` alone: become N ranks (one process per GPU) under torch.distributed.run."""
    if gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    port = 29400 + (os.getpid() % 500)
    os.execvp(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
                               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv))


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_limits() -> dict:
    """What this process may actually use of the host: affinity mask, cgroup CPU quota (a container with a quota of N CPUs scales
    to N whatever os.cpu_count() says)."""
    out = {}
    try:
        out["affinity_cpus"] = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                out["cgroup_cpu_max"] = " ".join(txt)
                if txt and txt[0] != "max" and len(txt) > 1:
                    out["cgroup_cpus"] = round(int(txt[0]) / int(txt[1]), 2)
            else:
                q = int(txt[0]); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                out["cgroup_cpus"] = round(q / per, 2) if q > 0 else None
            break
        except (OSError, ValueError):
            continue
    try:
        out["loadavg"] = [float(x) for x in open("/proc/loadavg").read().split()[:3]]
    except OSError:
        pass
    return out


def cpu_baseline(batch, params, variants, n1: int, nall: int):
    """The oracle (CPU restatement of the reference path, oracle/liblancet_oracle.so) on a bounded sample of the same
    windows: one thread, then one worker PROCESS per hardware thread over chunks of windows (windows are independent, as
    the reference's own --num-threads fan-out, src/Lancet.cc:910-928; processes, not threads of one process: oracle/cpu_fanout.py)."""
    from lancet_amd import workload
    from oracle import cpu_fanout, oracle
    oracle.lib()
    n1 = min(n1, batch.n_windows)
    sample = workload.sub_batch(batch, 0, n1)
    t = time.perf_counter()
    ov, ostats, _ = oracle.run(sample, params)
    dt1 = time.perf_counter() - t
    same = ov == [v for v in variants if v["window"] < n1]
    threads = os.cpu_count() or 1
    try:
        threads = min(threads, len(os.sched_getaffinity(0)))      # (what this process may run on)
    except (AttributeError, OSError):
        pass
    lim = cpu_limits()
    # a container with a CPU quota (cgroup cpu.max) gets that many CPUs' worth of time however many hardware threads it sees:
    # the quota is what "cores" means here, and more runnable processes than about twice the quota only add throttling
    quota = lim.get("cgroup_cpus")
    cores = threads if not quota else max(1, min(threads, int(round(quota))))
    procs = threads if not quota else max(1, min(threads, 2 * cores))
    phys = cpu_fanout.physical_cores()
    nall = min(nall, batch.n_windows)
    per = max(4, nall // (procs * 6))
    dta, kma, done, orecs = cpu_fanout.run(batch, {}, nall, procs, per, want_records=True)
    # every record the fan-out produced against the engine's, field by field, in (window, emission) order: the full sample, not its head
    erecs = [v for v in variants if v["window"] < nall]
    same_all = orecs == erecs
    km1 = sum(s["n_kmers"] for s in ostats)
    return {"value": round(done / dta, 2), "unit": "windows/s", "cores": cores, "processes": procs, "physical_cores": phys, "hardware_threads": threads,
            "kind": "port", "cpu_model": cpu_model(),
            "sample": f"oracle/liblancet_oracle.so on the first {nall} windows of the same batch, {procs} worker processes over chunks of {per} windows, {dta:.1f} s; "
                      f"the host has {threads} hardware threads ({phys} physical cores)" + (f", this container a CPU quota of {quota:g} (cgroup cpu.max): `cores` is the quota" if quota else ""),
            "mkmers_per_s": round(kma / dta / 1e6, 3),
            "one_thread": {"value": round(n1 / dt1, 2), "mkmers_per_s": round(km1 / dt1 / 1e6, 3),
                           "sample": f"first {n1} windows, 1 thread, {dt1:.1f} s"},
            "scaling_over_one_thread": round((done / dta) / (n1 / dt1), 1), "host_limits": lim,
            "gpu_results_identical_on_sample": bool(same and same_all),
            "windows_compared": int(nall), "records_compared": len(erecs), "one_thread_windows_compared": int(n1),
            "note": "the reference binary cannot travel to this box; in the authoring container it runs the golden cases at 13-23 windows/s/thread, this port at ~45 (DESIGN.md §7)"}


def kernel_rooflines(alg_build: int, alg_window: int, per_kernel: dict) -> dict:
    """Roofline of each kernel on its own share of the algorithmic bytes (workload.algorithmic_bytes_split)."""
    out = {}
    for nm, alg in (("build_kernel", alg_build), ("window_kernel", alg_window)):
        ms = float(per_kernel.get(nm, 0.0))
        if ms > 0:
            out[nm] = {"algorithmic_bytes_per_launch": int(alg), "ms": round(ms, 3), "achieved": round(alg / (ms * 1e-3) / 1e9, 3),
                       "frac": round(alg / (ms * 1e-3) / 1e9 / 8000.0, 6)}
    return out


def side_config(eng_cls, params, name, windows, cov_t, cov_n, steps, streaming=False, **kw):
    """One more BASELINE.md configuration on this GPU (smaller batch, reported beside the headline), with its own roofline:
    algorithmic bytes of the batch / HIP-event durations of its kernels.  `windows_per_s` is one batch at a time on the GPU (submit, wait,
    submit ...).  streaming=True adds `windows_per_s_streaming`: two engines with the batch resident, submitted in turn and NOT chained -- the
    next batch's window kernel fills the slots the last multi-build windows of this one leave idle, as in a scan that streams batch after
    batch (for a configuration that runs the window kernel alone: next to a build kernel the overlap costs more than it hides, bench --chain)."""
    import numpy as np
    from lancet_amd import workload
    b = workload.make_scan_batch(windows, cov_t, cov_n, seed=22, **kw)
    eng = eng_cls(params, device=0)
    eng.upload(b)
    eng.run()
    kms = []
    t = time.perf_counter()
    for _ in range(steps):
        eng.run()
        kms.append(eng.kernel_times())
    dt = (time.perf_counter() - t) / steps
    variants, stats = eng.results()
    ks, cnt = np.unique([s["final_k"] for s in stats if s["status"] == 0], return_counts=True)
    names = eng.kernel_names()
    per_kernel = {nm: round(float(np.mean([k[i] for k in kms])), 3) for i, nm in enumerate(names)}
    ms_all = float(sum(per_kernel.values()))
    alg = workload.algorithmic_bytes(b, stats, len(variants))
    alg_b, alg_w = workload.algorithmic_bytes_split(b, stats, len(variants))
    built = sum(1 for s in stats if s["n_builds"] > 0)
    out = {"name": name, "windows": windows, "steps": steps, "coverage": [cov_t, cov_n], "windows_per_s": round(windows / dt, 1),
           "mkmers_per_s": round(sum(s["n_kmers"] for s in stats) / dt / 1e6, 1), "reads_per_window": round(b.n_reads / windows, 1),
           "builds_per_window": round(sum(s["n_builds"] for s in stats) / windows, 3),
           "final_k_histogram": {int(k): int(c) for k, c in zip(ks, cnt)},
           "k_exhausted": sum(1 for s in stats if s["status"] == 2), "overflowed": sum(1 for s in stats if s["status"] < 0),
           "windows_first_graph_in_lds": eng.prebuilt_count(), "windows_that_build": built, "building_windows_per_s": round(built / dt, 1),
           "windows_rerun_worst_case_tier": eng.rerun_count(),
           "build_service": dict(zip(("posted", "served", "not_buildable", "taken_back"), eng.svc_counts())),
           "roofline": {"bound": "hbm", "achieved": round(alg / (ms_all * 1e-3) / 1e9, 3), "peak": 8000.0, "unit": "GB/s",
                        "frac": round(alg / (ms_all * 1e-3) / 1e9 / 8000.0, 6), "algorithmic_bytes_per_launch": int(alg),
                        "kernel_ms": round(ms_all, 3), "per_kernel_ms": per_kernel, "per_kernel": kernel_rooflines(alg_b, alg_w, per_kernel)}}
    if streaming:
        e2 = eng_cls(params, device=0)
        e2.upload(b)
        pair = [eng, e2]
        pend = []
        for i in range(2):                      # (warm-up: both engines once)
            pair[i].submit(); pend.append(pair[i])
        while pend:
            pend.pop(0).wait()
        t = time.perf_counter()
        for i in range(2 * steps):
            e = pair[i % 2]
            e.submit(); pend.append(e)
            if len(pend) >= 2:
                pend.pop(0).wait()
        while pend:
            pend.pop(0).wait()
        out["windows_per_s_streaming"] = round(windows * 2 * steps / (time.perf_counter() - t), 1)
        out["streaming_note"] = "two engines with the batch resident, submitted in turn, not chained; results read back every step"
        v2, _ = e2.results()
        out["streaming_results_identical"] = bool(v2 == variants)
        e2.close()
    eng.close()
    return out


def bam_e2e(d: str, region_args, gen_args, what: str, extra=("--devices", "0,0", "--batch-windows", "8192")):
    """BAM -> VCF with the native command-line program (lancet_amd/bin/lancet_gpu: BGZF inflate, alignment decode, tiling, read selection,
    trim + pack, upload, kernels, VariantDB, VCF) on a synthetic tumor / normal pair made by tools/make_scan_bams.py -- made here when it
    is not there yet (build/ is not part of the repository's history; always 14 worker processes: the stretches, and so the reads, depend on
    that number).  A whole process per run -- start-up, device initialisation and the work-space allocation included -- three times, the
    fastest reported; windows / wall seconds."""
    import hashlib
    import re
    import subprocess
    exe = os.path.join(ROOT, "lancet_amd", "bin", "lancet_gpu")
    need = [os.path.join(d, f) for f in ("tumor.bam", "normal.bam", "ref.fa")]
    made_s = None
    if os.path.exists(exe) and not all(os.path.exists(f) for f in need):
        t = time.perf_counter()
        try:
            subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_scan_bams.py"), d] + [str(x) for x in gen_args], capture_output=True, text=True, timeout=600)
        except (OSError, subprocess.TimeoutExpired):
            return None
        made_s = round(time.perf_counter() - t, 1)
    if not os.path.exists(exe) or not all(os.path.exists(f) for f in need):
        return None
    res = None; walls = []; md5s = set()
    for _ in range(3):
        time.sleep(5)          # (a process that allocates tens of GB right after another one released as much waits seconds in hipMalloc for the driver's wipe)
        try:
            r = subprocess.run([exe, "--tumor", need[0], "--normal", need[1], "--ref", need[2]] + [a.replace("{d}", d) for a in region_args] + ["--active-region-off"] + list(extra),
                               capture_output=True, text=True, timeout=300, env=dict(os.environ, LANCET_HOST_TIMING="1"))
        except (OSError, subprocess.TimeoutExpired):
            return None
        m = re.search(r"\[lancet_gpu\] (\d+) windows tiled, (\d+) assembled.*?(\d+) variants", r.stderr)
        w = re.search(r"\[lancet_gpu\] wall ([0-9.]+) s: input decode \+ tiling ([0-9.]+), window filters \+ batches ([0-9.]+), engine \(upload \+ kernels \+ results\) ([0-9.]+) \(kernels ([0-9.]+)\), VariantDB ([0-9.]+)", r.stderr)
        if r.returncode not in (0, 3) or not m or not w:
            return None
        md5s.add(hashlib.md5("".join(l + "\n" for l in r.stdout.splitlines() if not l.startswith(("##fileDate", "##cmdline", "##reference"))).encode()).hexdigest())
        wall = float(w.group(1))
        walls.append(wall)
        if res is not None and wall >= res["wall_s"]:
            continue
        res = {"value": round(int(m.group(2)) / wall, 1), "unit": "windows/s", "windows": int(m.group(2)), "variants": int(m.group(3)), "wall_s": wall,
               "decode_tiling_s": float(w.group(2)), "filters_batches_s": float(w.group(3)), "engine_s": float(w.group(4)), "kernels_s": float(w.group(5)),
               "vcf_records": sum(1 for l in r.stdout.splitlines() if l and not l.startswith("#")),
               "what": f"lancet_gpu --tumor/--normal/--ref {' '.join(region_args)} --active-region-off {' '.join(extra)} on {os.path.relpath(d, ROOT)} ({what}), the fastest of three runs, "
                       "a whole process: start-up, device initialisation and allocation included (a run that follows a process which just released tens of GB "
                       "of device memory may wait in hipMalloc for the driver to wipe it: DESIGN_HISTORY.md 7a)"}
    if res:
        res["wall_s_all_runs"] = walls
        res["vcf_md5"] = sorted(md5s)[0] if len(md5s) == 1 else sorted(md5s)          # (one value: the three runs wrote the same VCF)
        if made_s is not None:
            res["inputs_made_s"] = made_s
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100, help="timed steps (default: ~5 s of timed region at ~50 ms per step)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--windows", type=int, default=32768, help="windows per GPU per step")
    ap.add_argument("--cov", type=float, default=30.0, help="coverage per sample")
    ap.add_argument("--cov-normal", type=float, default=None, help="coverage of the normal sample when it differs from --cov (config 4: --cov 100 --cov-normal 40)")
    ap.add_argument("--str-fraction", type=float, default=0.0, help="share of the synthetic reference that is short tandem repeats (config 4: 0.30)")
    ap.add_argument("--lowcomplex-fraction", type=float, default=0.0, help="share that is low-complexity sequence (config 4: 0.05)")
    ap.add_argument("--linked", action="store_true", help="BX / HP tags on every pair and --linked-reads in the engine (config 5)")
    ap.add_argument("--cpu-sample", type=int, default=1024, help="windows timed on one thread of the CPU oracle (0 = skip the CPU legs)")
    ap.add_argument("--cpu-sample-all", type=int, default=16384, help="windows timed on all host cores")
    ap.add_argument("--no-bam", action="store_true", help="skip the BAM -> VCF run of the native program (value_bam_e2e)")
    ap.add_argument("--no-configs", action="store_true", help="skip the side configurations (60x/60x, 100x/40x STR)")
    ap.add_argument("--settle", type=int, default=int(os.environ.get("LANCET_BENCH_SETTLE_STEPS", "300")),
                    help="untimed steps before the warmup (a device fresh from boot runs the first seconds 7 %% slower; 0 = none)")
    ap.add_argument("--in-flight", type=int, default=2, help="batches in flight per GPU (engines submitted in turn); 1 = every step alone on the GPU")
    ap.add_argument("--chain", type=int, default=1, help="1: a batch's kernels start when those of the batch before it (other engine) are through -- back to back, no host gap; 0: as soon as submitted")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: every rank assembles its own contig of --windows windows; strong: ONE contig of --windows windows dealt out over the ranks in chunks (dist.shard_windows)")
    args = ap.parse_args()
    maybe_spawn(sys.argv[1:], args.gpus)

    import numpy as np
    import torch
    import torch.distributed as dist
    from lancet_amd import abi, engine, workload
    from lancet_amd import dist as ldist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # LANCET_BENCH_ONE_GPU=1 (a check of the N-rank path on a box with one GPU, never a measurement): every rank on device 0,
    # the record gather over gloo through host memory instead of RCCL.  The JSON line says so ("comm").
    one_gpu = world > 1 and os.environ.get("LANCET_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
        os.environ.setdefault("LANCET_MEM_GB", str(max(8, 160 // (world * max(1, args.in_flight)))))   # the ranks share one device's HBM
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if one_gpu:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    comm_device = torch.device("cpu") if one_gpu else device

    cov_n = args.cov if args.cov_normal is None else args.cov_normal
    wl_kw = dict(str_fraction=args.str_fraction, lowcomplex_fraction=args.lowcomplex_fraction, linked=args.linked)
    rccl_ranks = dist.get_world_size() if world > 1 else 1     # what the process group reports, not what was asked for
    strong = args.scaling == "strong" and world > 1
    if strong:
        # one contig, its windows dealt out in chunks of 1024 (dist.shard_windows): every rank holds interleaved runs of windows, so the
        # records reach rank 0 out of window order and the replay has to restore it
        chrom = "chr22"
        full = workload.make_scan_batch(args.windows, args.cov, cov_n, seed=22, chrom=chrom, **wl_kw)
        mine = np.array(ldist.shard_windows(args.windows, rank, world, chunk=1024), dtype=np.int64)
        runs = np.split(mine, np.where(np.diff(mine) != 1)[0] + 1) if len(mine) else []
        batch = workload.concat_batches([workload.sub_batch(full, int(r[0]), int(r[-1]) + 1) for r in runs])
        windex = mine
    else:
        chrom = f"chr{22 + rank}" if rank else "chr22"           # one synthetic contig per rank
        batch = workload.make_scan_batch(args.windows, args.cov, cov_n, seed=22 + 1000 * rank, chrom=chrom, **wl_kw)
        windex = rank * args.windows + np.arange(args.windows, dtype=np.int64)
    n_local = batch.n_windows
    params = abi.default_params(lr_mode=1) if args.linked else abi.default_params()
    # `--in-flight 2` (default): two engines on the GPU, each with the batch resident, submitted in turn -- the kernels of step
    # i+1 are queued while step i drains, so the tail of a batch (a few windows that need several k attempts) overlaps the bulk of
    # the next one, as in a scan that streams batch after batch.  Every step is completed (results on the host) inside the timed region.
    nfl = max(1, args.in_flight)
    engs = [engine.Engine(params, device=local_rank) for _ in range(nfl)]
    eng = engs[0]
    t_up = time.perf_counter()
    eng.upload(batch)                      # host -> HBM + trim/pack: outside the timed region
    upload_first_ms = 1000.0 * (time.perf_counter() - t_up)       # includes the one-off work-space allocation
    for e2 in engs[1:]:
        e2.upload(batch)
    n_slots, slot_bytes = eng.geometry()
    last = {}
    tm = {"gather": 0.0, "replay": 0.0, "pack": 0.0, "host": 0.0, "steps": 0}

    # N ranks: what happens to a finished step's records -- keys + per-key reduction (pack_records), then the sizes all_gather and the payload
    # send / recv (gather_bytes), on rank 0 then the replay -- runs on threads of its own, a stage each, so that the thread that submits the
    # kernels only waits for them (step i's gather overlaps step i + 1's kernels; the C side and the collectives release the GIL) and a
    # slow gather (rank 0 waits for the slowest rank) does not hold up the packing of the next step's records.  An engine's result
    # buffers are valid until its next submit: `busy[engine]` is set when its records have been packed.
    comm_q = queue.Queue()
    gather_q = queue.Queue()
    busy = {}

    def comm_worker():                           # stage 1: the finished engine's records -> one payload (keys, per-key reduction)
        while True:
            item = comm_q.get()
            try:
                if item is None:
                    return
                e, packed = item
                try:
                    vp, n, blob, _ = e.raw_results()
                    tg = time.perf_counter()
                    payload = ldist.pack_records(vp, n, blob, chr_names=[chrom], window_index=windex)
                    tm["pack"] += time.perf_counter() - tg
                finally:
                    packed.set()                 # (the engine may be submitted again)
                gather_q.put(payload)
            except BaseException as ex:          # surfaced by run_steps
                last["error"] = ex
                gather_q.put(b"")                # (the other ranks wait in the collective: take part with nothing)
            finally:
                comm_q.task_done()

    def gather_worker():                         # stage 2: the only thread that issues collectives while steps run, in step order on every rank
        torch.cuda.set_device(device)            # (the current device is per thread)
        while True:
            payload = gather_q.get()
            try:
                if payload is None:
                    return
                tg = time.perf_counter()
                parts = ldist.gather_bytes(payload, comm_device)
                tm["gather"] += time.perf_counter() - tg; tm["steps"] += 1
                if rank == 0:
                    merge_q.put(parts)
            except BaseException as ex:
                last["error"] = ex
            finally:
                gather_q.task_done()

    done_at = []                                 # host clock at every completion of the timed steps (config.step_ms: where a slow run lost its time)

    kern_in_loop = []                            # the kernels' HIP-event durations of every timed step, as they ran in the pipeline (config.step_ms)

    def complete(e):
        e.wait()
        done_at.append(time.perf_counter())
        kern_in_loop.append(e.kernel_times())
        if world > 1:
            th = time.perf_counter()
            busy[id(e)] = threading.Event()
            comm_q.put((e, busy[id(e)]))
            tm["host"] += time.perf_counter() - th

    # rank 0 replays the gathered records on a further host thread, so that the replay of step i overlaps the gather of step i+1;
    # run_steps() returns only when every step it submitted has been gathered and replayed.
    merge_q = queue.Queue()

    def merger():
        while True:
            parts = merge_q.get()
            try:
                if parts is None:
                    return
                db = engine.VariantDB()
                tr = time.perf_counter()
                last["n"] = ldist.merge_into_vdb(parts, db)
                tm["replay"] += time.perf_counter() - tr
                last["db"] = db
            except BaseException as ex:          # surfaced by run_steps
                last["error"] = ex
            finally:
                merge_q.task_done()

    if world > 1:
        threading.Thread(target=comm_worker, daemon=True).start()
        threading.Thread(target=gather_worker, daemon=True).start()
        if rank == 0:
            threading.Thread(target=merger, daemon=True).start()

    def run_steps(k):
        _run_steps(k)
        comm_q.join()
        gather_q.join()
        merge_q.join()
        if "error" in last:
            raise last["error"]

    def _run_steps(k):
        pend = []
        for i in range(k):
            e = engs[i % nfl]
            ev = busy.pop(id(e), None)
            if ev is not None:
                th = time.perf_counter()
                ev.wait()                        # (its previous step's records are still being packed)
                tm["host"] += time.perf_counter() - th
            e.submit(after=pend[-1] if pend and args.chain else None)
            pend.append(e)
            if len(pend) >= nfl:
                complete(pend.pop(0))
        while pend:
            complete(pend.pop(0))

    # Settling (untimed, before the warmup steps that were asked for): in three of eight runs that were the first GPU process on a fresh
    # box every step of the timed loop took 3.1 ms longer than in the next process on the same box, with the kernels' own durations
    # unchanged (profiles/README.md, round 5) -- a device that has not been under load yet.  A fixed number of steps (the same on every
    # rank: no collective decides it), reported in config.settle.
    settle = {"steps": 0}
    if args.settle > 0:
        chunk = max(1, min(25, args.settle))
        ms = []
        while settle["steps"] < args.settle:
            n = min(chunk, args.settle - settle["steps"])
            tc = time.perf_counter(); run_steps(n); ms.append(1e3 * (time.perf_counter() - tc) / n)
            settle["steps"] += n
        settle.update({"ms_per_step_first_chunk": round(ms[0], 2), "ms_per_step_last_chunk": round(ms[-1], 2), "ms_per_step_min_chunk": round(min(ms), 2)})
    run_steps(args.warmup)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    tm["gather"] = tm["replay"] = tm["pack"] = tm["host"] = 0.0; tm["steps"] = 0
    del done_at[:]
    del kern_in_loop[:]
    t0 = time.perf_counter()
    run_steps(args.steps)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=comm_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        g = torch.tensor([1000.0 * tm["gather"] / max(1, tm["steps"])], dtype=torch.float64, device=comm_device)
        gl = [torch.zeros(1, dtype=torch.float64, device=comm_device) for _ in range(world)]
        dist.all_gather(gl, g)
        gather_ms = [round(float(x.item()), 3) for x in gl]
    step_ms = [1e3 * (b - a) for a, b in zip([t0] + done_at[:-1], done_at)][:args.steps]
    q = max(1, len(step_ms) // 5)
    step_summary = {"median": round(float(np.median(step_ms)), 2), "max": round(max(step_ms), 2), "slowest_step": int(np.argmax(step_ms)),
                    "mean_first_fifth": round(float(np.mean(step_ms[:q])), 2), "mean_last_fifth": round(float(np.mean(step_ms[-q:])), 2),
                    "over_1.25x_median": int(sum(1 for x in step_ms if x > 1.25 * np.median(step_ms)))} if step_ms else None
    if step_summary is not None and kern_in_loop:
        kl = [k for k in kern_in_loop[:args.steps] if k]
        if kl:      # (with two batches in flight these include the time a kernel shares the device with the other batch's copies; the roofline uses launches that are alone)
            step_summary["kernel_ms_in_the_pipeline"] = [round(float(np.mean([k[i] for k in kl])), 3) for i in range(len(kl[0]))]
    if os.environ.get("LANCET_BENCH_STEPS"):
        print("[bench] step ms:", [round(x, 1) for x in step_ms], file=sys.stderr)
    # kernel durations for the roofline: launches that have the GPU to themselves (with two batches in flight the HIP events of a
    # kernel also cover the time it shares the device with the other batch's kernels)
    kernel_ms = []
    for _ in range(2):
        eng.run()
        kernel_ms.append(eng.kernel_times())

    variants, stats = eng.results()
    n_kmers = int(sum(s["n_kmers"] for s in stats))
    n_bad = sum(1 for s in stats if s["status"] < 0)
    alg_bytes = workload.algorithmic_bytes(batch, stats, len(variants))
    alg_build, alg_window = workload.algorithmic_bytes_split(batch, stats, len(variants))
    if world > 1:
        t = torch.tensor([n_kmers, n_bad], dtype=torch.int64, device=comm_device)
        dist.all_reduce(t)
        n_kmers_all, n_bad_all = int(t[0].item()), int(t[1].item())
    else:
        n_kmers_all, n_bad_all = n_kmers, n_bad

    # PCIe-inclusive rate (never `value`): the batch handed over as host buffers every step; upload + trim/pack of one batch overlap
    # the kernels of the previous one when two engines take turns
    # (steady state: the clock runs from the completion of the second step to that of the last -- the first upload has no kernels to hide
    #  behind and the last kernels no upload to hide)
    pend = []
    ne2e = 10
    done_at = []
    tl = []
    for i in range(ne2e):
        e = engs[i % nfl]
        ta = time.perf_counter()
        e.upload(batch)
        tb = time.perf_counter()
        e.submit(after=pend[-1] if pend and args.chain else None)
        tc = time.perf_counter()
        pend.append(e)
        if len(pend) >= nfl:
            pend.pop(0).wait()
            done_at.append(time.perf_counter())
        tl.append((ta, tb, tc, time.perf_counter()))
    while pend:
        pend.pop(0).wait()
        done_at.append(time.perf_counter())
    e2e_s = (done_at[-1] - done_at[1]) / (ne2e - 2)
    if os.environ.get("LANCET_BENCH_TIMELINE"):
        z = tl[0][0]
        for i, (ta, tb, tc, td) in enumerate(tl):
            print(f"[e2e] step {i}: upload {1e3 * (ta - z):7.1f} .. {1e3 * (tb - z):7.1f}  submit .. {1e3 * (tc - z):7.1f}  wait .. {1e3 * (td - z):7.1f}", file=sys.stderr)
        print("[e2e] completions", [round(1e3 * (x - z), 1) for x in done_at], file=sys.stderr)
    t1 = time.perf_counter()
    eng.upload(batch)
    up_ms = 1000.0 * (time.perf_counter() - t1)

    if rank == 0:
        names = eng.kernel_names()
        per_kernel = {nm: round(float(np.mean([k[i] for k in kernel_ms])), 3) for i, nm in enumerate(names)}
        ms_all = float(sum(per_kernel.values()))
        achieved = alg_bytes / (ms_all * 1e-3) / 1e9
        own = [v for v in variants]
        h = hashlib.sha256()
        for v in own:
            h.update(repr((v["window"], v["seq"], v["pos"], v["code"], v["ref"], v["alt"], v["cov"], v["kmer"], v["str"])).encode())
        out = {
            "metric": "assembled windows/sec (whole node), 600bp windows, self-tuning k, synthetic T/N",
            "value": round((args.windows if strong else world * args.windows) * args.steps / dt, 2),
            "unit": "windows/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1000.0 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "mkmers_per_s": round(n_kmers_all * args.steps / dt / 1e6, 2),
            "overflowed_windows": n_bad_all,
            "value_e2e": round(n_local / e2e_s, 2),
            "value_e2e_note": f"PCIe-inclusive, per GPU: host buffers -> upload + trim/pack ({up_ms:.0f} ms per batch) + kernels, {nfl} batch(es) in flight; never `value`",
            "config": {"workload": f"chr22-scan proxy: {args.windows} windows/GPU x 600 bp, stride 100, "
                                   f"{args.cov:g}x tumor / {cov_n:g}x normal, 2x150 bp, k=11..101, active-region-off"
                                   + (f", {100 * args.str_fraction:g} % STR / {100 * args.lowcomplex_fraction:g} % low complexity" if args.str_fraction or args.lowcomplex_fraction else "")
                                   + (", --linked-reads (BX / HP tags)" if args.linked else ""),
                       "windows_per_gpu": args.windows, "coverage": [args.cov, cov_n], "reads_per_gpu": int(batch.n_reads),
                       "records_rank0_contig": len(variants), "records_sha256_rank0_contig": h.hexdigest()[:16],
                       "slots_in_flight": n_slots, "batches_in_flight": nfl, "step_ms": step_summary, "settle": settle, "upload_first_ms": round(upload_first_ms, 1), "upload_ms": round(up_ms, 1),
                       "windows_first_graph_in_lds": eng.prebuilt_count(), "graphs_built_ahead": eng.ahead_counts()[0], "graphs_taken_from_pool": eng.ahead_counts()[1],
                       "build_service": dict(zip(("posted", "served", "not_buildable", "taken_back"), eng.svc_counts())),
                       "windows_rerun_worst_case_tier": eng.rerun_count(), "workspace_MB_per_slot": round(slot_bytes / 2 ** 20, 1),
                       "kernel_ms": per_kernel},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(achieved / 8000.0, 6), "traffic": None,
                         "kernel": "+".join(names), "kernel_ms": round(ms_all, 3), "algorithmic_bytes_per_launch": int(alg_bytes),
                         "per_kernel_ms": per_kernel, "per_kernel": kernel_rooflines(alg_build, alg_window, per_kernel),
                         "note": "one pass over the batch = the listed kernels back to back on one stream; achieved = algorithmic bytes of the batch / the sum of their HIP-event durations, measured on launches that have the GPU to themselves (profiles/: rocprofv3 of `bench.py --in-flight 1`)",
                         "pipelined": {"ms_per_step": round(1000.0 * dt / args.steps, 3), "achieved": round(alg_bytes / (dt / args.steps) / 1e9, 3), "frac": round(alg_bytes / (dt / args.steps) / 1e9 / 8000.0, 6)},
                         "peak_measured_copy": 6290.0, "frac_of_measured_copy": round(achieved / 6290.0, 6)},
        }
        if world > 1:
            out["config"]["merged_records_vdb"] = last.get("n", 0)
            out["config"]["vdb_variants"] = last["db"].size() if "db" in last else 0
            out["config"]["gather"] = "sizes all_gather + send/recv to rank 0 (RCCL), replay in window order into lancet_vdb inside the step (second host thread on rank 0)"
            out["rccl_ranks"] = rccl_ranks                     # dist.get_world_size() after init_process_group("nccl")
            out["comm_backend"] = dist.get_backend()
            out["gather_ms_per_step_by_rank"] = gather_ms      # sizes all_gather + payload send / recv, as each rank saw it
            out["replay_ms_per_step_rank0"] = round(1000.0 * tm["replay"] / max(1, args.steps), 3)
            out["pack_ms_per_step_rank0"] = round(1000.0 * tm["pack"] / max(1, args.steps), 3)           # keys + per-key reduction, communication thread
            out["submit_thread_comm_ms_per_step_rank0"] = round(1000.0 * tm["host"] / max(1, args.steps), 3)   # what the submitting thread spends on the gather (hand-over + waits)
            out["windows_per_rank"] = n_local
            if one_gpu:
                out["config"]["comm"] = "LANCET_BENCH_ONE_GPU=1: all ranks on device 0, gather over gloo -- a check of the N-rank path, not a measurement"
        # HBM-side bytes per launch from the committed PMC passes (tools/pmc_total.sh -> profiles/r6_traffic.json): rocprofv3 cannot run
        # inside this process, so the figure is looked up for the exact workload it was taken on -- and only when it was taken on THESE
        # kernels (sha1 over lancet_amd/csrc/*.h, *.hip, recorded with the passes): a record of other kernels is not quoted.
        try:
            fp = workload.kernel_fingerprint()
            for name in ("r6_traffic.json", "r5_traffic.json", "r4_traffic.json", "r3_traffic.json"):
                tj = os.path.join(ROOT, "profiles", name)
                if os.path.exists(tj):
                    break
            with open(tj) as fh:
                for rec in json.load(fh)["measurements"]:
                    if rec["windows"] == args.windows and rec["coverage"] == args.cov and rec.get("coverage_normal", rec["coverage"]) == cov_n \
                            and rec.get("str_fraction", 0.0) == args.str_fraction and bool(rec.get("linked", False)) == bool(args.linked) and world == 1:
                        if rec.get("kernel_fingerprint") != fp:
                            out["roofline"]["traffic_note"] = (f"profiles/{name} holds PMC passes for this workload, but taken on other kernels (fingerprint "
                                                               f"{rec.get('kernel_fingerprint')}, these are {fp}): not quoted; re-run tools/profile_round6.sh")
                            break
                        out["roofline"]["traffic"] = int((rec["FETCH_SIZE_KB"] / rec.get("fetch_calibration", 1.0) + rec["WRITE_SIZE_KB"]) * 1024)
                        out["roofline"]["traffic_fetch_write"] = [int(rec["FETCH_SIZE_KB"] * 1024), int(rec["WRITE_SIZE_KB"] * 1024)]
                        out["roofline"]["traffic_note"] = rec["note"]
                        out["roofline"]["traffic_kernel_fingerprint"] = fp
                        break
        except (OSError, KeyError, ValueError):
            pass
        if args.cpu_sample and world == 1:
            out["cpu_baseline"] = cpu_baseline(batch, params, variants, args.cpu_sample, args.cpu_sample_all)
        for e2 in engs:
            e2.close()
        if world == 1 and not args.no_bam:
            # BASELINE config 2's 5 Mb contig (49 981 windows; every record and the VCF of this very input against the oracle:
            # tools/e2e_parity.py, profiles/r6_e2e_parity_5mb.json)
            bam = bam_e2e(os.path.join(ROOT, "build", "scan5m"), ["--reg", "chr22:1000-4999000"], [5000000, 30, 30, 14], "5 Mb contig, 30x/30x, 2x150 bp")
            if bam:
                out["value_bam_e2e"] = bam.pop("value")       # BAM -> VCF by the native program, a whole process: never `value`
                out["value_bam_e2e_detail"] = bam
            # BASELINE config 3 as a one-GPU proxy: 24 contigs through --bed -- two engines on the GPU, and the N-process route with one rank
            # (lancet_gpu --ranks 1: pack, RCCL gather, merge on rank 0; bringing the communicator up is a second and a half of every such run)
            d3 = os.path.join(ROOT, "build", "scan24x100k")
            c3 = bam_e2e(d3, ["--bed", "{d}/regions.bed"], [100000, 30, 30, 14, 24], "24 contigs x 100 kb, 30x/30x, 2x150 bp")
            if c3:
                c3["name"] = "config 3 proxy on one GPU: 24 contigs (23 544 windows), lancet_gpu --bed; the 24 x 1 Mb run (239 664 windows; engines / batch sizes / ranks / every record against the oracle): profiles/r6_config3_proxy.txt"
                c3["windows_per_s"] = c3.pop("value")
                r1 = bam_e2e(d3, ["--bed", "{d}/regions.bed"], [100000, 30, 30, 14, 24], "24 contigs x 100 kb", extra=("--ranks", "1", "--batch-windows", "8192"))
                if r1:
                    c3["n_process_route_one_rank"] = {"windows_per_s": r1["value"], "wall_s_all_runs": r1["wall_s_all_runs"], "vcf_md5_equal": r1["vcf_md5"] == c3["vcf_md5"],
                                                      "what": "the same input through lancet_gpu --ranks 1: records packed (keys + reduction), gathered over RCCL, replayed on rank 0"}
                out["config3_proxy"] = c3
        if world == 1 and not args.no_configs:
            out["configs"] = ([out.pop("config3_proxy")] if "config3_proxy" in out else []) + [
                side_config(engine.Engine, params, "config 2 at 60x/60x", 8192, 60.0, 60.0, 8),
                side_config(engine.Engine, params, "config 4: 100x tumor / 40x normal, 30 % STR + 5 % low complexity", 4096, 100.0, 40.0, 8,
                            str_fraction=0.30, lowcomplex_fraction=0.05),
                side_config(engine.Engine, abi.default_params(lr_mode=1), "config 5: --linked-reads (BX / HP tags on every pair), 30x/30x", 16384, 30.0, 30.0, 6,
                            streaming=True, linked=True),
            ]
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
