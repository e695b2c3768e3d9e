"""Full-size parity of a BAM -> VCF scan (the 5 Mb contig, 49 981 windows): the native program against the oracle.

    python tools/e2e_parity.py build/scan5m chr22:1000-4999000 [--batch-windows 8192] [--active-region-off] [--procs N]

TEST INFRASTRUCTURE (the oracle is the checker here, never the product).  Three routes over the SAME inputs:
  1. `lancet_amd/bin/lancet_gpu` (native host side + engine + native VariantDB): the VCF a user gets;
  2. the native host side's batches (lancet_host_batch, the batches route 1 assembles) through the ENGINE on the GPU -> records;
  3. the same batches through the ORACLE (oracle/liblancet_oracle.so fanned out over processes: oracle/cpu_fanout.py) -> records,
     replayed into oracle/vcf_oracle.py (the Python restatement of Variant_t / VariantDB_t / the VCF writer, reference
     src/VariantDB.cc:28-91, 154-179, pinned on the reference's own VCFs in tests/test_oracle_golden.py).
Compared: every record of 2 against 3, field by field, in (window, emission) order -- all windows, not a sample; the VCF of route 3
against route 1's byte for byte (without the ##fileDate / ##cmdline / ##reference lines, which name the run); route 2's records
replayed through the native VariantDB against both.  Prints one JSON line; exit code 1 on any difference."""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _body(text: str) -> str:
    return "".join(l + "\n" for l in text.splitlines()
                   if not l.startswith("##fileDate") and not l.startswith("##cmdline") and not l.startswith("##reference"))


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("dir"); ap.add_argument("region", help="chr:start-end, or a BED file (name ends in .bed): --bed")
    ap.add_argument("--batch-windows", type=int, default=8192)
    ap.add_argument("--active-region-off", action="store_true")
    ap.add_argument("--procs", type=int, default=0)
    ap.add_argument("--max-windows", type=int, default=0, help="stop after this many tiled windows (0: all) -- a bounded run for the test suite")
    a = ap.parse_args()
    from lancet_amd import abi, build, engine, host
    from oracle import cpu_fanout, vcf_oracle
    if os.path.isdir(a.dir):
        T, N, F = (os.path.join(a.dir, f) for f in ("tumor.bam", "normal.bam", "ref.fa"))
    else:                                                            # a prefix: the golden fixtures' naming (tests/golden/<case>.tumor.bam ...)
        T, N, F = a.dir + ".tumor.bam", a.dir + ".normal.bam", a.dir + ".fa"
    procs = a.procs
    if not procs:
        procs = os.cpu_count() or 1
        try:
            procs = min(procs, len(os.sched_getaffinity(0)))
            txt = open("/sys/fs/cgroup/cpu.max").read().split()
            if txt[0] != "max":
                procs = max(1, min(procs, 2 * int(round(int(txt[0]) / int(txt[1])))))
        except (OSError, ValueError, AttributeError, IndexError):
            pass
    out = {"inputs": a.dir, "region": a.region, "batch_windows": a.batch_windows, "active_region": not a.active_region_off, "oracle_processes": procs}
    # ---- route 1: the program
    t0 = time.time()
    bed = a.region.endswith(".bed")
    cmd = [build.BIN, "--tumor", T, "--normal", N, "--ref", F, "--bed" if bed else "--reg", a.region, "--batch-windows", str(a.batch_windows)]
    if a.active_region_off:
        cmd.append("--active-region-off")
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode not in (0,):
        sys.stderr.write(r.stderr[-2000:])
        print(json.dumps(dict(out, error=f"lancet_gpu exit code {r.returncode}")))
        return 1
    vcf1 = _body(r.stdout)
    out["lancet_gpu_s"] = round(time.time() - t0, 2)
    # ---- routes 2 and 3: the same batches through the engine and through the oracle
    p = abi.default_params()
    o = host.default_opts(active_region=0 if a.active_region_off else 1)
    nh = host.NativeHost(T, N, F)
    hdrs = nh.tile_regions([], o, bed=a.region) if bed else nh.tile(a.region, o)
    if not (nh.first_has_md(True) or nh.first_has_md(False)):
        o.active_region = 0                                          # (main() turns the module off: reference src/Lancet.cc:817-825)
    nwin = len(hdrs) if not a.max_windows else min(len(hdrs), a.max_windows)
    eng = engine.Engine(p, device=0)
    db_native = engine.VariantDB()
    db_oracle = vcf_oracle.VariantDB()
    n_rec = n_kept = n_diff_batches = 0
    t_eng = t_ora = 0.0
    first_diff = None
    for w0 in range(0, nwin, a.batch_windows):
        w1 = min(nwin, w0 + a.batch_windows)
        batch, idx = nh.batch(w0, w1, o)
        if batch.n_windows == 0:
            continue
        n_kept += batch.n_windows
        t = time.time(); ev, st = eng.process(batch); t_eng += time.time() - t
        t = time.time()
        per = max(4, batch.n_windows // (procs * 6))
        _, _, done, ov = cpu_fanout.run(batch, {}, batch.n_windows, procs, per, timeout_s=1800.0, want_records=True)
        t_ora += time.time() - t
        if done != batch.n_windows or ov != ev:
            n_diff_batches += 1
            if first_diff is None:
                for i in range(max(len(ov), len(ev))):
                    x = ov[i] if i < len(ov) else None; y = ev[i] if i < len(ev) else None
                    if x != y:
                        first_diff = {"batch_from_window": w0, "record": i, "oracle": x, "engine": y}
                        break
        bad = [s for s in st if s["status"] not in (0, 1, 2)]          # OK, NO_READS, K_EXHAUSTED
        if bad:
            out.setdefault("windows_not_assembled", 0); out["windows_not_assembled"] += len(bad)
        n_rec += len(ev)
        chroms = nh.chroms()
        db_native.add_records(ev, chroms)
        for rec in ov:
            db_oracle.add(vcf_oracle.Variant(batch.chrom[rec["window"]], rec))
    sn, stt = nh.sample(False), nh.sample(True)
    vcf3 = _body(db_oracle.vcf(sample_n=sn, sample_t=stt))
    vcf2 = _body(db_native.vcf(sample_normal=sn, sample_tumor=stt))
    body = lambda s: "".join(l + "\n" for l in s.splitlines() if not l.startswith("##"))
    out.update({
        "windows_tiled": nwin, "windows_assembled": n_kept, "records_compared": n_rec, "batches_with_different_records": n_diff_batches,
        "records_identical": n_diff_batches == 0,
        "vcf_lines": sum(1 for l in vcf1.splitlines() if not l.startswith("#")),
        "vcf_oracle_equals_lancet_gpu": body(vcf3) == body(vcf1),
        "vcf_engine_records_native_vdb_equals_lancet_gpu": body(vcf2) == body(vcf1) if not a.max_windows else None,
        "vcf_header_lines_identical": [l for l in vcf3.splitlines() if l.startswith("##")] == [l for l in vcf1.splitlines() if l.startswith("##")],
        "vcf_md5_lancet_gpu": hashlib.md5(vcf1.encode()).hexdigest(), "vcf_md5_oracle": hashlib.md5(vcf3.encode()).hexdigest(),
        "engine_s": round(t_eng, 2), "oracle_s": round(t_ora, 2),
    })
    if a.max_windows:
        out["vcf_oracle_equals_lancet_gpu"] = None               # (the program ran the whole region)
        out["vcf_oracle_equals_engine_records_native_vdb"] = body(vcf3) == body(vcf2)
    if first_diff:
        out["first_difference"] = first_diff
    print(json.dumps(out))
    ok = out["records_identical"] and (out["vcf_oracle_equals_lancet_gpu"] is not False) and (out["vcf_engine_records_native_vdb_equals_lancet_gpu"] is not False) \
        and out.get("vcf_oracle_equals_engine_records_native_vdb", True)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
