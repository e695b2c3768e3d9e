"""Helpers shared by the parity tests: load tests/golden fixtures and turn them into window batches."""
from __future__ import annotations

import functools
import json
import os
import re

import numpy as np

from lancet_amd import frontend
from lancet_amd.synth import SamRead

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(f[:-5] for f in os.listdir(GOLDEN) if f.endswith(".json"))
LR_CASES = [c for c in CASES if "--linked-reads" in json.load(open(os.path.join(GOLDEN, c + ".json")))["flags"]]

_DIGEST = re.compile(
    r"^(== Processing|Repeat in reference|Near-perfect repeat|reads: |  \d+: nodes:| nodes: |ref trim5|"
    r"Ambiguous match|No match to reference|Cycle found|compressing graph|  removing |removing low coverage|"
    r"remove tips round| removed|remove short links| Found |FINISHED|>p_| refcomp:| perfect:|"
    r"searching from|Missing source|WARNING: DFS_LIMIT)")


def digest_trace(text: str) -> str:
    return "\n".join(l.rstrip() for l in text.splitlines() if _DIGEST.match(l)) + "\n"


def load_case(name: str):
    meta = json.load(open(os.path.join(GOLDEN, f"{name}.json")))
    z = np.load(os.path.join(GOLDEN, f"{name}.reads.npz"))
    ref = str(z["ref"])
    rname = str(z["rname"])
    reads = {}
    for rg in ("tumor", "normal"):
        a = {k: z[f"{rg}_{k}"].tolist() for k in ("qname", "flag", "pos", "mapq", "cigar", "seq", "qual", "as", "xs", "md")}
        reads[rg] = [SamRead(a["qname"][i], a["flag"][i], rname, a["pos"][i], a["mapq"][i], a["cigar"][i], a["seq"][i],
                             a["qual"][i], {"AS": a["as"][i], "XS": a["xs"][i], "MD": a["md"][i]})
                     for i in range(len(a["qname"]))]
        if f"{rg}_bx" in z.files:
            bx, hp = z[f"{rg}_bx"].tolist(), z[f"{rg}_hp"].tolist()
            for i, r in enumerate(reads[rg]):
                if bx[i] != "":
                    r.tags["BX"] = bx[i]
                if hp[i] >= 0:
                    r.tags["HP"] = hp[i]
    return meta, ref, rname, reads


_ENGINE_FLAGS = {   # reference command-line flag -> lancet_params field (reference src/Lancet.cc:655-800, src/Microassembler.cc:726-753)
    "--min-k": "min_k", "--max-k": "max_k", "--tip-len": "max_tip_len", "--cov-thr": "cov_threshold", "--cov-ratio": "min_cov_ratio",
    "--low-cov": "low_cov_threshold", "--dfs-limit": "dfs_limit", "--max-indel-len": "max_indel_len", "--max-mismatch": "max_mismatch",
    "--max-unit-length": "max_unit_len", "--min-report-unit": "min_report_units", "--min-report-len": "min_report_len",
    "--dist-from-str": "dist_from_str", "--trim-lowqual": "min_qual_trim", "--min-base-qual": "min_qual_call"}


def _flag_values(meta):
    flags = [f for f in meta["flags"] if f not in ("--linked-reads", "--active-region-on")]
    return {flags[i]: flags[i + 1] for i in range(0, len(flags), 2)}


def case_params(meta):
    """reference CLI flags of the case -> (padding, min_k, max_k)."""
    opt = {"--padding": 250, "--min-k": 11, "--max-k": 101}
    opt.update(_flag_values(meta))
    return int(opt["--padding"]), int(opt["--min-k"]), int(opt["--max-k"])


def case_window(meta) -> int:
    """--window-size of the case (reference default 600, src/Lancet.cc:662)."""
    return int(_flag_values(meta).get("--window-size", 600))


def params(meta, **over):
    """lancet_params of the case: reference defaults, the case's graph / STR / quality flags, lr_mode."""
    from lancet_amd import abi
    kw = {}
    for flag, val in _flag_values(meta).items():
        if flag not in _ENGINE_FLAGS:
            continue
        field = _ENGINE_FLAGS[flag]
        kw[field] = float(val) if field == "min_cov_ratio" else int(val) + (33 if field in ("min_qual_trim", "min_qual_call") else 0)
    kw["lr_mode"] = int(case_lr(meta))
    kw.update(over)
    return abi.default_params(**kw)


def case_lr(meta) -> bool:
    return "--linked-reads" in meta["flags"]


def case_active_region(meta) -> bool:
    """goldens are reference runs with --active-region-off unless the case carries this marker"""
    return "--active-region-on" in meta["flags"]


@functools.lru_cache(maxsize=None)
def case_batch(name: str):
    meta, ref, rname, reads = load_case(name)
    padding, min_k, max_k = case_params(meta)
    windows = frontend.tile_region(ref, rname, meta["region"], padding=padding, window_size=case_window(meta))
    batch, kept = frontend.batch_from_sam(windows, reads["tumor"], reads["normal"], max_k=max_k, linked=case_lr(meta),
                                          active_region=case_active_region(meta))
    return meta, batch, kept, (min_k, max_k)


def golden_vcf(name: str) -> str:
    return open(os.path.join(GOLDEN, f"{name}.vcf")).read()


def golden_trace(name: str) -> str:
    return open(os.path.join(GOLDEN, f"{name}.trace.txt")).read()


def load_batches_npz(name: str):
    """Window batches stored as arrays (tools/make_ahead_fixture.py): [WindowBatch, ...]"""
    from lancet_amd import frontend
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    fields = ("chr_id", "ref_start", "ref_off", "ref_bases", "read_begin", "seq_off", "seq", "qual", "label", "strand", "mate", "mapped", "name_rank")
    out, i = [], 0
    while f"p{i}_hdr" in z:
        kw = {f: z[f"p{i}_{f}"] for f in fields}
        hdr = [str(x) for x in z[f"p{i}_hdr"]]
        out.append(frontend.WindowBatch(n_windows=len(hdr), hdr=hdr, chrom=[str(x) for x in z[f"p{i}_chrom"]], **kw))
        i += 1
    return out
