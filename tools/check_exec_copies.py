#!/usr/bin/env python3
"""Looks through the compiled gfx950 kernels for a miscompile ROCm 7.2's backend produced in round 4: register-to-register VALU copies
placed at the top of a join block IN FRONT OF the `s_or_b64 exec, exec, ...` that re-enables the lanes which skipped the region --
those lanes (or a whole wave that branched around the region with EXEC = 0) never execute the copy and go on with a stale register
(bl_large::bl_build_window: the EngineCaps pointer, a memory fault on the next load through it).  Constant moves in that place are the
ordinary lowering of a phi and are not reported.
usage: check_exec_copies.py [engine.hip ...]   (default: every .hip of lancet_amd/csrc); exit code 1 when something is found."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "lancet_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result", "--cuda-device-only", "-S"]


def scan(asm: str):
    lines = asm.split("\n")
    func, hits = None, []
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+|\w+):\s", l + " ")
        if m and not l.startswith(".L"):
            func = m.group(1)
        if not re.match(r"^\.LBB\d+_\d+:", l):
            continue
        j, copies = i + 1, []
        while j < len(lines) and lines[j].startswith("\t"):
            t = lines[j].strip()
            if re.match(r"v_mov_b(32|64)_e32 v[\[\d:\]]+, v[\[\d:\]]+$", t) or t.startswith("v_accvgpr_read"):
                copies.append(t); j += 1; continue
            if t.startswith("v_mov_b"):          # a constant: phi lowering
                j += 1; continue
            break
        if copies and j < len(lines) and re.match(r"\ts_or_b64 exec, exec", lines[j]):
            hits.append((func, l.split(":")[0], copies))
    return hits


def main():
    srcs = sys.argv[1:] or [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")]
    bad = 0
    for src in srcs:
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "k.s")
            r = subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + FLAGS + [src, "-o", out], cwd=CSRC, capture_output=True, text=True)
            if r.returncode != 0:
                sys.stderr.write(r.stderr); return 2
            for func, label, copies in scan(open(out).read()):
                name = subprocess.run(["c++filt", func or "?"], capture_output=True, text=True).stdout.strip()[:100]
                print(f"{os.path.basename(src)}: {name}: {label}: {'; '.join(copies)}  -- in front of an EXEC restore")
                bad += 1
    print("check_exec_copies:", "nothing found" if not bad else f"{bad} place(s) to look at")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
