"""Builds lancet_amd/csrc/liblancet_engine.so (HIP kernels + C-ABI + host VariantDB) for gfx950, in-tree."""
from __future__ import annotations

import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "liblancet_engine.so")
SOURCES = ["engine.hip", "host_vdb.cc"]
HEADERS = ["kernels.h", "wave.h", "layout.h", "host_common.h", os.path.join("..", "..", "include", "lancet_engine.h")]


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # -ffp-contract=off: float coverage averaging must round like the reference's SSE code
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
           "-Wno-unused-result", "-x", "hip", "engine.hip", "-x", "c++", "host_vdb.cc", "-o", LIB]
    r = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed building liblancet_engine.so")
    if verbose:
        sys.stderr.write(r.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
