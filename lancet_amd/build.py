"""Builds lancet_amd/csrc/liblancet_engine.so (HIP kernels + C-ABI + host VariantDB + native host front end) for gfx950
and the command-line program lancet_amd/bin/lancet_gpu, in-tree."""
from __future__ import annotations

import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "liblancet_engine.so")
SOURCES = ["engine.hip", "window_fat.hip", "host_vdb.cc", "host_frontend.cc", "host_trace.cc", "host_gather.cc", "lancet_main.cc"]
BIN = os.path.join(os.path.dirname(CSRC), "bin", "lancet_gpu")
HEADERS = ["kernels.h", "build_lds.h", "build_lds_impl.h", "wave.h", "layout.h", "host_common.h", "host_pack.h", os.path.join("..", "..", "include", "lancet_engine.h"),
           os.path.join("..", "..", "include", "lancet_host.h"), os.path.join("..", "..", "include", "lancet_gather.h")]


def needs_build() -> bool:
    if not os.path.exists(LIB) or not os.path.exists(BIN):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def device_flags() -> list:
    """Compile flags of the two .hip files (part of workload.kernel_fingerprint).  HIPCC_EXTRA: -D knobs of tuning builds."""
    return (["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result"]
            + os.environ.get("HIPCC_EXTRA", "").split())


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # -ffp-contract=off: float coverage averaging must round like the reference's SSE code
    steps = [
        [hipcc] + device_flags() + ["-c", "engine.hip", "-o", "engine.o"],
        [hipcc] + device_flags() + ["-c", "window_fat.hip", "-o", "window_fat.o"],
        [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-c", "host_vdb.cc", "-o", "host_vdb.o"],
        [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-pthread", "-c", "host_frontend.cc", "-o", "host_frontend.o"],
        [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-c", "host_trace.cc", "-o", "host_trace.o"],
        # the multi-process record gather: HIP runtime API + RCCL's header only (librccl is loaded on first use, not linked)
        [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-pthread", "-I/opt/rocm/include", "-c", "host_gather.cc", "-o", "host_gather.o"],
        [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "engine.o", "window_fat.o", "host_vdb.o", "host_frontend.o", "host_trace.o", "host_gather.o", "-lz", "-lpthread", "-ldl", "-o", LIB],
        # the reference's command line on the native host side; finds the library next to itself
        [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-pthread", "lancet_main.cc", "-o", BIN, "-L.", "-llancet_engine",
         "-Wl,-rpath,$ORIGIN/../csrc", "-Wl,-rpath-link,/opt/rocm/lib"],
    ]
    os.makedirs(os.path.dirname(BIN), exist_ok=True)
    for cmd in steps:
        r = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError("build of liblancet_engine.so failed: " + " ".join(cmd))
        if verbose:
            sys.stderr.write(r.stderr)
    _check_gfx950(LIB)
    return LIB


def _check_gfx950(lib: str) -> None:
    """The device code object must be gfx950 (a silently defaulted arch would only fail on the GPU box)."""
    llvm = "/opt/rocm/lib/llvm/bin"
    if not os.path.exists(os.path.join(llvm, "clang-offload-bundler")):
        return
    fb = lib + ".fatbin"
    try:
        subprocess.run([os.path.join(llvm, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fb], check=True)
        out = subprocess.run([os.path.join(llvm, "clang-offload-bundler"), "--list", "--type=o", "--input=" + fb],
                             capture_output=True, text=True).stdout
    finally:
        if os.path.exists(fb):
            os.remove(fb)
    if "gfx950" not in out:
        raise RuntimeError(f"liblancet_engine.so has no gfx950 code object (found: {out.split()})")


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
