"""CPU-side checks of the drop-in boundary: the shared library loads, exports every entry point declared in
include/lancet_engine.h, fails loudly without a GPU, and its host half (VariantDB/VCF) reproduces the
reference VCFs from oracle records."""
import ctypes
import os
import re

import pytest

import golden_util as gu
from lancet_amd import abi, build, engine
from oracle import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build.build()
    return engine.lib()


def test_library_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "lancet_engine.h")).read() + open(os.path.join(ROOT, "include", "lancet_host.h")).read() + open(os.path.join(ROOT, "include", "lancet_gather.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(lancet_[a-z_]+)\s*\(", hdr))
    assert len(names) >= 25 and "lancet_host_batch" in names
    for n in names:
        assert hasattr(lib, n), n


def test_library_carries_every_gfx950_kernel(lib):
    """The device code of both translation units is in the shipped library: the LDS build kernel in its two sizes, the
    window kernel on one wave and on several (the re-run tier), ordering and read preparation."""
    blob = open(build.LIB, "rb").read()
    for k in (b"build_kernel", b"build_kernel_large", b"svc_kernel", b"svc_kernel_large", b"order_class_kernel", b"order_place_kernel", b"prep_kernel", b"window_kernel", b"window_kernel_fat"):
        assert re.search(rb"_Z\d+" + k + rb"[Pi]", blob), k
    assert b"gfx950" in blob


def test_struct_layouts_match_header(lib):
    assert ctypes.sizeof(abi.LancetParams) == 72
    assert ctypes.sizeof(abi.LancetVariant) == 64
    assert ctypes.sizeof(abi.LancetWindowStats) == 32
    assert ctypes.sizeof(abi.LancetPackedReads) == 72 and abi.LancetPackedReads.struct_size.offset == 0      # (the size travels in the first field)
    p = abi.LancetParams()
    lib.lancet_params_default(ctypes.byref(p))
    d = abi.default_params()
    assert bytes(p) == bytes(d)


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(engine.EngineError, match="no HIP device"):
        engine.Engine()
    h = ctypes.c_void_p()
    p = abi.default_params(max_k=200)
    assert lib.lancet_engine_create(ctypes.byref(p), 0, ctypes.byref(h)) == -4      # LANCET_E_UNSUPPORTED (k > 127)
    p = abi.default_params(lr_mode=1)
    assert lib.lancet_engine_create(ctypes.byref(p), 0, ctypes.byref(h)) == -2      # LANCET_E_NO_DEVICE: no CPU path in any mode
    p = abi.default_params(min_k=12)
    assert lib.lancet_engine_create(ctypes.byref(p), 0, ctypes.byref(h)) == -2      # even k is accepted (golden `evenk`); still no CPU path


@pytest.mark.parametrize("case", gu.CASES)
def test_host_variantdb_and_vcf_writer_reproduce_reference_vcf(case, lib):
    meta, batch, kept, (min_k, max_k) = gu.case_batch(case)
    lr = gu.case_lr(meta)
    records, _, _ = oracle.run(batch, gu.params(meta))
    id2chr = {}
    for c, i in zip(batch.chrom, batch.chr_id):
        id2chr[int(i)] = c
    db = engine.VariantDB()
    db.add_records(records, [id2chr[i] for i in range(len(id2chr))], bx_names=batch.bx_names if lr else None)
    assert db.vcf() == gu.golden_vcf(case)
    full = db.vcf(cmdline="lancet --x", reference="ref.fa", date_line="Sun Sep 27 05:27:00 2026\n")
    assert "##fileDate=Sun Sep 27 05:27:00 2026\n##source=lancet 1.1.0" in full and "##cmdline=lancet --x\n##reference=ref.fa\n" in full


@pytest.mark.parametrize("name", ["tile30", "lr30", "dups"])
def test_native_trace_formatter_matches_the_python_one(lib, name):
    """lancet_trace_format (host C++, what `lancet_gpu -v` prints) against lancet_amd/trace.py -- itself pinned on the
    reference's -v output by the golden traces -- on the event streams of the emulated kernels: every window, byte for byte."""
    import numpy as np
    import golden_util as gu
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import emu
    from lancet_amd import trace
    meta, batch, kept, (min_k, max_k) = gu.case_batch(name)
    p = abi.default_params(min_k=min_k, max_k=max_k, lr_mode=int(gu.case_lr(meta)))
    _, _, text = emu.run(batch, p, evt_cap=1 << 17)
    assert len(emu.LAST_EVENTS) == batch.n_windows and len(text) > 1000
    got = []
    for w, words in enumerate(emu.LAST_EVENTS):
        words = np.ascontiguousarray(words, dtype=np.uint32)
        end = int(batch.ref_start[w]) + int(batch.ref_off[w + 1] - batch.ref_off[w])
        ptr = lib.lancet_trace_format(words.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), len(words), w + 1, batch.hdr[w].encode(),
                                      batch.chrom[w].encode(), int(batch.ref_start[w]), end, 1000000)
        assert ptr
        got.append(ctypes.string_at(ptr).decode())
        lib.lancet_free(ptr)
    assert "".join(got) == text


def test_no_register_copies_in_front_of_an_exec_restore():
    """A miscompile of ROCm 7.2's backend met in round 4 (tools/check_exec_copies.py): live-range copies of the register allocator placed
    at the top of a join block BEFORE `s_or_b64 exec` -- waves that skipped the region went on with a stale register (a memory fault in
    the 1024-lane build kernel on windows with fewer nodes than lanes).  The compiled kernels of every .hip are scanned for that shape.
    The scan compiles every .hip to assembly (a minute): skipped without hipcc, and its verdict is kept under .pytest_cache for as long as
    the kernel sources do not change."""
    import hashlib, shutil, subprocess, sys
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which(hipcc)):
        pytest.skip("no hipcc here")
    csrc = os.path.join(ROOT, "lancet_amd", "csrc")
    h = hashlib.sha1()
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode()); h.update(open(os.path.join(csrc, f), "rb").read())
    h.update(open(os.path.join(ROOT, "tools", "check_exec_copies.py"), "rb").read())
    stamp = os.path.join(ROOT, ".pytest_cache", "exec_copies_ok")
    if os.path.exists(stamp) and open(stamp).read().strip() == h.hexdigest():
        return
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_exec_copies.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    os.makedirs(os.path.dirname(stamp), exist_ok=True)
    open(stamp, "w").write(h.hexdigest())
