"""The kernel source (lancet_amd/csrc/kernels.h) compiled for the host with the wave emulator
(tests/emu, test infrastructure) must already agree with the oracle and the reference goldens: this is how the
graph logic is debugged on a GPU-less box.  The GPU parity tests are in test_engine_gpu.py."""
import os
import sys

import pytest

import golden_util as gu
from lancet_amd import abi
from oracle import oracle

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
import emu  # noqa: E402


@pytest.mark.parametrize("case", gu.CASES)
def test_emulated_kernels_match_oracle_and_reference_trace(case):
    meta, batch, kept, (min_k, max_k) = gu.case_batch(case)
    p = abi.default_params(min_k=min_k, max_k=max_k)
    v, st, tr = emu.run(batch, p, evt_cap=1 << 17)
    ov, ost, _ = oracle.run(batch, p)
    assert v == ov
    key = lambda s: (s["status"], s["final_k"], s["n_builds"], s["n_variants"], s["n_kmers"], s["max_nodes"])
    assert [key(s) for s in st] == [key(s) for s in ost]
    assert gu.digest_trace(tr) == gu.golden_trace(case)
