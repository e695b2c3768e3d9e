"""Window tiler / read selection (host callers of the hot path), checked against facts established with the
reference binary (SURVEY.md §8(d) window arithmetic) and against the golden traces' read counts."""
import re

import golden_util as gu
from lancet_amd import frontend


def test_tiler_matches_reference_window_arithmetic():
    seq = "ACGT" * 10000
    w = frontend.tile_region(seq, "chr22", "chr22:1000-29000")
    assert len(w) == 281                                   # verified with the reference in SURVEY.md
    assert all(len(x.seq) == 600 for x in w[:-1])
    assert w[0].hdr == "chr22:750-1350"
    assert len(w[-1].seq) == (29250 - 750 + 1) - 28000 - 1
    one = frontend.tile_region(seq, "chr22", "chr22:2200-2799", padding=0)
    assert len(one) == 1 and len(one[0].seq) == 599 and one[0].hdr == "chr22:2200-2799"


def test_tiler_uppercases_and_masks_iupac():
    seq = "acgtRYKM" * 200
    w = frontend.tile_region(seq, "c", "c:1-700", padding=0)
    assert set(w[0].seq) <= set("ACGTN")


def test_processing_order_is_lexicographic_on_header():
    seq = "ACGT" * 5000
    w = frontend.tile_region(seq, "chr22", "chr22:500-12000")
    o = frontend.windows_in_processing_order(w)
    assert [x.hdr for x in o] == sorted(x.hdr for x in w)
    assert o != w                                           # "chr22:1050-.." sorts before "chr22:250-.."


def test_read_selection_counts_match_reference_trace():
    """`numsequences` printed by the reference for every window == reads our front-end selects."""
    for case in ("tile30", "err_hi"):
        meta, batch, kept, _ = gu.case_batch(case)
        want = {m.group(1): int(m.group(2)) for m in
                re.finditer(r"== Processing \d+: (\S+) numsequences: (\d+)", gu.golden_trace(case))}
        got = {h: int(batch.read_begin[i + 1] - batch.read_begin[i]) for i, h in enumerate(batch.hdr)}
        assert got == want
