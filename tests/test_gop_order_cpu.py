"""bench.gop_order: the decoding order of a hierarchical-B group of pictures (the random-access structure the bench's default step decodes)
— pure host logic: every picture once, every picture after the two pictures it predicts from, the two references on either side of it."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


@pytest.mark.parametrize("g", [2, 4, 8, 16, 32])
def test_gop_order(g):
    order = bench.gop_order(g)
    assert sorted(p for p, _lo, _hi in order) == list(range(1, g + 1))
    seen = {0}                                  # POC 0 is the previous group's last picture
    for poc, lo, hi in order:
        assert lo in seen and hi in seen, (poc, lo, hi)
        if poc == g:
            assert (lo, hi) == (0, 0)           # the group's anchor predicts from the previous anchor on both lists
        else:
            assert lo < poc < hi and poc - lo == hi - poc
        seen.add(poc)
    # coarsest level first: the temporal distance to the references never grows along the order
    dist = [hi - lo for poc, lo, hi in order if poc != g]
    assert dist == sorted(dist, reverse=True)
    assert bench.REF_FREE_STAGES <= {"inter_mvf_fill", "inter_job_build", "itx_job_build", "intra_tb_dequant_lfnst_itx", "side_tables_fill", "deblock_bs", "alf_job_build"}
