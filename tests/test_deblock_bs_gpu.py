"""GPU parity of the boundary-strength stage (vvc355_deblock_bs_pass: one lane per 4x4 unit gathers its bS / max filter length
entries for both edge directions) vs the oracle's restatement of vvc_deblock_bs (vvc_filter.c:308-783, scatter per transform
unit) on the same synthetic side tables (tests/bs_cases.py)."""
import ctypes

import numpy as np
import pytest

import bs_cases
from ffvvc_amd import batch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cfg", [
    dict(width=328, height=200, ctb_log2=6, n_slices=1, tiles=False, lfase=1, lfate=1),
    dict(width=328, height=200, ctb_log2=6, n_slices=3, tiles=True, lfase=0, lfate=0),
    dict(width=416, height=240, ctb_log2=7, n_slices=4, tiles=True, lfase=1, lfate=0),
    dict(width=416, height=240, ctb_log2=7, n_slices=4, tiles=True, lfase=0, lfate=1),
    dict(width=264, height=136, ctb_log2=5, n_slices=5, tiles=True, lfase=0, lfate=0),
    dict(width=328, height=200, ctb_log2=6, n_slices=3, tiles=True, lfase=0, lfate=0, hs=1, vs=0),
    dict(width=328, height=200, ctb_log2=6, n_slices=2, tiles=False, lfase=1, lfate=1, hs=0, vs=0),
    dict(width=1920, height=1080 - 1080 % 8, ctb_log2=7, n_slices=2, tiles=False, lfase=1, lfate=1, inter_frac=0.95),
])
def test_deblock_bs_pass(dev, orc, cfg):
    rng = np.random.default_rng(0x5EED0B50 + cfg["width"] + cfg["n_slices"] + 2 * cfg["lfase"] + cfg["lfate"])
    t = bs_cases.BsTables(rng, **cfg)
    want = bs_cases.run_oracle(orc, t)
    for name in t.OUT:
        getattr(t, name)[:] = 0xEE               # the device must write every entry itself
    bufs = {name: batch.DeviceBuffer.from_host(getattr(t, name)) for name in t.IN + t.OUT}
    f = t.frame(lambda name: bufs[name].ptr)
    d_f = batch.DeviceBuffer.from_host(np.frombuffer(bytes(f), np.uint8))
    dev.vvc355_deblock_bs_pass(None, d_f.ptr, ctypes.addressof(f))
    dev.vvc355_stream_sync(None)
    for name in t.OUT:
        got = bufs[name].to_host(np.uint8, want[name].shape)
        bad = np.argwhere(got != want[name])
        assert len(bad) == 0, f"{name}: {len(bad)} entries differ, first at (row, col) {bad[0].tolist()}: got {got[tuple(bad[0])]}, want {want[name][tuple(bad[0])]}"
    # the case exercises every rule: all three strengths and the long filter lengths occur
    assert set(np.unique(want["bs10"])) == {0, 1, 2} and {1, 2, 3, 5, 7} <= set(np.unique(want["p1"]))
