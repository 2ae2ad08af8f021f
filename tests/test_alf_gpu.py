"""GPU parity: ALF slots through the C ABI vs the CPU oracle, bit-exact (integer pixel work)."""
import numpy as np
import pytest

import alf_cases as ac
from conftest import P

pytestmark = pytest.mark.gpu

SIZES = [(4, 4), (8, 4), (4, 8), (16, 16), (32, 8), (64, 64), (128, 32), (100, 60), (128, 128), (124, 128), (128, 92)]


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_alf_filter_luma(dev, orc, bd):
    rng = np.random.default_rng(0x5EED0001 + bd)
    for (w, h) in SIZES:
        for vb_pos in (124, h - 4, h, 8, 1000):
            src, off = ac.make_src(rng, bd)
            coeff, clip = ac.luma_params(rng, bd, w, h)
            want = ac.run_filter(orc, "orc_", "luma", bd, src, off, w, h, coeff, clip, vb_pos)
            got = ac.run_filter(dev, "vvc355_", "luma", bd, src, off, w, h, coeff, clip, vb_pos)
            assert np.array_equal(got, want), f"bd={bd} {w}x{h} vb={vb_pos}"


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_alf_filter_chroma(dev, orc, bd):
    rng = np.random.default_rng(0x5EED0002 + bd)
    for (w, h) in SIZES:
        for vb_pos in (62, h - 2, h, 4, 1000):
            src, off = ac.make_src(rng, bd)
            coeff = rng.integers(-128, 128, size=6).astype(np.int16)
            clip = ac.clip_values(bd)[rng.integers(0, 4, size=6)].copy()
            want = ac.run_filter(orc, "orc_", "chroma", bd, src, off, w, h, coeff, clip, vb_pos)
            got = ac.run_filter(dev, "vvc355_", "chroma", bd, src, off, w, h, coeff, clip, vb_pos)
            assert np.array_equal(got, want), f"bd={bd} {w}x{h} vb={vb_pos}"


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_alf_classify(dev, orc, bd):
    rng = np.random.default_rng(0x5EED0003 + bd)
    for (w, h) in SIZES:
        for vb_pos in (124, h - 4, 8, 1000):
            for smooth in (False, True):
                src, off = ac.make_src(rng, bd, smooth)
                want = ac.run_classify(orc, "orc_", bd, src, off, w, h, vb_pos)
                got = ac.run_classify(dev, "vvc355_", bd, src, off, w, h, vb_pos)
                assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), f"bd={bd} {w}x{h} vb={vb_pos}"


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_alf_recon_coeff_and_clip(dev, orc, bd):
    rng = np.random.default_rng(0x5EED0004 + bd)
    for size in (1, 7, 256, 1024):
        cls = rng.integers(0, 25, size=size).astype(np.int32)
        tr = rng.integers(0, 4, size=size).astype(np.int32)
        coeff_set = rng.integers(-128, 128, size=(64, 12)).astype(np.int16)
        clip_idx = rng.integers(0, 4, size=(25, 12)).astype(np.uint8)
        c2f = rng.integers(0, 64, size=25).astype(np.uint8)
        out = []
        for lib, pre in ((orc, "orc_"), (dev, "vvc355_")):
            coeff = np.zeros((size, 12), np.int16)
            clip = np.zeros((size, 12), np.int16)
            getattr(lib, pre + "alf_recon_coeff_and_clip")(bd, P(coeff), P(clip), P(cls), P(tr), size, P(coeff_set), P(clip_idx), P(c2f))
            out.append((coeff, clip))
        assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])


@pytest.mark.parametrize("bd", [8, 10, 12])
@pytest.mark.parametrize("hs,vs", [(1, 1), (1, 0), (0, 0)])
def test_alf_filter_cc(dev, orc, bd, hs, vs):
    rng = np.random.default_rng(0x5EED0005 + bd + 16 * hs + 32 * vs)
    for (w, h) in [(4, 4), (16, 8), (64, 64), (60, 34)]:
        if (w << hs) > 128 or (h << vs) > 128:
            continue
        for vb_pos in ((h << vs) - 4, 1000, 2):
            luma, off = ac.make_src(rng, bd)
            coeff = rng.integers(-64, 64, size=7).astype(np.int16)
            dst0 = ac.rand_pixels(rng, (h + 4, w + 16), bd)
            res = []
            for lib, pre in ((orc, "orc_"), (dev, "vvc355_")):
                dst = dst0.copy()
                getattr(lib, pre + "alf_filter_cc")(bd, P(dst, 2 * dst.shape[1] + 8), dst.shape[1] * dst.itemsize,
                                                    P(luma, off), luma.shape[1] * luma.itemsize, w, h, hs, vs, P(coeff), vb_pos)
                res.append(dst)
            assert np.array_equal(res[0], res[1]), f"bd={bd} {w}x{h} hs={hs} vs={vs} vb={vb_pos}"
