"""GPU parity: inter-prediction slots through the C ABI vs the CPU oracle, bit-exact."""
import numpy as np
import pytest

import inter_cases as ic
from conftest import P, rand_pixels

pytestmark = pytest.mark.gpu
LIBS = (("orc", "orc_"), ("dev", "vvc355_"))


def both(orc, dev, fn):
    """Run fn(lib, prefix) on the oracle and the device library; return both results."""
    return fn(orc, "orc_"), fn(dev, "vvc355_")


def eq(a, b):
    if isinstance(a, tuple):
        for i, (x, y) in enumerate(zip(a, b)):
            if not np.array_equal(x, y):
                bad = np.argwhere(x != y)
                print(f"output #{i} differs at {bad[:8].tolist()} ({len(bad)} elements): want {x[tuple(bad[0])]} got {y[tuple(bad[0])]}")
                return False
        return True
    return np.array_equal(a, b)


@pytest.mark.parametrize("bd", [8, 10, 12])
@pytest.mark.parametrize("chroma", [0, 1])
def test_put_family(dev, orc, bd, chroma):
    rng = np.random.default_rng(0x5EED0200 + bd + chroma)
    luma_f, chroma_f = ic.tables(dev, "vvc355_")
    widths = [2, 4, 8, 16, 32, 64, 128] if chroma else [4, 8, 16, 32, 64, 128]
    for w in widths:
        for h in sorted({2 if chroma else 4, w, min(128, 2 * w), 12 if w >= 4 else 4, 128 if w == 128 else 36}):
            for vfrac in (0, 1):
                for hfrac in (0, 1):
                    plane, off = ic.src_plane(rng, bd)
                    ps = plane.itemsize
                    fset = int(rng.integers(0, 3))
                    nph = 32 if chroma else 16
                    mx, my = int(rng.integers(1, nph)), int(rng.integers(1, nph))
                    tab = chroma_f if chroma else luma_f
                    hf = np.ascontiguousarray(tab[fset, mx]); vf = np.ascontiguousarray(tab[fset, my])
                    denom, wx, ox = int(rng.integers(0, 8)), int(rng.integers(-128, 128)), int(rng.integers(-128, 128))

                    def run(lib, pre):
                        d16 = np.full((h + 2, ic.PB), 0x1234, np.int16)
                        getattr(lib, pre + "put")(bd, chroma, vfrac, hfrac, P(d16, ic.PB), P(plane, off), plane.shape[1] * ps, h, P(hf), P(vf), w)
                        du = np.full((h + 2, w + 8), 0x55, plane.dtype)
                        getattr(lib, pre + "put_uni")(bd, chroma, vfrac, hfrac, P(du, du.shape[1] + 4), du.shape[1] * ps, P(plane, off),
                                                      plane.shape[1] * ps, h, P(hf), P(vf), w)
                        dw = np.full((h + 2, w + 8), 0x55, plane.dtype)
                        getattr(lib, pre + "put_uni_w")(bd, chroma, vfrac, hfrac, P(dw, dw.shape[1] + 4), dw.shape[1] * ps, P(plane, off),
                                                        plane.shape[1] * ps, h, denom, wx, ox, P(hf), P(vf), w)
                        return d16, du, dw

                    a, b = both(orc, dev, run)
                    assert eq(a, b), f"bd={bd} chroma={chroma} {w}x{h} frac=({vfrac},{hfrac})"


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_avg_w_avg_ciip_gpm(dev, orc, bd):
    rng = np.random.default_rng(0x5EED0210 + bd)
    for (w, h) in [(2, 2), (4, 4), (8, 16), (16, 4), (32, 32), (64, 128), (128, 128), (128, 2)]:
        s0, s1 = ic.signed_i16_plane(rng), ic.signed_i16_plane(rng)
        denom = int(rng.integers(0, 8)); w0, w1, o0, o1 = (int(v) for v in rng.integers(-128, 128, size=4))
        inter = rand_pixels(rng, (h, w + 8), bd)
        intra = rand_pixels(rng, (h + 2, w + 8), bd)
        iw = int(rng.integers(1, 4))
        wmask = rng.integers(0, 9, size=(112 * 112)).astype(np.uint8)
        sx = int(rng.choice([1, -1, 2])); sy = int(rng.choice([112, -112, 224, 1]))
        woff = 112 * 56 + 56
        if abs(sx) * (w - 1) + abs(sy) * (h - 1) >= woff:
            sx, sy = 1, 1

        def run(lib, pre):
            ps = inter.itemsize
            d0 = np.full((h + 2, w + 8), 0x55, inter.dtype)
            getattr(lib, pre + "avg")(bd, P(d0, d0.shape[1] + 4), d0.shape[1] * ps, P(s0), P(s1), w, h)
            d1 = np.full((h + 2, w + 8), 0x55, inter.dtype)
            getattr(lib, pre + "w_avg")(bd, P(d1, d1.shape[1] + 4), d1.shape[1] * ps, P(s0), P(s1), w, h, denom, w0, w1, o0, o1)
            d2 = intra.copy()
            getattr(lib, pre + "put_ciip")(bd, P(d2, d2.shape[1] + 4), d2.shape[1] * ps, w, h, P(inter), inter.shape[1] * ps, iw)
            d3 = np.full((h + 2, w + 8), 0x55, inter.dtype)
            getattr(lib, pre + "put_gpm")(bd, P(d3, d3.shape[1] + 4), d3.shape[1] * ps, w, h, P(s0), P(s1), P(wmask, woff), sx, sy)
            return d0, d1, d2, d3

        a, b = both(orc, dev, run)
        assert eq(a, b), f"bd={bd} {w}x{h}"


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_bdof_prof_fetch(dev, orc, bd):
    rng = np.random.default_rng(0x5EED0220 + bd)
    for (w, h) in [(8, 8), (16, 16), (16, 8), (8, 16), (4, 4), (12, 16)]:
        plane, off = ic.src_plane(rng, bd)
        ps = plane.itemsize
        base0, base1 = ic.signed_i16_plane(rng, 32), ic.signed_i16_plane(rng, 32)
        xf, yf = int(rng.integers(0, 16)), int(rng.integers(0, 16))
        org = 4 * ic.PB + 8

        def run(lib, pre):
            s0, s1 = base0.copy(), base1.copy()
            for s in (s0, s1):
                getattr(lib, pre + "bdof_fetch_samples")(bd, P(s, org), P(plane, off), plane.shape[1] * ps, xf, yf, w, h)
            d = np.full((h + 2, w + 8), 0x55, plane.dtype)
            getattr(lib, pre + "apply_bdof")(bd, P(d, d.shape[1] + 4), d.shape[1] * ps, P(s0, org), P(s1, org), w, h)
            return s0, s1, d

        a, b = both(orc, dev, run)
        assert eq(a, b), f"bdof bd={bd} {w}x{h}"

    for _ in range(8):
        plane, off = ic.src_plane(rng, bd)
        ps = plane.itemsize
        base = ic.signed_i16_plane(rng, 32)
        dmx = rng.integers(-32, 33, size=16).astype(np.int16); dmy = rng.integers(-32, 33, size=16).astype(np.int16)
        denom, wx, ox = int(rng.integers(0, 8)), int(rng.integers(-128, 128)), int(rng.integers(-128, 128))
        org = 4 * ic.PB + 8
        pad = int(rng.integers(0, 2)); gw, gh_ = int(rng.choice([4, 8, 16])), int(rng.choice([4, 8, 16]))

        def run(lib, pre):
            s = base.copy()
            getattr(lib, pre + "fetch_samples")(bd, P(s, org), P(plane, off), plane.shape[1] * ps, int(dmx[0]) & 15, int(dmy[0]) & 15)
            d16 = np.full((8, ic.PB), 0x1234, np.int16)
            getattr(lib, pre + "apply_prof")(bd, P(d16, ic.PB), P(s, org), P(dmx), P(dmy))
            du = np.full((6, 12), 0x55, plane.dtype)
            getattr(lib, pre + "apply_prof_uni")(bd, P(du, 12 + 4), 12 * ps, P(s, org), P(dmx), P(dmy))
            dw = np.full((6, 12), 0x55, plane.dtype)
            getattr(lib, pre + "apply_prof_uni_w")(bd, P(dw, 12 + 4), 12 * ps, P(s, org), P(dmx), P(dmy), denom, wx, ox)
            g0 = np.full((20, 24), 0x1111, np.int16); g1 = np.full((20, 24), 0x2222, np.int16)
            getattr(lib, pre + "prof_grad_filter")(bd, P(g0), P(g1), 24, P(base, org), ic.PB, gw, gh_, pad)
            return s, d16, du, dw, g0, g1

        a, b = both(orc, dev, run)
        assert eq(a, b), f"prof bd={bd}"


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_dmvr_sad(dev, orc, bd):
    rng = np.random.default_rng(0x5EED0230 + bd)
    for (w, h) in [(12, 12), (20, 20), (20, 12), (12, 20), (8, 8), (128, 128)]:
        for vfrac in (0, 1):
            for hfrac in (0, 1):
                plane, off = ic.src_plane(rng, bd)
                ps = plane.itemsize
                mx, my = int(rng.integers(1, 16)), int(rng.integers(1, 16))

                def run(lib, pre):
                    d = np.full((h + 2, ic.PB), 0x1234, np.int16)
                    getattr(lib, pre + "dmvr")(bd, vfrac, hfrac, P(d, ic.PB), P(plane, off), plane.shape[1] * ps, h, mx, my, w)
                    return d

                a, b = both(orc, dev, run)
                assert eq(a, b), f"dmvr bd={bd} {w}x{h} ({vfrac},{hfrac})"
    for (w, h) in [(8, 8), (16, 16), (16, 8), (8, 16)]:
        s0 = rng.integers(0, 1 << 10, size=(h + 4, ic.PB)).astype(np.int16)
        s1 = rng.integers(0, 1 << 10, size=(h + 4, ic.PB)).astype(np.int16)
        for dx in range(5):
            for dy in range(5):
                want = orc.orc_sad(P(s0), P(s1), dx, dy, w, h)
                got = dev.vvc355_sad(P(s0), P(s1), dx, dy, w, h)
                assert want == got, f"sad {w}x{h} ({dx},{dy})"
