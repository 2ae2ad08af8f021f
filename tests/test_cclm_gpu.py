"""GPU parity: the flattened intra_cclm_pred and lmcs_scale_chroma slots through the C ABI vs the CPU oracle, bit-exact.
No reference unit test exists for these; inputs are random planes, every chroma format (4:2:0, 4:2:2, 4:4:4), the three
CCLM modes, both chroma sample locations, CTU-boundary and picture-edge positions, random availability."""
import ctypes

import numpy as np
import pytest

from conftest import P, rand_pixels
from ffvvc_amd import abi

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bd", [8, 10, 12])
@pytest.mark.parametrize("fmt", [(1, 1), (1, 0), (0, 0)])
def test_cclm_flat(dev, orc, bd, fmt):
    hs, vs = fmt
    rng = np.random.default_rng(0x5EED0600 + bd + 16 * hs + 32 * vs)
    orc.orc_intra_cclm_pred_flat.restype = None
    orc.orc_intra_cclm_pred_flat.argtypes = [ctypes.c_int, ctypes.c_void_p]
    pw, ph = 256, 192
    changed = 0
    for it in range(250):
        luma = rand_pixels(rng, (ph, pw), bd)
        if it % 3 == 0:       # smooth luma => small luma range => exercises the diff == 0 / small-k branches
            luma = np.clip((luma >> (bd - 2)).astype(np.int64) + int(rng.integers(0, 1 << bd) * 0.7), 0, (1 << bd) - 1).astype(luma.dtype)
        cb0 = rand_pixels(rng, (ph >> vs, pw >> hs), bd)
        cr0 = rand_pixels(rng, (ph >> vs, pw >> hs), bd)
        j = abi.CclmJob()
        j.hs, j.vs = hs, vs
        j.width = 1 << int(rng.integers(2 if hs else 1, 6))
        j.height = 1 << int(rng.integers(2 if vs else 1, 6))
        at_left, at_top = bool(rng.integers(0, 6) == 0), bool(rng.integers(0, 6) == 0)
        j.x0 = 0 if at_left else int(rng.integers(1, (pw - 2 * j.width - 8) // 8)) * 8
        j.y0 = 0 if at_top else int(rng.integers(1, (ph - 2 * j.height - 8) // 8)) * 8
        j.avail_l = 0 if at_left else int(rng.integers(0, 5) != 0)
        j.avail_t = 0 if at_top else int(rng.integers(0, 5) != 0)
        j.mode = int(rng.choice([81, 82, 83]))
        w, h = j.width >> hs, j.height >> vs
        j.top_avail_c = min(int(rng.choice([w, 2 * w, int(rng.integers(1, 2 * w + 1))])), (pw - j.x0) >> hs) if j.avail_t else 0
        j.left_avail_c = min(int(rng.choice([h, 2 * h, int(rng.integers(1, 2 * h + 1))])), (ph - j.y0) >> vs) if j.avail_l else 0
        j.collocated = int(rng.integers(0, 2))
        j.ctu_boundary = int(j.y0 % 128 == 0) if not at_top else 1
        j.luma_stride = pw * luma.itemsize
        j.cb_stride = j.cr_stride = (pw >> hs) * luma.itemsize
        res = []
        for which in ("orc", "dev"):
            cb, cr = cb0.copy(), cr0.copy()
            j.luma, j.cb, j.cr = luma.ctypes.data, cb.ctypes.data, cr.ctypes.data
            if which == "orc":
                orc.orc_intra_cclm_pred_flat(bd, ctypes.addressof(j))
            else:
                dev.vvc355_intra_cclm_pred_flat(bd, ctypes.addressof(j), pw, ph)
            res.append((cb, cr))
        assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]), (
            f"it={it} bd={bd} fmt={fmt} {j.width}x{j.height}@({j.x0},{j.y0}) mode={j.mode} avail=({j.avail_t},{j.avail_l}) "
            f"c=({j.top_avail_c},{j.left_avail_c}) col={j.collocated} ctu={j.ctu_boundary}")
        changed += int(not np.array_equal(res[0][0], cb0))
    assert changed > 200


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_lmcs_scale_chroma_flat(dev, orc, bd):
    rng = np.random.default_rng(0x5EED0610 + bd)
    orc.orc_lmcs_scale_chroma_flat.restype = None
    orc.orc_lmcs_scale_chroma_flat.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    pw, ph = 200, 136
    for it in range(60):
        luma = rand_pixels(rng, (ph, pw), bd)
        j = abi.LmcsScaleJob()
        j.luma_stride = pw * luma.itemsize
        j.pic_w, j.pic_h, j.size_y = pw, ph, int(rng.choice([32, 64]))
        j.x_vpdu = int(rng.integers(0, pw // j.size_y + 1)) * j.size_y
        j.y_vpdu = int(rng.integers(0, ph // j.size_y + 1)) * j.size_y
        j.x_vpdu = min(j.x_vpdu, (pw - 1) // j.size_y * j.size_y)
        j.y_vpdu = min(j.y_vpdu, (ph - 1) // j.size_y * j.size_y)
        j.avail_l = int(j.x_vpdu > 0 and rng.integers(0, 4) != 0)
        j.avail_t = int(j.y_vpdu > 0 and rng.integers(0, 4) != 0)
        j.min_bin_idx, j.max_bin_idx = int(rng.integers(0, 4)), int(rng.integers(10, 16))
        piv = np.sort(rng.integers(0, 1 << bd, size=17))
        for i in range(17):
            j.pivot[i] = int(piv[i])
        for i in range(16):
            j.chroma_scale_coeff[i] = int(rng.integers(512, 4096))
        w, h = int(rng.choice([2, 4, 8, 16, 32])), int(rng.choice([2, 4, 8, 16, 32]))
        coeff = rng.integers(-(1 << (bd + 2)), 1 << (bd + 2), size=w * h).astype(np.int32)
        out = []
        for which in ("orc", "dev"):
            dst = np.zeros(w * h, np.int32)
            j.luma = luma.ctypes.data
            fn = orc.orc_lmcs_scale_chroma_flat if which == "orc" else dev.vvc355_lmcs_scale_chroma_flat
            fn(bd, ctypes.addressof(j), P(dst), P(coeff), w, h)
            out.append(dst)
        assert np.array_equal(out[0], out[1]), f"it={it}"
        assert np.any(out[0] != 0)
