"""GPU parity: intra leaf predictors and the flattened intra_pred slot through the C ABI vs the CPU oracle, bit-exact.
The reference has no checkasm test for intra; inputs are uniform random reference samples, every block size 4..64 (2..32
chroma), every mode reachable through the wide-angle mapping for that aspect ratio, MRL / ISP / BDPCM / MIP variants and
random neighbour availability."""
import ctypes

import numpy as np
import pytest

from conftest import P, px_dtype, rand_pixels
from ffvvc_amd import abi

pytestmark = pytest.mark.gpu
ORG = 512


def wide_angle(mode, w, h):
    """ff_vvc_wide_angle_mode_mapping (vvc_intra.c:693) for a non-ISP block."""
    ratio = abs(int(np.log2(w)) - int(np.log2(h)))
    mx = 8 + 2 * ratio if ratio > 1 else 8
    mn = 60 - 2 * ratio if ratio > 1 else 60
    if w > h and 2 <= mode < mx:
        return mode + 65
    if h > w and mn < mode <= 66:
        return mode - 67
    return mode


def edges(rng, bd):
    return rand_pixels(rng, (1024,), bd), rand_pixels(rng, (1024,), bd)


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_leaf_predictors(dev, orc, bd):
    rng = np.random.default_rng(0x5EED0500 + bd)
    sizes = [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64), (4, 16), (16, 4), (8, 32), (32, 8), (64, 16), (4, 64), (64, 4), (2, 8), (8, 2), (2, 2)]
    for (w, h) in sizes:
        top, left = edges(rng, bd)
        ps = top.itemsize
        stride = 80

        def call(name, args):
            out = []
            for lib, pre in ((orc, "orc_"), (dev, "vvc355_")):
                d = np.full((h + 2, stride), 0x55, top.dtype)
                getattr(lib, pre + name)(bd, P(d, stride + 4), *args(d), stride)
                out.append(d)
            assert np.array_equal(out[0], out[1]), f"{name} {w}x{h} bd={bd}"

        call("pred_planar", lambda d: (P(top, ORG), P(left, ORG), w, h))
        call("pred_dc", lambda d: (P(top, ORG), P(left, ORG), w, h))
        call("pred_v", lambda d: (P(top, ORG), w, h))
        call("pred_h", lambda d: (P(left, ORG), w, h))
        if w >= 4 and h >= 4:
            size_id = 0 if (w == 4 and h == 4) else 1 if (w == 4 or h == 4 or (w == 8 and h == 8)) else 2
            for mip_mode in range((16, 8, 6)[size_id]):
                for tr in (0, 1):
                    out = []
                    for lib, pre in ((orc, "orc_"), (dev, "vvc355_")):
                        d = np.full((h + 2, stride), 0x55, top.dtype)
                        getattr(lib, pre + "pred_mip")(bd, P(d, stride + 4), P(top, ORG), P(left, ORG), w, h, stride, mip_mode, tr)
                        out.append(d)
                    assert np.array_equal(out[0], out[1]), f"mip {w}x{h} mode={mip_mode} tr={tr}"
        modes = sorted({wide_angle(m, w, h) for m in range(2, 67)} - {18, 50})
        for mode in modes:
            for c_idx in (0, 1):
                for ref_idx in ((0, 1, 2) if not c_idx else (0,)):
                    for filter_flag in (0, 1):
                        pdpc_ok = orc.orc_intra_need_pdpc(w, h, 0, mode, ref_idx)
                        for need_pdpc in ({0, pdpc_ok}):
                            name = "pred_angular_v" if mode >= 34 else "pred_angular_h"
                            out = []
                            for lib, pre in ((orc, "orc_"), (dev, "vvc355_")):
                                d = np.full((h + 2, stride), 0x55, top.dtype)
                                getattr(lib, pre + name)(bd, P(d, stride + 4), P(top, ORG), P(left, ORG), w, h, stride,
                                                         c_idx, mode, ref_idx, filter_flag, need_pdpc)
                                out.append(d)
                            assert np.array_equal(out[0], out[1]), f"{name} {w}x{h} mode={mode} c={c_idx} ref={ref_idx} f={filter_flag} pdpc={need_pdpc}"


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_intra_pred_flat(dev, orc, bd):
    rng = np.random.default_rng(0x5EED0510 + bd)
    orc.orc_intra_pred_flat.restype = None
    orc.orc_intra_pred_flat.argtypes = [ctypes.c_int, ctypes.c_void_p]
    pw, ph = 256, 192
    n_cases = 0
    for it in range(600):
        plane = rand_pixels(rng, (ph, pw), bd)
        c_idx = int(rng.integers(0, 3))
        lw, lh = int(rng.integers(1 if c_idx else 2, 7)), int(rng.integers(1 if c_idx else 2, 7))
        if c_idx:
            lw, lh = min(lw, 5), min(lh, 5)
        w, h = 1 << lw, 1 << lh
        j = abi.IntraJob()
        j.plane = plane.ctypes.data
        j.stride = pw * plane.itemsize
        j.plane_w, j.plane_h = pw, ph
        j.w, j.h, j.c_idx = w, h, c_idx
        at_left, at_top = bool(rng.integers(0, 6) == 0), bool(rng.integers(0, 6) == 0)
        j.x = 0 if at_left else int(rng.integers(1, (pw - 2 * w - 8) // 4)) * 4
        j.y = 0 if at_top else int(rng.integers(1, (ph - 2 * h - 8) // 4)) * 4
        kind = int(rng.integers(0, 8))
        base_mode = (0, 1, 18, 50)[kind] if kind < 4 else int(rng.integers(2, 67))
        j.isp_split = int(not c_idx and rng.integers(0, 5) == 0)
        if j.isp_split:
            j.cb_width, j.cb_height = min(64, w * int(rng.choice([1, 2, 4]))), min(64, h * int(rng.choice([1, 2, 4])))
            mode = base_mode            # the mapping then uses the coding-block shape; keep the unmapped mode (always valid)
            if mode not in (0, 1, 18, 50):
                mode = wide_angle(base_mode, j.cb_width, j.cb_height)
        else:
            j.cb_width, j.cb_height = w, h
            mode = base_mode if base_mode in (0, 1, 18, 50) else wide_angle(base_mode, w, h)
        j.mode = mode
        j.ref_idx = 0 if (c_idx or mode == 0 or rng.integers(0, 3)) else int(rng.choice([1, 2, 3]))
        j.bdpcm_flag = int(rng.integers(0, 6) == 0)
        j.is_mip = int(not c_idx and not j.isp_split and not j.ref_idx and w >= 4 and h >= 4 and w <= 64 and h <= 64 and rng.integers(0, 5) == 0)
        if j.is_mip:
            size_id = 0 if (w == 4 and h == 4) else 1 if (w == 4 or h == 4 or (w == 8 and h == 8)) else 2
            j.mip_mode = int(rng.integers(0, (16, 8, 6)[size_id]))
            j.mip_transposed = int(rng.integers(0, 2))
        reach_w = (j.cb_width + w) if j.isp_split else 2 * w
        reach_h = (j.cb_height + h) if j.isp_split else 2 * h
        j.left_avail = 0 if at_left else int(rng.choice([0, h, reach_h, int(rng.integers(0, reach_h + 1))]))
        j.top_avail = 0 if at_top else int(rng.choice([0, w, reach_w, int(rng.integers(0, reach_w + 1))]))
        j.left_avail = min(j.left_avail, ph - j.y)
        j.top_avail = min(j.top_avail, pw - j.x)
        j.cand_up_left = int(not at_left and not at_top and rng.integers(0, 4) != 0)
        want, got = plane.copy(), plane.copy()
        j.plane = want.ctypes.data
        orc.orc_intra_pred_flat(bd, ctypes.addressof(j))
        j.plane = got.ctypes.data
        dev.vvc355_intra_pred_flat(bd, ctypes.addressof(j))
        assert np.array_equal(want, got), (f"it={it} bd={bd} c={c_idx} {w}x{h}@({j.x},{j.y}) mode={mode} ref={j.ref_idx} mip={j.is_mip} "
                                           f"isp={j.isp_split} avail=({j.left_avail},{j.top_avail},{j.cand_up_left})")
        n_cases += int(not np.array_equal(want, plane))
    assert n_cases > 500


@pytest.mark.parametrize("bd", [8, 10])
def test_intra_pred_batch_many_jobs(dev, orc, bd):
    """Many independent blocks in one launch, once per lanes-per-block mapping of the batched entry (half a wave, a wave, a
    workgroup per block): the blocks sit on a grid with untouched gaps, so each reads only original samples around it."""
    from ffvvc_amd import batch
    rng = np.random.default_rng(0x5EED0520 + bd)
    orc.orc_intra_pred_flat.restype = None
    orc.orc_intra_pred_flat.argtypes = [ctypes.c_int, ctypes.c_void_p]
    for (size, lg2_area) in [(8, 6), (16, 8), (32, 10)]:
        cell = 4 * size
        pw, ph = 8 * cell, 4 * cell
        plane = rand_pixels(rng, (ph, pw), bd)
        want = plane.copy()
        pitched = batch.to_pitched(plane)
        pitch = pitched.shape[1] * plane.itemsize
        d_plane = batch.DeviceBuffer.from_host(pitched)
        n = (pw // cell) * (ph // cell)
        arr = (abi.IntraJob * n)()
        for i in range(n):
            j = abi.IntraJob()
            j.x, j.y = (i % (pw // cell)) * cell + size, (i // (pw // cell)) * cell + size
            j.w = j.h = j.cb_width = j.cb_height = size
            j.c_idx = int(rng.integers(0, 3))
            j.mode = int(rng.choice([0, 1, 18, 50] + list(range(2, 67))))
            j.plane_w, j.plane_h = pw, ph
            j.left_avail, j.top_avail, j.cand_up_left = 2 * size, 2 * size, 1
            hj = abi.IntraJob.from_buffer_copy(j)
            hj.plane, hj.stride = want.ctypes.data, pw * plane.itemsize
            orc.orc_intra_pred_flat(bd, ctypes.addressof(hj))
            j.plane, j.stride = d_plane.ptr, pitch
            arr[i] = j
        d_jobs = batch.jobs_to_device(arr)
        dev.vvc355_intra_pred_batch(None, bd, d_jobs.ptr, n, lg2_area)
        dev.vvc355_stream_sync(None)
        got = d_plane.to_host(pitched.dtype, pitched.shape)[:, :pw]
        bad = np.argwhere(got != want)
        assert len(bad) == 0, f"{size}x{size} bd={bd}: {len(bad)} samples differ, first at {bad[0].tolist()}"
