"""GPU parity of the affine sub-block stage (affine-filter MC + PROF, uni / bi, weighted or not, edge emulation) vs the
oracle's restatement of luma_prof_uni / luma_prof_bi (vvc_inter.c:369-447)."""
import ctypes

import numpy as np
import pytest

from conftest import P, rand_pixels
from ffvvc_amd import abi, batch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_affine_frame(dev, orc, bd):
    orc.orc_affine_block.argtypes = [ctypes.c_int, ctypes.POINTER(abi.AffineJob)]
    orc.orc_affine_block.restype = None
    rng = np.random.default_rng(0x5EED0900 + bd)
    pw, ph = 128, 96
    isz = 1 if bd == 8 else 2
    refs = [rand_pixels(rng, (ph, pw), bd) for _ in range(2)]
    want = np.full((ph, pw), 0x17, refs[0].dtype)
    d_out = batch.DeviceBuffer.from_host(want)
    d_refs = [batch.DeviceBuffer.from_host(r) for r in refs]
    n = (pw // 4) * (ph // 4)
    dmv = rng.integers(-32, 33, size=(n, 2, 2, 16)).astype(np.int16)          # [job][list][x | y][16]
    d_dmv = batch.DeviceBuffer.from_host(dmv)
    lut = np.sort(np.random.default_rng(0x10C5 + bd).integers(0, 1 << bd, size=1 << bd)).astype(refs[0].dtype)          # LMCS forward map on a third of the sub-blocks
    d_lut = batch.DeviceBuffer.from_host(lut)
    arr = (abi.AffineJob * n)()
    kinds = set()
    for i in range(n):
        x, y = (i % (pw // 4)) * 4, (i // (pw // 4)) * 4
        j = abi.AffineJob()
        j.x, j.y, j.pic_w, j.pic_h = x, y, pw, ph
        j.pred_flag = int(rng.integers(1, 4))
        far = rng.random() < 0.1                                               # far outside the picture: all reads emulated
        for k in range(4):
            j.mv[k] = int(rng.integers(-3000, 3000)) if far else int(rng.integers(-300, 301))
        if rng.random() < 0.15:                                                # whole-sample motion in one or both directions
            j.mv[0] &= ~15
            j.mv[3] &= ~15
        j.prof0, j.prof1 = int(rng.integers(0, 2)), int(rng.integers(0, 2))
        j.weight_flag = int(rng.random() < 0.4)
        j.denom = int(rng.integers(0, 8))
        j.w0, j.w1, j.o0, j.o1 = (int(v) for v in rng.integers(-128, 128, size=4))
        j.dst_stride = j.ref0_stride = j.ref1_stride = pw * isz
        kinds.add((j.pred_flag, j.prof0, j.prof1, j.weight_flag))
        hj = abi.AffineJob.from_buffer_copy(j)
        hj.dst, hj.ref0, hj.ref1 = P(want, y * pw + x), P(refs[0]), P(refs[1])
        hj.diff_mv = P(dmv, i * 64)
        hj.lmcs_lut = P(lut) if i % 3 == 0 else 0
        orc.orc_affine_block(bd, ctypes.byref(hj))
        j.lmcs_lut = d_lut.ptr if i % 3 == 0 else 0
        j.dst, j.ref0, j.ref1 = d_out.ptr + (y * pw + x) * isz, d_refs[0].ptr, d_refs[1].ptr
        j.diff_mv = d_dmv.ptr + i * 128
        arr[i] = j
    d_jobs = batch.jobs_to_device(arr)
    dev.vvc355_affine_batch(None, bd, d_jobs.ptr, n)
    dev.vvc355_stream_sync(None)
    got = d_out.to_host(want.dtype, want.shape)
    bad = np.argwhere(got != want)
    assert len(bad) == 0, (f"{len(bad)} samples differ, first at {bad[0].tolist()}: job "
                           f"{[(jj.pred_flag, jj.prof0, jj.prof1, jj.weight_flag, list(jj.mv)) for jj in arr if jj.x <= bad[0][1] < jj.x + 4 and jj.y <= bad[0][0] < jj.y + 4]}")
    assert len(kinds) > 20
