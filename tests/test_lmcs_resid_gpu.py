"""GPU parity of vvc355_lmcs_chroma_resid_batch — chroma residual scaling outside the in-order pass (the tail of itransform with
chroma_scale, vvc_intra.c:449-472, joint blocks :179-183; scale per 64x64 unit from lmcs_derive_chroma_scale,
vvc_intra_template.c:390-429) — against the oracle's orc_lmcs_chroma_resid_block on the same jobs."""
import ctypes

import numpy as np
import pytest

import recon_cases
from conftest import P, rand_pixels
from ffvvc_amd import abi, batch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bd,fmt,ctb_log2", [(10, (1, 1), 7), (8, (1, 1), 6), (12, (0, 0), 5), (10, (1, 0), 7)])
def test_lmcs_chroma_resid_batch(dev, orc, bd, fmt, ctb_log2):
    orc.orc_lmcs_chroma_resid_block.argtypes = [ctypes.c_int, ctypes.POINTER(abi.LmcsResidJob), ctypes.POINTER(abi.LmcsModel)]
    orc.orc_lmcs_chroma_resid_block.restype = None
    rng = np.random.default_rng(0x5EED0EC0 + bd + ctb_log2)
    hs, vs = fmt
    pw, ph = 328, 200                                  # neither a multiple of 64: units cut by the right and bottom picture edges
    isz = 1 if bd == 8 else 2
    size_y = min(1 << ctb_log2, 64)
    luma = rand_pixels(rng, (ph, pw), bd)
    cw, chh = pw >> hs, ph >> vs
    chroma = [rand_pixels(rng, (chh, cw), bd) for _ in range(2)]
    want = [p.copy() for p in chroma]
    model = recon_cases.ReconWork.lmcs_model(rng, bd)
    d_luma = batch.DeviceBuffer.from_host(luma)
    d_c = [batch.DeviceBuffer.from_host(p) for p in chroma]
    d_model = batch.DeviceBuffer.from_host(np.frombuffer(bytes(model), np.uint8))
    # disjoint chroma blocks on an 8x8-luma grid position, sizes 2..32
    blocks, resid_len = [], 0
    for y in range(0, ph - 63, 64):
        for x in range(0, pw - 63, 64):
            w, h = int(rng.choice([2, 4, 8, 16, 32])), int(rng.choice([2, 4, 8, 16, 32]))
            w, h = min(w, 64 >> hs), min(h, 64 >> vs)
            for c in range(2):
                blocks.append((c, x >> hs, y >> vs, w, h, x + int(rng.integers(0, 64)), y + int(rng.integers(0, 64)), resid_len))
                resid_len += w * h
    resid = rng.integers(-(1 << (bd + 1)), 1 << (bd + 1), size=resid_len).astype(np.int32)      # beyond the clip range of lmcs_scale_chroma now and then
    d_res = batch.DeviceBuffer.from_host(resid)
    n = len(blocks)
    arr = (abi.LmcsResidJob * n)()
    for i, (c, bx, by, w, h, cux, cuy, off) in enumerate(blocks):
        j = abi.LmcsResidJob()
        j.w, j.h = w, h
        j.x_vpdu, j.y_vpdu = cux & ~(size_y - 1), cuy & ~(size_y - 1)
        j.pic_w, j.pic_h, j.size_y = pw, ph, size_y
        j.avail_l = int(j.x_vpdu > 0 and rng.random() < 0.8)
        j.avail_t = int(j.y_vpdu > 0 and rng.random() < 0.8)
        j.joint = int(rng.choice([8, 8, 8 | 1, 8 | 1 | 2, 8 | 1 | 4, 8 | 1 | 2 | 4, 0, 1 | 2]))
        hj = abi.LmcsResidJob.from_buffer_copy(j)
        hj.dst, hj.dst_stride = P(want[c], by * cw + bx), cw * isz
        hj.resid, hj.luma, hj.luma_stride = P(resid, off), P(luma), pw * isz
        orc.orc_lmcs_chroma_resid_block(bd, ctypes.byref(hj), ctypes.byref(model))
        j.dst, j.dst_stride = d_c[c].ptr + (by * cw + bx) * isz, cw * isz
        j.resid, j.luma, j.luma_stride = d_res.ptr + off * 4, d_luma.ptr, pw * isz
        arr[i] = j
    d_jobs = batch.jobs_to_device(arr)
    dev.vvc355_lmcs_chroma_resid_batch(None, bd, d_jobs.ptr, n, d_model.ptr)
    dev.vvc355_stream_sync(None)
    for c in range(2):
        got = d_c[c].to_host(want[c].dtype, want[c].shape)
        bad = np.argwhere(got != want[c])
        assert len(bad) == 0, f"component {c + 1}: {len(bad)} samples differ, first at {bad[0].tolist()}"
        assert np.any(want[c] != chroma[c])
    assert len({int(a.joint) for a in arr}) >= 6 and {(int(a.avail_l), int(a.avail_t)) for a in arr} == {(0, 0), (0, 1), (1, 0), (1, 1)}
