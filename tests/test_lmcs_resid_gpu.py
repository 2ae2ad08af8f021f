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


@pytest.mark.parametrize("bd,ctb_log2,pw,ph", [(10, 7, 456, 328), (8, 6, 328, 200), (12, 5, 200, 136), (10, 7, 64, 64)])
def test_lmcs_vpdu_scale_pass(dev, orc, bd, ctb_log2, pw, ph):
    """The scale of every 64x64 unit in one launch (vvc355_lmcs_vpdu_scale_pass) against the oracle's table, with tiles and slices cutting
    the neighbours off; then residual jobs that read the table (joint bit 4) against jobs that derive the scale themselves."""
    orc.orc_lmcs_vpdu_scale_pass.argtypes = [ctypes.c_int, ctypes.POINTER(abi.LmcsScaleFrame)]
    orc.orc_lmcs_vpdu_scale_pass.restype = None
    orc.orc_lmcs_chroma_resid_block.argtypes = [ctypes.c_int, ctypes.POINTER(abi.LmcsResidJob), ctypes.POINTER(abi.LmcsModel)]
    orc.orc_lmcs_chroma_resid_block.restype = None
    rng = np.random.default_rng(0x5EED0EC8 + bd + ctb_log2)
    isz = 1 if bd == 8 else 2
    ctb = 1 << ctb_log2
    size_y = min(ctb, 64)
    cw, chh = (pw + ctb - 1) // ctb, (ph + ctb - 1) // ctb
    ux, uy = (pw + size_y - 1) // size_y, (ph + size_y - 1) // size_y
    luma = rand_pixels(rng, (ph, pw), bd)
    # a smooth ramp under the noise so that units land in different bins of the model
    luma = np.clip(luma.astype(np.int64) // 4 + (np.add.outer(np.arange(ph), np.arange(pw)) * ((3 << bd) // 4) // (pw + ph)), 0, (1 << bd) - 1).astype(luma.dtype)
    model = recon_cases.ReconWork.lmcs_model(rng, bd)
    n = cw * chh
    cut = int(rng.integers(1, n)) if n > 1 else 1
    slice_idx = (np.arange(n) >= cut).astype(np.int16)
    col_bd = np.array([0 if x < 2 else 2 for x in range(cw)] + [cw], np.int16)
    row_bd = np.array([0 if y < 1 else 1 for y in range(chh)] + [chh], np.int16)
    want = np.full(ux * uy, -1, np.int16)
    hf = abi.LmcsScaleFrame()
    hf.luma, hf.scale, hf.model = P(luma), P(want), ctypes.addressof(model)
    hf.slice_idx, hf.ctb_to_col_bd, hf.ctb_to_row_bd = P(slice_idx), P(col_bd), P(row_bd)
    hf.luma_stride, hf.width, hf.height, hf.ctb_width, hf.ctb_log2, hf.size_y = pw * isz, pw, ph, cw, ctb_log2, size_y
    orc.orc_lmcs_vpdu_scale_pass(bd, ctypes.byref(hf))

    p_luma = batch.to_pitched(luma)
    d_luma = batch.DeviceBuffer.from_host(p_luma)
    d_model = batch.DeviceBuffer.from_host(np.frombuffer(bytes(model), np.uint8))
    d_tabs = [batch.DeviceBuffer.from_host(t) for t in (slice_idx, col_bd, row_bd)]
    d_scale = batch.DeviceBuffer.from_host(np.full(ux * uy + 8, -1, np.int16))
    df = abi.LmcsScaleFrame.from_buffer_copy(hf)
    df.luma, df.scale, df.model = d_luma.ptr, d_scale.ptr, d_model.ptr
    df.slice_idx, df.ctb_to_col_bd, df.ctb_to_row_bd = (d.ptr for d in d_tabs)
    df.luma_stride = p_luma.shape[1] * isz
    d_f = batch.DeviceBuffer.from_host(np.frombuffer(bytes(df), np.uint8))
    dev.vvc355_lmcs_vpdu_scale_pass(None, bd, d_f.ptr, ctypes.addressof(df))
    dev.vvc355_stream_sync(None)
    got = d_scale.to_host(np.int16, (ux * uy + 8,))
    assert np.array_equal(got[:ux * uy], want), f"units differ: {np.argwhere(got[:ux * uy] != want).ravel().tolist()[:8]}"
    assert np.all(got[ux * uy:] == -1)
    if ux * uy > 4:
        assert len(set(want.tolist())) > 2                       # the ramp reaches several bins

    # residual jobs reading the table: one chroma block per unit, 4:2:0
    cpw, cph = pw >> 1, ph >> 1
    chroma = rand_pixels(rng, (cph, cpw), bd)
    expect = chroma.copy()
    d_c = batch.DeviceBuffer.from_host(chroma)
    jobs, resid_len = [], 0
    for vy in range(uy):
        for vx in range(ux):
            x, y = vx * size_y, vy * size_y
            w, h = min(int(rng.choice([2, 4, 8, 16])), (pw - x) >> 1), min(int(rng.choice([2, 4, 8, 16])), (ph - y) >> 1)
            if w < 2 or h < 2:
                continue
            jobs.append((vx, vy, x >> 1, y >> 1, w, h, resid_len))
            resid_len += w * h
    resid = rng.integers(-(1 << bd), 1 << bd, size=max(resid_len, 1)).astype(np.int32)
    d_res = batch.DeviceBuffer.from_host(resid)
    arr = (abi.LmcsResidJob * len(jobs))()
    for i, (vx, vy, bx, by, w, h, off) in enumerate(jobs):
        j = abi.LmcsResidJob()
        j.w, j.h, j.x_vpdu, j.y_vpdu, j.pic_w, j.pic_h, j.size_y = w, h, vx * size_y, vy * size_y, pw, ph, size_y
        j.joint = 8 | 16 | int(rng.choice([0, 1, 1 | 2]))
        hj = abi.LmcsResidJob.from_buffer_copy(j)
        hj.dst, hj.dst_stride, hj.resid = P(expect, by * cpw + bx), cpw * isz, P(resid, off)
        hj.luma = P(want, vy * ux + vx)
        orc.orc_lmcs_chroma_resid_block(bd, ctypes.byref(hj), ctypes.byref(model))
        j.dst, j.dst_stride, j.resid = d_c.ptr + (by * cpw + bx) * isz, cpw * isz, d_res.ptr + off * 4
        j.luma = d_scale.ptr + (vy * ux + vx) * 2
        arr[i] = j
    d_jobs = batch.jobs_to_device(arr)
    dev.vvc355_lmcs_chroma_resid_batch(None, bd, d_jobs.ptr, len(jobs), d_model.ptr)
    dev.vvc355_stream_sync(None)
    got_c = d_c.to_host(chroma.dtype, chroma.shape)
    assert np.array_equal(got_c, expect) and np.any(expect != chroma)
