"""GPU parity of vvc355_itx_frame_build — the TU loop of itransform (vvc_intra.c:431-472) as a descriptor builder: transform-block jobs (and,
with chroma residual scaling, the scaled-residual jobs) written on the device from one 16-byte record per transform block.  The records of a
small picture go through the builder, the transform stage (vvc355_itx_shape_batch on the device-built jobs), vvc355_lmcs_vpdu_scale_pass and
vvc355_lmcs_chroma_resid_batch; the oracle walks the same records block by block: dequant + itx (+ add_residual), and for the blocks that
keep their residual the 64x64 unit's scale from the reconstructed luma, lmcs_scale_chroma and the add."""
import ctypes

import numpy as np
import pytest

import recon_cases
from conftest import P, rand_pixels
from ffvvc_amd import abi, batch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bd,scaled", [(10, True), (8, True), (12, False)])
def test_itx_frame_build(dev, orc, bd, scaled):
    orc.orc_lmcs_chroma_resid_block.argtypes = [ctypes.c_int, ctypes.POINTER(abi.LmcsResidJob), ctypes.POINTER(abi.LmcsModel)]
    orc.orc_lmcs_chroma_resid_block.restype = None
    rng = np.random.default_rng(0x5EED0B10 + bd)
    isz = 1 if bd == 8 else 2
    pw, ph = 256, 128                                    # 4 x 2 units of 64x64; chroma 4:2:0
    dims = [(pw, ph), (pw // 2, ph // 2), (pw // 2, ph // 2)]
    planes = [rand_pixels(rng, (d[1], d[0]), bd) for d in dims]
    want = [p.copy() for p in planes]
    model = recon_cases.ReconWork.lmcs_model(rng, bd)
    # transform blocks: luma 16x16 on every position; per 16x16 luma area one 8x8 block per chroma component
    recs, coeff_off = [], 0
    for y in range(0, ph, 16):
        for x in range(0, pw, 16):
            recs.append((0, x, y, 4, 4, coeff_off)); coeff_off += 256
            for c in (1, 2):
                recs.append((c, x // 2, y // 2, 3, 3, coeff_off)); coeff_off += 64
    n = len(recs)
    levels = (rng.integers(-64, 65, size=coeff_off) * (rng.random(coeff_off) < 0.3)).astype(np.int32)
    tus = np.zeros(n, np.dtype(abi.ItxTu, align=True))
    keep = np.zeros(n, bool)
    for i, (c, x, y, lw, lh, off) in enumerate(recs):
        t = tus[i]
        t["coeff_off"], t["x0"], t["y0"], t["log2_w"], t["log2_h"], t["c_idx"] = off, x, y, lw, lh, c
        t["nzw"], t["nzh"] = int(rng.integers(1, (1 << lw) + 1)), int(rng.integers(1, (1 << lh) + 1))
        t["qp"], t["tr"] = int(rng.integers(22, 40)), int(rng.integers(0, 3)) | int(rng.integers(0, 3)) << 4
        flags = 1 | (int(rng.integers(0, 2)) << 1)
        if scaled and c and rng.random() < 0.7:          # the chroma residual stays in the arena and is scaled + added afterwards
            lx, ly = (2 * x) & ~63, (2 * y) & ~63
            flags |= 4 | 64 | ((lx > 0) << 4) | ((ly > 0) << 5)
            keep[i] = True
        t["flags"] = flags
    # levels outside a block's nz window do not exist
    for i, (c, x, y, lw, lh, off) in enumerate(recs):
        blk = levels[off:off + (1 << (lw + lh))].reshape(1 << lh, 1 << lw)
        blk[int(tus[i]["nzh"]):, :] = 0
        blk[:, int(tus[i]["nzw"]):] = 0

    # ---- oracle: luma first (the scale reads reconstructed luma), then chroma
    res_of = {}
    for i, (c, x, y, lw, lh, off) in sorted(enumerate(recs), key=lambda e: e[1][0] != 0):
        t = tus[i]
        co = levels[off:off + (1 << (lw + lh))].copy()
        orc.orc_dequant(co.ctypes.data, lw, lh, 0, 0, int(t["nzw"]) - 1, int(t["nzh"]) - 1, int(t["qp"]), 0, (int(t["flags"]) >> 1) & 1, bd, 15, None, 1, -1)
        assert orc.orc_itx(int(t["tr"]) & 15, int(t["tr"]) >> 4, lw, lh, co.ctypes.data, int(t["nzw"]), int(t["nzh"]), 15, bd) == 0
        if keep[i]:
            res_of[i] = co
            continue
        w_ = dims[c][0]
        orc.orc_add_residual(bd, P(want[c], y * w_ + x), co.ctypes.data, 1 << lw, 1 << lh, w_ * isz)
    for i, co in res_of.items():
        c, x, y, lw, lh, off = recs[i]
        j = abi.LmcsResidJob()
        w_ = dims[c][0]
        j.dst, j.dst_stride, j.resid, j.luma, j.luma_stride = P(want[c], y * w_ + x), w_ * isz, co.ctypes.data, P(want[0]), pw * isz
        j.w, j.h, j.x_vpdu, j.y_vpdu, j.pic_w, j.pic_h, j.size_y = 1 << lw, 1 << lh, (2 * x) & ~63, (2 * y) & ~63, pw, ph, 64
        j.avail_l, j.avail_t, j.joint = int(j.x_vpdu > 0), int(j.y_vpdu > 0), 8
        orc.orc_lmcs_chroma_resid_block(bd, ctypes.byref(j), ctypes.byref(model))

    # ---- device
    pitched = [batch.to_pitched(p) for p in planes]
    d_planes = [batch.DeviceBuffer.from_host(p) for p in pitched]
    d_levels, d_tus = batch.DeviceBuffer.from_host(levels), batch.DeviceBuffer.from_host(tus.view(np.uint8))
    d_jobs = batch.DeviceBuffer.from_host(np.zeros(n * ctypes.sizeof(abi.ItxJob), np.uint8))
    d_rjobs = batch.DeviceBuffer.from_host(np.full(n * ctypes.sizeof(abi.LmcsResidJob), 0xA5, np.uint8))
    d_scale = batch.DeviceBuffer.from_host(np.zeros(4 * 2, np.int16))
    d_model = batch.DeviceBuffer.from_host(np.frombuffer(bytes(model), np.uint8))
    f = abi.ItxFrame()
    f.tus, f.jobs, f.coeffs, f.n_tus = d_tus.ptr, d_jobs.ptr, d_levels.ptr, n
    for c in range(3):
        f.plane[c], f.stride[c] = d_planes[c].ptr, pitched[c].shape[1] * isz
    f.range, f.bd, f.pixel_shift = 15, bd, int(isz == 2)
    f.width, f.height, f.hs, f.vs, f.size_y = pw, ph, 1, 1, 64
    f.resid_jobs, f.scale_table = (d_rjobs.ptr, d_scale.ptr) if scaled else (0, 0)
    d_f = batch.DeviceBuffer.from_host(np.frombuffer(bytes(f), np.uint8))
    dev.vvc355_itx_frame_build(None, d_f.ptr, ctypes.addressof(f))
    jsz = ctypes.sizeof(abi.ItxJob)
    # the builder keeps the records' order, and the transform stage wants one launch per shape: the caller bins its records by shape.  Built once
    # in parse order above (any order is valid), then again from the binned records, which is what the launches below use.
    order = np.argsort([r[3] for r in recs], kind="stable")
    tus = tus[order]; recs = [recs[i] for i in order]; keep = keep[order]
    dev.vvc355_upload(d_tus.ptr, np.ascontiguousarray(tus).view(np.uint8).ctypes.data, tus.nbytes)
    dev.vvc355_itx_frame_build(None, d_f.ptr, ctypes.addressof(f))
    n8 = sum(1 for r in recs if r[3] == 3)
    dev.vvc355_itx_shape_batch(None, bd, d_jobs.ptr, n8, 3, 3)
    dev.vvc355_itx_shape_batch(None, bd, d_jobs.ptr + n8 * jsz, n - n8, 4, 4)
    if scaled:
        slice_idx, col_bd, row_bd = np.zeros(2, np.int16), np.zeros(3, np.int16), np.zeros(2, np.int16)       # 2 x 1 CTBs of 128: one slice, one tile
        d_tabs = [batch.DeviceBuffer.from_host(t_) for t_ in (slice_idx, col_bd, row_bd)]
        sf = abi.LmcsScaleFrame()
        sf.luma, sf.scale, sf.model, sf.luma_stride = d_planes[0].ptr, d_scale.ptr, d_model.ptr, pitched[0].shape[1] * isz
        sf.slice_idx, sf.ctb_to_col_bd, sf.ctb_to_row_bd = (d.ptr for d in d_tabs)
        sf.width, sf.height, sf.ctb_width, sf.ctb_log2, sf.size_y = pw, ph, 2, 7, 64
        d_sf = batch.DeviceBuffer.from_host(np.frombuffer(bytes(sf), np.uint8))
        dev.vvc355_lmcs_vpdu_scale_pass(None, bd, d_sf.ptr, ctypes.addressof(sf))
        dev.vvc355_lmcs_chroma_resid_batch(None, bd, d_rjobs.ptr, n, d_model.ptr)
    dev.vvc355_stream_sync(None)
    for c in range(3):
        got = d_planes[c].to_host(pitched[c].dtype, pitched[c].shape)[:, :dims[c][0]]
        bad = np.argwhere(got != want[c])
        assert len(bad) == 0, f"component {c}: {len(bad)} samples differ, first at {bad[0].tolist()}"
        assert np.any(want[c] != planes[c])
    if scaled:
        assert keep.sum() > 20 and (~keep[[r[0] > 0 for r in recs]]).sum() > 5
