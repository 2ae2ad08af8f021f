"""GPU parity of the batched, fused ALF-luma stage on whole (small) pictures: one launch over all CTBs, frame resident
in HBM, vs the oracle run the way the reference caller chains the slots per CTB (vvc_filter.c:1139-1186,1254-1318):
padded copy with edge replication -> classify -> recon_coeff_and_clip -> filter[LUMA]."""
import numpy as np
import pytest

import alf_cases as ac
from conftest import P
from ffvvc_amd import batch

pytestmark = pytest.mark.gpu


def oracle_alf_luma_frame(orc, bd, src, ctb, sets):
    h, w = src.shape
    dst = src.copy()
    padded = np.pad(src, 8, mode="edge")          # alf_prepare_buffer at picture edges == clamp-to-edge
    pw = padded.shape[1]
    for ry in range((h + ctb - 1) // ctb):
        for rx in range((w + ctb - 1) // ctb):
            x0, y0 = rx * ctb, ry * ctb
            cw, ch = min(ctb, w - x0), min(ctb, h - y0)
            coeff_set, clip_idx, c2f = sets[(rx + ry) % len(sets)]
            n = (cw // 4) * (ch // 4)
            off = (y0 + 8) * pw + x0 + 8
            cls, tr = ac.run_classify(orc, "orc_", bd, padded, off, cw, ch, ctb - 4)
            coeff = np.zeros((n, 12), np.int16)
            clip = np.zeros((n, 12), np.int16)
            orc.orc_alf_recon_coeff_and_clip(bd, P(coeff), P(clip), P(cls), P(tr), n, P(coeff_set), P(clip_idx), P(c2f))
            orc.orc_alf_filter_luma(bd, P(dst, y0 * w + x0), w * dst.itemsize, P(padded, off), pw * padded.itemsize,
                                    cw, ch, P(coeff), P(clip), ctb - 4)
    return dst


@pytest.mark.parametrize("bd", [8, 10, 12])
@pytest.mark.parametrize("dims", [(416, 240, 128), (128, 128, 128), (200, 136, 64), (64, 36, 32)])
def test_alf_luma_fused_frame(dev, orc, bd, dims):
    width, height, ctb = dims
    rng = np.random.default_rng(0x5EED0100 + bd + width)
    src = ac.rand_pixels(rng, (height, width), bd)
    # a smooth region so that directional classes occur, not only the high-activity ones
    sm = ac.make_src(rng, bd, smooth=True)[0]
    sh, sw = min(height // 2, sm.shape[0]), min(width // 2, sm.shape[1])
    src[:sh, :sw] = sm[:sh, :sw]
    sets = []
    for _ in range(3):
        sets.append((rng.integers(-128, 128, size=(64, 12)).astype(np.int16),
                     rng.integers(0, 4, size=(25, 12)).astype(np.uint8),
                     rng.integers(0, 64, size=25).astype(np.uint8)))
    want = oracle_alf_luma_frame(orc, bd, src, ctb, sets)

    pitched = batch.to_pitched(src)
    pitch = pitched.shape[1] * pitched.itemsize
    d_src = batch.DeviceBuffer.from_host(pitched)
    d_dst = batch.DeviceBuffer.from_host(np.full_like(pitched, 0x33))
    d_sets = [tuple(batch.DeviceBuffer.from_host(a) for a in s) for s in sets]
    ncx = (width + ctb - 1) // ctb

    def per_ctb(rx, ry):
        s = d_sets[(rx + ry) % len(d_sets)]
        return s[0].ptr, s[1].ptr, s[2].ptr

    jobs = batch.alf_luma_jobs(d_dst.ptr, d_src.ptr, pitch, pitched.itemsize, width, height, ctb, per_ctb)
    d_jobs = batch.jobs_to_device(jobs)
    dev.vvc355_alf_luma_batch(None, bd, 1, d_jobs.ptr, len(jobs))
    dev.vvc355_stream_sync(None)
    got = d_dst.to_host(pitched.dtype, pitched.shape)
    assert np.array_equal(got[:, :width], want)
    assert np.all(got[:, width:] == 0x33)      # nothing written outside the picture
    assert ncx >= 1


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_alf_chroma_and_cc_frame(dev, orc, bd):
    """ALF chroma then CC-ALF of a whole 4:2:0 picture, one launch each over all CTBs, vs the oracle per CTB on edge-replicated
    copies (vvc_filter.c:1188-1252: alf_filter_chroma, alf_filter_cc)."""
    from ffvvc_amd import abi
    rng = np.random.default_rng(0x5EED0180 + bd)
    cw_, ch_, ctb = 208, 120, 64                      # chroma plane of a 416x240 picture, 64x64 chroma CTBs
    luma = ac.rand_pixels(rng, (2 * ch_, 2 * cw_), bd)
    src = ac.rand_pixels(rng, (ch_, cw_), bd)
    isz = src.itemsize
    clipv = np.array([1 << bd, 1 << (bd - 3), 1 << (bd - 5), 1 << (bd - 7)], np.int16)
    filt = [(rng.integers(-64, 64, size=6).astype(np.int16), clipv[rng.integers(0, 4, size=6)].astype(np.int16),
             np.concatenate([rng.integers(-32, 32, size=7), [0]]).astype(np.int16)) for _ in range(5)]
    want = src.copy()
    pad_c = np.pad(src, 8, mode="edge")
    pad_l = np.pad(luma, 8, mode="edge")
    pcw, plw = pad_c.shape[1], pad_l.shape[1]
    ctbs = [(x0, y0, min(ctb, cw_ - x0), min(ctb, ch_ - y0)) for y0 in range(0, ch_, ctb) for x0 in range(0, cw_, ctb)]
    for i, (x0, y0, w, h) in enumerate(ctbs):
        f, c, cc = filt[i % len(filt)]
        orc.orc_alf_filter_chroma(bd, P(want, y0 * cw_ + x0), cw_ * isz, P(pad_c, (y0 + 8) * pcw + x0 + 8), pcw * isz, w, h, P(f), P(c), ctb - 2)
    for i, (x0, y0, w, h) in enumerate(ctbs):
        cc = filt[i % len(filt)][2]
        orc.orc_alf_filter_cc(bd, P(want, y0 * cw_ + x0), cw_ * isz, P(pad_l, (2 * y0 + 8) * plw + 2 * x0 + 8), plw * isz, w, h, 1, 1, P(cc), 2 * ctb - 4)

    p_src, p_luma = batch.to_pitched(src), batch.to_pitched(pad_l)      # CC-ALF reads the luma picture with its apron in place
    d_src, d_luma = batch.DeviceBuffer.from_host(p_src), batch.DeviceBuffer.from_host(p_luma)
    d_dst = batch.DeviceBuffer.from_host(np.full_like(p_src, 0x33))
    pitch, lpitch = p_src.shape[1] * isz, p_luma.shape[1] * isz
    d_f = [tuple(batch.DeviceBuffer.from_host(a) for a in t) for t in filt]
    cj, ccj = (abi.AlfJob * len(ctbs))(), (abi.AlfJob * len(ctbs))()
    for i, (x0, y0, w, h) in enumerate(ctbs):
        j = cj[i]
        j.dst, j.src = d_dst.ptr + y0 * pitch + x0 * isz, d_src.ptr + y0 * pitch + x0 * isz
        j.dst_stride = j.src_stride = pitch
        j.coeff, j.clip = d_f[i % len(filt)][0].ptr, d_f[i % len(filt)][1].ptr
        j.w, j.h, j.vb_pos = w, h, ctb - 2
        j.ext_l, j.ext_t, j.ext_r, j.ext_b = min(2, x0), min(2, y0), min(2, cw_ - x0 - w), min(2, ch_ - y0 - h)
        k = ccj[i]
        k.dst, k.dst_stride = j.dst, pitch
        k.src, k.src_stride = d_luma.ptr + (2 * y0 + 8) * lpitch + (2 * x0 + 8) * isz, lpitch
        k.coeff = d_f[i % len(filt)][2].ptr
        k.w, k.h, k.vb_pos, k.hs, k.vs = w, h, 2 * ctb - 4, 1, 1
        k.ext_l = k.ext_t = k.ext_r = k.ext_b = 3             # the luma picture carries its apron: read in place
    d_cj, d_ccj = batch.jobs_to_device(cj), batch.jobs_to_device(ccj)
    dev.vvc355_alf_chroma_batch(None, bd, d_cj.ptr, len(ctbs))
    dev.vvc355_alf_cc_batch(None, bd, d_ccj.ptr, len(ctbs))
    dev.vvc355_stream_sync(None)
    got = d_dst.to_host(p_src.dtype, p_src.shape)
    bad = np.argwhere(got[:, :cw_] != want)
    assert len(bad) == 0, f"{len(bad)} samples differ, first at {bad[0].tolist()}"
    assert np.all(got[:, cw_:] == 0x33)
