"""GPU parity of the vectorised SAO frame stage (vvc355_sao_ctb_batch: band, or edge + restore fused, one launch over all
CTBs of a plane) vs the oracle chained the way the reference caller does per CTB (vvc_filter.c:154-300): padded CTB copy ->
edge_filter -> edge_restore[variant], or band_filter."""
import numpy as np
import pytest

from conftest import P, rand_pixels
from ffvvc_amd import abi, batch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bd", [8, 10, 12])
@pytest.mark.parametrize("dims", [(416, 240, 128), (200, 136, 64), (128, 128, 128)])
def test_sao_ctb_batch(dev, orc, bd, dims):
    width, height, ctb = dims
    rng = np.random.default_rng(0x5EED0800 + bd + width)
    src = rand_pixels(rng, (height, width), bd)
    if bd > 8:
        src = (src >> 2 << 2).astype(src.dtype)          # plateaus: equal neighbours occur
    isz = src.itemsize
    want = src.copy()
    pitched = batch.to_pitched(src)
    pitch = pitched.shape[1] * isz
    d_src = batch.DeviceBuffer.from_host(pitched)
    d_dst = batch.DeviceBuffer.from_host(np.full_like(pitched, 0x33))
    ncx, ncy = (width + ctb - 1) // ctb, (height + ctb - 1) // ctb
    jobs = (abi.SaoJob * (ncx * ncy))()
    ss = 320 // isz                                       # implicit edge source stride, in pixels
    for ry in range(ncy):
        for rx in range(ncx):
            x0, y0 = rx * ctb, ry * ctb
            w, h = min(ctb, width - x0), min(ctb, height - y0)
            j = jobs[ry * ncx + rx]
            offs = np.concatenate([[0], rng.integers(-(1 << (bd - 5)) + 1, 1 << (bd - 5), size=4)]).astype(np.int16)
            typ = int(rng.choice([1, 3, 3]))
            eo, band = int(rng.integers(0, 4)), int(rng.integers(0, 32))
            borders = np.array([x0 == 0, y0 == 0, x0 + w == width, y0 + h == height], np.int32)
            restore = int(rng.integers(0, 2))
            ve = (rng.integers(0, 2, size=2) * (1 - borders[[0, 2]])).astype(np.uint8)       # no restore flag on a picture border
            he = (rng.integers(0, 2, size=2) * (1 - borders[[1, 3]])).astype(np.uint8)
            de = rng.integers(0, 2, size=4).astype(np.uint8)
            de[0] *= not (borders[0] or borders[1]); de[1] *= not (borders[1] or borders[2])
            de[2] *= not (borders[2] or borders[3]); de[3] *= not (borders[0] or borders[3])
            # ---- oracle, per CTB like ff_vvc_sao_filter
            dstv = want[y0:y0 + h, x0:x0 + w]
            tmp = np.zeros((h, w), src.dtype)
            if typ == 1:
                blk = np.ascontiguousarray(src[y0:y0 + h, x0:x0 + w])
                orc.orc_sao_band_filter(bd, P(tmp), P(blk), w * isz, w * isz, P(offs), band, w, h)
            else:
                padded = np.zeros((h + 2, ss), src.dtype)
                ys, xs = slice(max(y0 - 1, 0), min(y0 + h + 1, height)), slice(max(x0 - 1, 0), min(x0 + w + 1, width))
                padded[ys.start - (y0 - 1):ys.stop - (y0 - 1), 8 + xs.start - x0:8 + xs.stop - x0] = src[ys, xs]
                orc.orc_sao_edge_filter(bd, P(tmp), P(padded, ss + 8), w * isz, P(offs), eo, w, h)
                orc.orc_sao_edge_restore(bd, restore, P(tmp), P(padded, ss + 8), w * isz, ss * isz, P(offs), eo, P(borders), w, h,
                                         P(ve), P(he), P(de))
            dstv[:, :] = tmp
            # ---- device job
            j.dst, j.src = d_dst.ptr + y0 * pitch + x0 * isz, d_src.ptr + y0 * pitch + x0 * isz
            j.dst_stride = j.src_stride = pitch
            j.w, j.h, j.type, j.eo, j.band_position, j.restore = w, h, typ, eo, band, restore
            for k in range(5):
                j.offset_val[k] = int(offs[k])
            for k in range(4):
                j.borders[k], j.diag_edge[k] = int(borders[k]), int(de[k]) if restore else 0
            for k in range(2):
                j.vert_edge[k], j.horiz_edge[k] = (int(ve[k]), int(he[k])) if restore else (0, 0)
    d_jobs = batch.jobs_to_device(jobs)
    dev.vvc355_sao_ctb_batch(None, bd, d_jobs.ptr, len(jobs), ctb)
    dev.vvc355_stream_sync(None)
    got = d_dst.to_host(pitched.dtype, pitched.shape)[:, :width]
    bad = np.argwhere(got != want)
    assert len(bad) == 0, f"{len(bad)} samples differ, first at {bad[0].tolist()}"
