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


@pytest.mark.parametrize("bd,fmt", [(8, (1, 1)), (10, (1, 1)), (12, (1, 1)), (10, (1, 0)), (10, (0, 0))])
@pytest.mark.parametrize("mode", ["across", "slices", "tiles", "both"])
def test_sao_frame_pass(dev, orc, bd, fmt, mode):
    """The SAO stage driver (vvc355_sao_frame_pass: per-CTB flags and parameters derived on the device from the decoder's
    tables) vs the oracle's restatement of ff_vvc_sao_filter on the same tables; 4:2:0, partial CTBs at the right / bottom."""
    import ctypes
    orc.orc_sao_frame_pass.argtypes = [ctypes.c_int, ctypes.POINTER(abi.SaoFrame)]
    orc.orc_sao_frame_pass.restype = None
    hs, vs = fmt                                          # 4:2:0, 4:2:2, 4:4:4
    rng = np.random.default_rng(0x5EED0850 + bd + len(mode) + 16 * hs + 32 * vs)
    w, h, ctb_log2 = 328, 200, 6
    ctb = 1 << ctb_log2
    cw, ch = (w + ctb - 1) // ctb, (h + ctb - 1) // ctb
    dims = [(w, h), (w >> hs, h >> vs), (w >> hs, h >> vs)]
    isz = 1 if bd == 8 else 2
    src = [rand_pixels(rng, (d[1], d[0]), bd) for d in dims]
    if bd > 8:
        src = [(p >> 2 << 2).astype(p.dtype) for p in src]
    want = [np.full_like(p, 0x21) for p in src]
    p_src = [batch.to_pitched(p) for p in src]
    d_src = [batch.DeviceBuffer.from_host(p) for p in p_src]
    d_dst = [batch.DeviceBuffer.from_host(np.full_like(p, 0x21)) for p in p_src]
    tab = (abi.SaoCtb * (cw * ch))()
    for t in tab:
        for c in range(3):
            t.type_idx[c], t.band_position[c], t.eo_class[c] = int(rng.integers(0, 3)), int(rng.integers(0, 32)), int(rng.integers(0, 4))
            for k in range(1, 5):
                t.offset_val[c][k] = int(rng.integers(-(1 << (bd - 5)) + 1, 1 << (bd - 5)))
    # slices: horizontal bands of CTB rows cut at a random CTB; tiles: two columns x two rows
    cut = int(rng.integers(1, cw * ch))
    slice_idx = (np.arange(cw * ch) >= cut).astype(np.int16) + (np.arange(cw * ch) >= min(cw * ch - 1, cut + cw + 1)).astype(np.int16)
    col_bd = np.array([0 if x < 3 else 3 for x in range(cw)] + [cw], np.int16)
    row_bd = np.array([0 if y < 2 else 2 for y in range(ch)] + [ch], np.int16)
    tabs_host = [np.frombuffer(bytes(tab), np.uint8).copy(), slice_idx, col_bd, row_bd]
    tabs_dev = [batch.DeviceBuffer.from_host(t) for t in tabs_host]

    def fill(f, dst_ptrs, src_ptrs, dstrides, sstrides, tp):
        for c in range(3):
            f.dst[c], f.src[c], f.dst_stride[c], f.src_stride[c] = dst_ptrs[c], src_ptrs[c], dstrides[c], sstrides[c]
        f.sao, f.slice_idx, f.ctb_to_col_bd, f.ctb_to_row_bd = tp
        f.width, f.height, f.ctb_width, f.ctb_height = w, h, cw, ch
        f.ctb_log2, f.hs, f.vs, f.n_comp = ctb_log2, hs, vs, 3
        f.lfase = int(mode in ("across", "tiles"))
        f.no_tile_filter = int(mode in ("tiles", "both"))

    hf = abi.SaoFrame()
    fill(hf, [P(p) for p in want], [P(p) for p in src], [d[0] * isz for d in dims], [d[0] * isz for d in dims], [P(t) for t in tabs_host])
    orc.orc_sao_frame_pass(bd, ctypes.byref(hf))
    df = abi.SaoFrame()
    fill(df, [d.ptr for d in d_dst], [d.ptr for d in d_src], [p.shape[1] * isz for p in p_src], [p.shape[1] * isz for p in p_src],
         [d.ptr for d in tabs_dev])
    d_f = batch.DeviceBuffer.from_host(np.frombuffer(bytes(df), np.uint8))
    dev.vvc355_sao_frame_pass(None, bd, d_f.ptr, ctypes.addressof(df))
    dev.vvc355_stream_sync(None)
    for c in range(3):
        got = d_dst[c].to_host(p_src[c].dtype, p_src[c].shape)
        bad = np.argwhere(got[:, :dims[c][0]] != want[c])
        assert len(bad) == 0, f"mode={mode} component {c}: {len(bad)} samples differ, first at {bad[0].tolist()}"
        assert np.all(got[:, dims[c][0]:] == 0x21)
