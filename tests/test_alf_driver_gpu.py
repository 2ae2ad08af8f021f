"""GPU parity of the ALF stage driver (vvc355_alf_frame_pass: per-CTB job descriptors — edge flags from picture borders, tiles and
slices, filter-set selection, clip values — built on the device from the decoder's tables, then the three batched ALF kernels)
vs the oracle's restatement of ff_vvc_alf_filter (vvc_filter.c:1254-1318) on the same tables."""
import ctypes

import numpy as np
import pytest

from conftest import P, rand_pixels
from ffvvc_amd import abi, batch

pytestmark = pytest.mark.gpu


def smooth(rng, shape, bd):
    """Pixels with structure (gradients + noise) so that ALF classes and transposes vary."""
    h, w = shape
    yy, xx = np.mgrid[0:h, 0:w]
    base = ((np.sin(xx / 7.0) + np.cos(yy / 5.0) + np.sin((xx + yy) / 11.0)) * (1 << (bd - 3)) + (1 << (bd - 1)))
    noise = rng.integers(-(1 << (bd - 4)), 1 << (bd - 4), size=shape)
    return np.clip(base + noise, 0, (1 << bd) - 1).astype(np.uint8 if bd == 8 else np.uint16)


@pytest.mark.parametrize("bd,fmt", [(8, (1, 1)), (10, (1, 1)), (12, (1, 1)), (10, (1, 0)), (10, (0, 0))])
@pytest.mark.parametrize("mode", ["across", "slices", "tiles", "both"])
def test_alf_frame_pass(dev, orc, bd, fmt, mode):
    run_alf_frame(dev, orc, bd, fmt, mode, 328, 200, 6, "random")


# Clip indices decide which form of the filter a CTB takes (alf.hip, alf_ctb_kernel): all-zero clip indices (what the fixed filter
# sets and APSs without non-linear clipping carry, vvc_filter.c:1147-1158) run the clamp-free dot-product form, anything else the
# clamped one; "mixed" has one APS of each kind plus the fixed sets, "sparse" a single non-zero index in one class of one APS.
@pytest.mark.parametrize("bd", [8, 10, 12])
@pytest.mark.parametrize("clips", ["zero", "mixed", "sparse"])
@pytest.mark.parametrize("ctb_log2,w,h", [(7, 392, 280), (6, 328, 200), (5, 136, 104)])
def test_alf_frame_pass_clip_forms(dev, orc, bd, clips, ctb_log2, w, h):
    run_alf_frame(dev, orc, bd, (1, 1), "both" if clips == "mixed" else "across", w, h, ctb_log2, clips)


@pytest.mark.parametrize("fmt", [(1, 0), (0, 0)])
def test_alf_frame_pass_clip_forms_other_formats(dev, orc, fmt):
    run_alf_frame(dev, orc, 10, fmt, "across", 392, 280, 7, "zero")


def run_alf_frame(dev, orc, bd, fmt, mode, w, h, ctb_log2, clips):
    hs, vs = fmt                                          # 4:2:0, 4:2:2, 4:4:4
    orc.orc_alf_frame_pass.argtypes = [ctypes.c_int, ctypes.POINTER(abi.AlfFrame)]
    orc.orc_alf_frame_pass.restype = None
    rng = np.random.default_rng(0x5EED0A00 + bd + len(mode) + 16 * hs + 32 * vs + 64 * ctb_log2 + len(clips))
    ctb = 1 << ctb_log2
    cw, ch = (w + ctb - 1) // ctb, (h + ctb - 1) // ctb
    n = cw * ch
    dims = [(w, h), (w >> hs, h >> vs), (w >> hs, h >> vs)]
    isz = 1 if bd == 8 else 2
    src = [smooth(rng, (d[1], d[0]), bd) for d in dims]
    want = [np.full_like(p, 0x21) for p in src]
    p_src = [batch.to_pitched(p) for p in src]
    d_src = [batch.DeviceBuffer.from_host(p) for p in p_src]
    d_dst = [batch.DeviceBuffer.from_host(np.full_like(p, 0x21)) for p in p_src]

    # ---- APS tables: two luma APSs, one chroma APS (8 alternatives), CC-ALF for Cb only on slice 1 (no Cr APS there)
    luma_coeff = [rng.integers(-40, 40, size=(25, 12)).astype(np.int16) for _ in range(2)]
    luma_clip = [rng.integers(0, 4, size=(25, 12)).astype(np.uint8) for _ in range(2)]
    chroma_coeff = rng.integers(-48, 48, size=(8, 6)).astype(np.int16)
    chroma_clip = rng.integers(0, 4, size=(8, 6)).astype(np.uint8)
    if clips == "zero":
        luma_clip = [np.zeros_like(c) for c in luma_clip]
        chroma_clip[:] = 0
    elif clips == "mixed":
        luma_clip[0][:] = 0
        chroma_clip[::2] = 0                                # every other chroma alternative is linear
        luma_coeff[0] = rng.integers(-128, 128, size=(25, 12)).astype(np.int16)      # full coefficient range on the clamp-free form
    elif clips == "sparse":
        luma_clip = [np.zeros_like(c) for c in luma_clip]
        luma_clip[1][int(rng.integers(0, 25)), int(rng.integers(0, 12))] = 3
        chroma_clip[:] = 0
        chroma_clip[3, 2] = 1
        # coefficients far outside the standard's range in a few classes / alternatives: the centre weight -2 * sum(f) no longer fits
        # the 16-bit operand of the clamp-free form, those blocks must take the clamped form (same result, int32 arithmetic)
        luma_coeff[0][::6] = rng.integers(-3000, 3000, size=luma_coeff[0][::6].shape)
        chroma_coeff[5] = rng.integers(-6000, -3000, size=6)
    cc_coeff = [rng.integers(-32, 32, size=(4, 7)).astype(np.int16) for _ in range(2)]
    aps_host = luma_coeff + luma_clip + [chroma_coeff, chroma_clip] + cc_coeff
    aps_dev = [batch.DeviceBuffer.from_host(a) for a in aps_host]

    def slices_for(ptrs):
        sl = (abi.AlfSlice * 3)()
        for i, s in enumerate(sl):
            order = [0, 1] if i != 1 else [1, 0]              # sh_alf_aps_id_luma[] differs per slice
            for k in range(2):
                s.luma_coeff[k], s.luma_clip_idx[k] = ptrs[order[k]], ptrs[2 + order[k]]
            s.chroma_coeff, s.chroma_clip_idx = ptrs[4], ptrs[5]
            s.cc_coeff[0] = ptrs[6]
            s.cc_coeff[1] = ptrs[7] if i != 1 else 0
        return sl

    tab = (abi.AlfCtb * n)()
    for t in tab:
        for c in range(3):
            t.ctb_flag[c] = int(rng.integers(0, 4) > 0)
        t.filt_set_idx_y = int(rng.integers(0, 18))
        for c in range(2):
            t.alt_idx[c], t.cc_idc[c] = int(rng.integers(0, 8)), int(rng.integers(0, 5))
    cut = int(rng.integers(1, n))
    slice_idx = (np.arange(n) >= cut).astype(np.int16) + (np.arange(n) >= min(n - 1, cut + cw + 1)).astype(np.int16)
    col_bd = np.array([0 if x < 3 else 3 for x in range(cw)] + [cw], np.int16)
    row_bd = np.array([0 if y < 2 else 2 for y in range(ch)] + [ch], np.int16)

    def fill(f, dst_ptrs, src_ptrs, dstrides, sstrides, alf_p, slices_p, tp):
        for c in range(3):
            f.dst[c], f.src[c], f.dst_stride[c], f.src_stride[c] = dst_ptrs[c], src_ptrs[c], dstrides[c], sstrides[c]
        f.alf, f.slices = alf_p, slices_p
        f.slice_idx, f.ctb_to_col_bd, f.ctb_to_row_bd = tp
        f.width, f.height, f.ctb_width, f.ctb_height = w, h, cw, ch
        f.ctb_log2, f.hs, f.vs, f.n_comp = ctb_log2, hs, vs, 3
        f.lfase = int(mode in ("across", "tiles"))
        f.lfate = int(mode in ("across", "slices"))

    tab_host = np.frombuffer(bytes(tab), np.uint8).copy()
    tabs_host = [slice_idx, col_bd, row_bd]
    sl_host = slices_for([P(a) for a in aps_host])
    hf = abi.AlfFrame()
    fill(hf, [P(p) for p in want], [P(p) for p in src], [d[0] * isz for d in dims], [d[0] * isz for d in dims],
         P(tab_host), ctypes.addressof(sl_host), [P(t) for t in tabs_host])
    orc.orc_alf_frame_pass(bd, ctypes.byref(hf))

    sl_dev = slices_for([d.ptr for d in aps_dev])
    d_sl = batch.DeviceBuffer.from_host(np.frombuffer(bytes(sl_dev), np.uint8))
    d_tab = batch.DeviceBuffer.from_host(tab_host)
    tabs_dev = [batch.DeviceBuffer.from_host(t) for t in tabs_host]
    df = abi.AlfFrame()
    fill(df, [d.ptr for d in d_dst], [d.ptr for d in d_src], [p.shape[1] * isz for p in p_src], [p.shape[1] * isz for p in p_src],
         d_tab.ptr, d_sl.ptr, [d.ptr for d in tabs_dev])
    d_f = batch.DeviceBuffer.from_host(np.frombuffer(bytes(df), np.uint8))
    work = batch.DeviceBuffer.from_host(np.zeros(dev.vvc355_alf_frame_work_bytes(n), np.uint8))
    if (bd + len(mode) + ctb_log2) % 2:
        dev.vvc355_alf_frame_pass(None, bd, d_f.ptr, ctypes.addressof(df), work.ptr)
    else:                             # the two halves on their own: descriptor builder, then the filter kernels
        dev.vvc355_alf_frame_build(None, bd, d_f.ptr, ctypes.addressof(df), work.ptr)
        dev.vvc355_alf_frame_filter(None, bd, ctypes.addressof(df), work.ptr)
    dev.vvc355_stream_sync(None)
    for c in range(3):
        got = d_dst[c].to_host(p_src[c].dtype, p_src[c].shape)
        bad = np.argwhere(got[:, :dims[c][0]] != want[c])
        assert len(bad) == 0, f"mode={mode} component {c}: {len(bad)} samples differ, first at {bad[0].tolist()}"
        assert np.all(got[:, dims[c][0]:] == 0x21)
    # the case is not vacuous: ALF changed samples in every component
    assert all(np.any(want[c] != src[c]) for c in range(3))
