"""GPU parity of the deblocking stage driver (vvc355_deblock_frame_pass: edge enumeration, QP / beta / tc / filter-length
derivation from the decoder's side tables, then the luma / chroma filters) vs the oracle's restatement of
ff_vvc_deblock_vertical / _horizontal (vvc_filter.c:864-1003) on the same tables."""
import ctypes

import numpy as np
import pytest

import bipred_cases as bc
from conftest import P
from ffvvc_amd import abi, batch

pytestmark = pytest.mark.gpu


def side_tables(rng, w, h, vertical, shift=1):
    """A random but self-consistent transform-block partition and the tables one pass reads (all per 4x4 luma unit)."""
    tw, th = w // 4, h // 4
    tsize = np.zeros((th, tw), np.uint8)                 # tb_width (vertical pass) / tb_height (horizontal) in luma samples
    bs = np.zeros((3, th, tw), np.uint8)
    for by in range(0, h, 32):
        for bx in range(0, w, 32):
            t = int(rng.choice([4, 8, 16, 32]))
            tsize[by // 4:(by + 32) // 4, bx // 4:(bx + 32) // 4] = t
    along, across = (th, tw) if vertical else (tw, th)
    for a in range(along):
        for e in range(1, across):
            y, x = (a, e) if vertical else (e, a)
            pos = e * 4
            t = int(tsize[y, x])
            if pos % t:
                continue                                  # not a transform edge
            bs[0, y, x] = rng.choice([0, 1, 2], p=[0.3, 0.35, 0.35])
            if pos % (8 << shift) == 0:                   # chroma edges live on the 8-sample chroma grid
                bs[1, y, x] = rng.choice([0, 1, 2], p=[0.3, 0.35, 0.35])
                bs[2, y, x] = rng.choice([0, 1, 2], p=[0.3, 0.35, 0.35])
    len_p, len_q = np.zeros((th, tw), np.uint8), np.zeros((th, tw), np.uint8)
    for y in range(th):
        for x in range(tw):
            yp, xp = (y, x - 1) if vertical else (y - 1, x)
            if xp < 0 or yp < 0:
                continue
            sp, sq = int(tsize[yp, xp]), int(tsize[y, x])        # derive_max_filter_length_luma, vvc_filter.c:375-398
            if sp <= 4 or sq <= 4:
                len_p[y, x] = len_q[y, x] = 1
            else:
                len_p[y, x], len_q[y, x] = (7 if sp >= 32 else 3), (7 if sq >= 32 else 3)
    return tsize, bs, len_p, len_q


@pytest.mark.parametrize("bd,fmt", [(8, (1, 1)), (10, (1, 1)), (12, (1, 1)), (10, (1, 0)), (10, (0, 0)), (8, (0, 0))])
def test_deblock_frame_pass(dev, orc, bd, fmt):
    hs, vs = fmt                                          # 4:2:0, 4:2:2, 4:4:4
    orc.orc_deblock_frame_pass.argtypes = [ctypes.c_int, ctypes.POINTER(abi.DeblockFrame)]
    orc.orc_deblock_frame_pass.restype = None
    rng = np.random.default_rng(0x5EED0340 + bd + 16 * hs + 32 * vs)
    w, h, ctb_log2 = 256, 160, 6
    isz = 1 if bd == 8 else 2
    dims = [(w, h), (w >> hs, h >> vs), (w >> hs, h >> vs)]
    planes = []
    for (pw, ph) in dims:
        base = bc.smooth_picture(rng, ph, pw, bd, scale=32).astype(np.int64)
        offs = rng.integers(-(1 << (bd - 6)), (1 << (bd - 6)) + 1, size=(ph // 4, pw // 4))
        planes.append(np.clip(base + np.kron(offs, np.ones((4, 4), np.int64)), 0, (1 << bd) - 1).astype(np.uint8 if bd == 8 else np.uint16))
    want = [p.copy() for p in planes]
    pitched = [batch.to_pitched(p) for p in planes]
    d_planes = [batch.DeviceBuffer.from_host(p) for p in pitched]
    tw, th = w // 4, h // 4
    qp_y = rng.integers(20, 46, size=(h // 8, w // 8)).astype(np.int8)           # min CB 8x8
    qp_c = [rng.integers(20 + 12 * (bd > 8), 46 + 12 * (bd > 8), size=(th, tw)).astype(np.int8) for _ in range(2)]
    ctb_w, ctb_h = (w + 63) // 64, (h + 63) // 64
    dbp = rng.integers(-7, 8, size=(ctb_w * ctb_h, 6)).astype(np.int8)
    changed = 0
    for vertical in (1, 0):                               # all vertical edges first, then all horizontal (vvc_thread.c:159-167)
        shift = hs if vertical else vs
        tsize, bs, len_p, len_q = side_tables(rng, w, h, vertical, shift)
        tb_c = np.maximum(tsize >> shift, 2).astype(np.uint8)
        host_tabs = [bs[0], bs[1], bs[2], len_p, len_q, tb_c, qp_y, qp_c[0], qp_c[1], dbp]
        dev_tabs = [batch.DeviceBuffer.from_host(np.ascontiguousarray(t)) for t in host_tabs]

        def fill(f, planes_ptr, strides, tabs):
            for c in range(3):
                f.plane[c], f.stride[c], f.bs[c] = planes_ptr[c], strides[c], tabs[c]
            f.max_len_p, f.max_len_q, f.tb_size_c, f.qp_y = tabs[3], tabs[4], tabs[5], tabs[6]
            f.qp_c[0], f.qp_c[1], f.db_params = tabs[7], tabs[8], tabs[9]
            f.width, f.height, f.min_tu_width, f.min_cb_width, f.ctb_width = w, h, tw, w // 8, ctb_w
            f.min_cb_log2, f.ctb_log2, f.hs, f.vs, f.n_comp, f.vertical = 3, ctb_log2, hs, vs, 3, vertical
            f.qp_bd_offset = 6 * (bd - 8)
            f.ladf_enabled, f.num_ladf_intervals, f.ladf_lowest_qp_offset = 1, 4, -3
            for k, v in enumerate((2, -1, 4, 0)):
                f.ladf_qp_offset[k] = v
            for k, v in enumerate((0, 1 << (bd - 3), 1 << (bd - 2), 1 << (bd - 1), 0)):
                f.ladf_lower_bound[k] = v

        hf = abi.DeblockFrame()
        before = [p.copy() for p in want]
        keep = [np.ascontiguousarray(t) for t in host_tabs]       # the oracle reads these through raw addresses
        fill(hf, [P(p) for p in want], [dims[c][0] * isz for c in range(3)], [P(t) for t in keep])
        orc.orc_deblock_frame_pass(bd, ctypes.byref(hf))
        changed += sum(int(np.count_nonzero(a != b)) for a, b in zip(before, want))

        df = abi.DeblockFrame()
        fill(df, [d.ptr for d in d_planes], [pitched[c].shape[1] * isz for c in range(3)], [d.ptr for d in dev_tabs])
        d_f = batch.DeviceBuffer.from_host(np.frombuffer(bytes(df), np.uint8))
        dev.vvc355_deblock_frame_pass(None, bd, d_f.ptr, ctypes.addressof(df))
        dev.vvc355_stream_sync(None)
        for c in range(3):
            got = d_planes[c].to_host(pitched[c].dtype, pitched[c].shape)[:, :dims[c][0]]
            bad = np.argwhere(got != want[c])
            assert len(bad) == 0, f"pass vertical={vertical} component {c} bd={bd}: {len(bad)} samples differ, first at {bad[0].tolist()}"
    assert changed > 2000
