"""GPU parity of the in-loop filter chain as a real frame runs it, stage drivers only and every stage consuming what the
previous one left in device memory: side tables -> vvc355_deblock_bs_pass -> vvc355_deblock_frame_pass (vertical, then
horizontal, on the bS / filter-length tables just derived) -> vvc355_sao_frame_pass -> vvc355_alf_frame_pass, against the
oracle's restatements chained the same way (vvc_thread.c:159-167 order; vvc_filter.c:154-1318)."""
import ctypes

import numpy as np
import pytest

import bipred_cases as bc
import bs_cases
from conftest import P
from ffvvc_amd import abi, batch

pytestmark = pytest.mark.gpu


class Side:
    """Run the same chain on one side: `up(arr)` makes an array visible to that side and returns (handle, address)."""

    def __init__(self, up):
        self.up, self.keep = up, []

    def a(self, arr):
        h, addr = self.up(np.ascontiguousarray(arr))
        self.keep.append(h)
        return addr


@pytest.mark.parametrize("bd,fmt,ctb_log2", [(10, (1, 1), 6), (8, (1, 1), 7), (10, (0, 0), 5), (12, (1, 0), 6), (10, (0, 0), 7)])
def test_loop_filter_chain(dev, orc, bd, fmt, ctb_log2):
    hs, vs = fmt
    for name, ty in (("orc_deblock_frame_pass", abi.DeblockFrame), ("orc_sao_frame_pass", abi.SaoFrame), ("orc_alf_frame_pass", abi.AlfFrame)):
        getattr(orc, name).argtypes = [ctypes.c_int, ctypes.POINTER(ty)]
        getattr(orc, name).restype = None
    rng = np.random.default_rng(0x5EED0C00 + bd + 16 * hs + 32 * vs + ctb_log2)
    w, h = 328, 200
    ctb = 1 << ctb_log2
    isz = 1 if bd == 8 else 2
    dims = [(w, h), (w >> hs, h >> vs), (w >> hs, h >> vs)]
    t = bs_cases.BsTables(rng, w, h, ctb_log2, n_slices=3, tiles=True, lfase=0, lfate=0, hs=hs, vs=vs, split=(0.9, 0.5), cbf_p=0.5)
    n_ctb = t.cw * t.ch
    rec = []
    for (pw, ph) in dims:
        base = bc.smooth_picture(rng, ph, pw, bd, scale=32).astype(np.int64)
        offs = rng.integers(-(1 << (bd - 6)), (1 << (bd - 6)) + 1, size=(ph // 4, pw // 4))
        rec.append(np.clip(base + np.kron(offs, np.ones((4, 4), np.int64)), 0, (1 << bd) - 1).astype(np.uint8 if bd == 8 else np.uint16))
    qp_y = rng.integers(20, 46, size=(h // 4, w // 4)).astype(np.int8)             # min CB 4x4 as in the bS tables
    qp_c = [rng.integers(20 + 12 * (bd > 8), 46 + 12 * (bd > 8), size=(t.th, t.tw)).astype(np.int8) for _ in range(2)]
    dbp = rng.integers(-7, 8, size=(n_ctb, 6)).astype(np.int8)
    sao_tab = (abi.SaoCtb * n_ctb)()
    alf_tab = (abi.AlfCtb * n_ctb)()
    for i in range(n_ctb):
        for c in range(3):
            sao_tab[i].type_idx[c], sao_tab[i].band_position[c], sao_tab[i].eo_class[c] = int(rng.integers(0, 3)), int(rng.integers(0, 32)), int(rng.integers(0, 4))
            for k in range(1, 5):
                sao_tab[i].offset_val[c][k] = int(rng.integers(-(1 << (bd - 5)) + 1, 1 << (bd - 5)))
            alf_tab[i].ctb_flag[c] = int(rng.integers(0, 4) > 0)
        alf_tab[i].filt_set_idx_y = int(rng.integers(0, 18))
        for c in range(2):
            alf_tab[i].alt_idx[c], alf_tab[i].cc_idc[c] = int(rng.integers(0, 8)), int(rng.integers(0, 5))
    aps = [rng.integers(-40, 40, size=(25, 12)).astype(np.int16), rng.integers(-40, 40, size=(25, 12)).astype(np.int16),
           rng.integers(0, 4, size=(25, 12)).astype(np.uint8), rng.integers(0, 4, size=(25, 12)).astype(np.uint8),
           rng.integers(-48, 48, size=(8, 6)).astype(np.int16), rng.integers(0, 4, size=(8, 6)).astype(np.uint8),
           rng.integers(-32, 32, size=(4, 7)).astype(np.int16), rng.integers(-32, 32, size=(4, 7)).astype(np.int16)]
    n_slices = int(t.slice_idx.max()) + 1

    def run(side, planes_in, pitch_of, launch):
        """planes_in: per component (handle, address) of the reconstructed planes on this side."""
        tabs = {name: side.a(getattr(t, name)) for name in t.IN + t.OUT}
        # ---- boundary strengths
        bsf = t.frame(lambda name: tabs[name])
        launch("bs", bsf)
        # ---- deblocking, vertical edges then horizontal, in place
        qy, qc0, qc1, dp = side.a(qp_y), side.a(qp_c[0]), side.a(qp_c[1]), side.a(dbp)
        for vertical in (1, 0):
            f = abi.DeblockFrame()
            for c in range(3):
                f.plane[c], f.stride[c], f.bs[c] = planes_in[c], pitch_of(c), tabs[f"bs{vertical}{c}"]
            f.max_len_p, f.max_len_q = tabs[f"p{vertical}"], tabs[f"q{vertical}"]
            f.tb_size_c = tabs["tbw1" if vertical else "tbh1"]
            f.qp_y, f.qp_c[0], f.qp_c[1], f.db_params = qy, qc0, qc1, dp
            f.width, f.height, f.min_tu_width, f.min_cb_width, f.ctb_width = w, h, t.tw, w // 4, t.cw
            f.min_cb_log2, f.ctb_log2, f.hs, f.vs, f.n_comp, f.vertical = 2, ctb_log2, hs, vs, 3, vertical
            f.qp_bd_offset = 6 * (bd - 8)
            launch("deblock", f)
        # ---- SAO: deblocked planes -> sao planes
        sao_planes = [side.a(np.full((dims[c][1], pitch_of(c) // isz), 0x21, rec[c].dtype)) for c in range(3)]
        sl, cb, rb = tabs["slice_idx"], tabs["col_bd"], tabs["row_bd"]
        f = abi.SaoFrame()
        for c in range(3):
            f.dst[c], f.src[c], f.dst_stride[c], f.src_stride[c] = sao_planes[c], planes_in[c], pitch_of(c), pitch_of(c)
        f.sao, f.slice_idx, f.ctb_to_col_bd, f.ctb_to_row_bd = side.a(np.frombuffer(bytes(sao_tab), np.uint8)), sl, cb, rb
        f.width, f.height, f.ctb_width, f.ctb_height = w, h, t.cw, t.ch
        f.ctb_log2, f.hs, f.vs, f.n_comp, f.lfase, f.no_tile_filter = ctb_log2, hs, vs, 3, 0, 1
        launch("sao", f)
        # ---- ALF: sao planes -> output planes
        out_handles, out_planes = [], []
        for c in range(3):
            hnd, addr = side.up(np.full((dims[c][1], pitch_of(c) // isz), 0x21, rec[c].dtype))
            out_handles.append(hnd)
            out_planes.append(addr)
        ap = [side.a(a) for a in aps]
        slices = (abi.AlfSlice * n_slices)()
        for i, s in enumerate(slices):
            order = [0, 1] if i != 1 else [1, 0]
            for k in range(2):
                s.luma_coeff[k], s.luma_clip_idx[k] = ap[order[k]], ap[2 + order[k]]
            s.chroma_coeff, s.chroma_clip_idx, s.cc_coeff[0], s.cc_coeff[1] = ap[4], ap[5], ap[6], (ap[7] if i != 1 else 0)
        f = abi.AlfFrame()
        for c in range(3):
            f.dst[c], f.src[c], f.dst_stride[c], f.src_stride[c] = out_planes[c], sao_planes[c], pitch_of(c), pitch_of(c)
        f.alf, f.slices = side.a(np.frombuffer(bytes(alf_tab), np.uint8)), side.a(np.frombuffer(bytes(slices), np.uint8))
        f.slice_idx, f.ctb_to_col_bd, f.ctb_to_row_bd = sl, cb, rb
        f.width, f.height, f.ctb_width, f.ctb_height = w, h, t.cw, t.ch
        f.ctb_log2, f.hs, f.vs, f.n_comp, f.lfase, f.lfate = ctb_log2, hs, vs, 3, 0, 0
        launch("alf", f)
        return out_handles

    # ---- host side: plain numpy arrays, pitch = width
    host = Side(lambda arr: (arr, arr.ctypes.data))
    h_rec = [host.a(p.copy()) for p in rec]

    def host_launch(kind, f):
        if kind == "bs":
            orc.orc_deblock_bs_pass(ctypes.byref(f))
        else:
            getattr(orc, {"deblock": "orc_deblock_frame_pass", "sao": "orc_sao_frame_pass", "alf": "orc_alf_frame_pass"}[kind])(bd, ctypes.byref(f))
    orc.orc_deblock_bs_pass.argtypes = [ctypes.POINTER(abi.BsFrame)]
    orc.orc_deblock_bs_pass.restype = None
    want = run(host, h_rec, lambda c: dims[c][0] * isz, host_launch)

    for name in t.OUT:
        getattr(t, name)[:] = 0xEE                     # the device derives its own tables
    # ---- device side: pitched planes
    pitch = [batch.to_pitched(p).shape[1] * isz for p in rec]

    def dev_up(arr):
        b = batch.DeviceBuffer.from_host(arr)
        return b, b.ptr
    devs = Side(dev_up)
    d_rec = [devs.a(batch.to_pitched(p)) for p in rec]
    work = batch.DeviceBuffer.from_host(np.zeros(dev.vvc355_alf_frame_work_bytes(n_ctb), np.uint8))

    def dev_launch(kind, f):
        d_f = batch.DeviceBuffer.from_host(np.frombuffer(bytes(f), np.uint8))
        devs.keep.append(d_f)
        if kind == "bs":
            dev.vvc355_deblock_bs_pass(None, d_f.ptr, ctypes.addressof(f))
        elif kind == "deblock":
            dev.vvc355_deblock_frame_pass(None, bd, d_f.ptr, ctypes.addressof(f))
        elif kind == "sao":
            dev.vvc355_sao_frame_pass(None, bd, d_f.ptr, ctypes.addressof(f))
        else:
            dev.vvc355_alf_frame_pass(None, bd, d_f.ptr, ctypes.addressof(f), work.ptr)
        dev.vvc355_stream_sync(None)
    got_bufs = run(devs, d_rec, lambda c: pitch[c], dev_launch)
    changed = 0
    for c in range(3):
        got = got_bufs[c].to_host(rec[c].dtype, (dims[c][1], pitch[c] // isz))[:, :dims[c][0]]
        bad = np.argwhere(got != want[c])
        assert len(bad) == 0, f"component {c}: {len(bad)} samples differ after the chain, first at {bad[0].tolist()}"
        changed += int(np.count_nonzero(want[c] != rec[c]))
    assert changed > (w * h) // 4
