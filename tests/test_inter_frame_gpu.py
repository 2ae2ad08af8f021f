"""GPU parity of the inter prediction stage driver (vvc355_inter_frame_pass): the job arrays a kernel writes from the decoder's
tables (coding-unit list, MvField table, reference lists, prediction weight tables) against the oracle's restatement of the same
walk (pred_regular_blk, vvc_inter.c:783-813; derive_weight :129-177), then the predicted picture and the DMVR records."""
import ctypes

import numpy as np
import pytest

import bipred_cases as bc
import inter_frame_cases as ifc
from ffvvc_amd import abi, batch

pytestmark = pytest.mark.gpu
ADDR = ("dst", "ref0", "ref1", "rec", "dst_stride", "ref0_stride", "ref1_stride", "lmcs_lut")       # host planes are packed, device planes pitched


def ref_table(ptrs, strides):
    t = (abi.RefPic * 32)()
    for l in range(2):
        for r in range(2):
            for c in range(3):
                t[l * 16 + r].plane[c] = ptrs[l][r][c]
                t[l * 16 + r].stride[c] = strides[l][r][c]
    return t


@pytest.mark.parametrize("bd,fmt,w,h,mv_range", [(10, (1, 1), 256, 192, 20 * 16), (8, (0, 0), 128, 128, 20 * 16), (12, (1, 0), 192, 128, 20 * 16),
                                                  (10, (1, 1), 128, 64, 300 * 16)])         # last: motion far outside the picture (edge emulation)
def test_inter_frame_pass(dev, orc, bd, fmt, w, h, mv_range):
    orc.orc_inter_frame_pass.argtypes = [ctypes.c_int, ctypes.POINTER(abi.InterFrame)]
    orc.orc_inter_frame_pass.restype = None
    rng = np.random.default_rng(0x1F2A + bd + 7 * fmt[0] + 3 * fmt[1])
    hs, vs = fmt
    isz = 1 if bd == 8 else 2
    dims = [(w, h), (w >> hs, h >> vs), (w >> hs, h >> vs)]
    work = ifc.InterWork(rng, w, h, mv_range=mv_range)
    base = [bc.smooth_picture(rng, ph, pw, bd) for (pw, ph) in dims]
    refs = [[[bc.shifted(base[c], (2 * l - 1) * (r + 1) >> (hs if c else 0), (1 - 2 * l) * (r + 2) >> (vs if c else 0)) for c in range(3)] for r in range(2)] for l in range(2)]
    jl_dt = batch.job_array(abi.BipredJob, 1).dtype
    lut = np.sort(np.random.default_rng(0x10C5 + bd).integers(0, 1 << bd, size=1 << bd)).astype(base[0].dtype)          # fc->ps.lmcs.fwd_lut
    d_lut = batch.DeviceBuffer.from_host(lut)

    # ---- oracle on host memory
    want = [np.zeros((ph, pw), base[0].dtype) for (pw, ph) in dims]
    h_jl, h_jc = np.zeros(work.n_jobs, jl_dt), np.zeros(2 * work.n_jobs, jl_dt)
    h_rec = np.zeros((work.n_jobs, 8), np.int32)
    h_dmvr = work.mvf.copy()                      # tab_dmvr_mvf as the parser leaves it: a copy of the motion field
    h_refs = ref_table([[[refs[l][r][c].ctypes.data for c in range(3)] for r in range(2)] for l in range(2)],
                       [[[refs[l][r][c].shape[1] * isz for c in range(3)] for r in range(2)] for l in range(2)])
    hf = work.frame([p.ctypes.data for p in want], [d[0] * isz for d in dims], work.mvf.ctypes.data, ctypes.addressof(h_refs), work.pus.ctypes.data,
                    ctypes.addressof(work.slices), h_jl.ctypes.data, h_jc.ctypes.data, h_rec.ctypes.data, hs, vs, isz, dmvr_ptr=h_dmvr.ctypes.data, lut_ptr=lut.ctypes.data)
    orc.orc_inter_frame_pass(bd, ctypes.byref(hf))

    # ---- device
    d_refs_planes = [[[batch.DeviceBuffer.from_host(batch.to_pitched(refs[l][r][c])) for c in range(3)] for r in range(2)] for l in range(2)]
    pitches = [batch.plane_pitch(d[0], isz) for d in dims]
    d_dst = [batch.DeviceBuffer.from_host(batch.to_pitched(np.zeros((ph, pw), base[0].dtype))) for (pw, ph) in dims]
    t_refs = ref_table([[[d_refs_planes[l][r][c].ptr for c in range(3)] for r in range(2)] for l in range(2)],
                       [[[pitches[c] for c in range(3)] for r in range(2)] for l in range(2)])
    d_reft = batch.DeviceBuffer.from_host(np.frombuffer(bytes(t_refs), np.uint8))
    d_mvf, d_pus = batch.DeviceBuffer.from_host(work.mvf.view(np.uint8)), batch.DeviceBuffer.from_host(work.pus.view(np.uint8))
    d_sl = batch.DeviceBuffer.from_host(np.frombuffer(bytes(work.slices), np.uint8))
    d_jl, d_jc, d_rec = batch.DeviceBuffer(h_jl.nbytes), batch.DeviceBuffer(h_jc.nbytes), batch.DeviceBuffer(h_rec.nbytes)
    d_dmvr = batch.DeviceBuffer.from_host(work.mvf.view(np.uint8))
    df = work.frame([b.ptr for b in d_dst], pitches, d_mvf.ptr, d_reft.ptr, d_pus.ptr, d_sl.ptr, d_jl.ptr, d_jc.ptr, d_rec.ptr, hs, vs, isz, dmvr_ptr=d_dmvr.ptr, lut_ptr=d_lut.ptr)
    d_f = batch.DeviceBuffer.from_host(np.frombuffer(bytes(df), np.uint8))
    dev.vvc355_inter_frame_pass(None, bd, d_f.ptr, ctypes.addressof(df))
    dev.vvc355_stream_sync(None)

    # the job arrays: every field but the addresses byte for byte; addresses as offsets into their planes
    g_jl, g_jc = d_jl.to_host(jl_dt, (work.n_jobs,)), d_jc.to_host(jl_dt, (2 * work.n_jobs,))
    kinds = set()
    for got, exp, comps in ((g_jl, h_jl, [0] * work.n_jobs), (g_jc, h_jc, [1, 2] * work.n_jobs)):
        for name in jl_dt.names:
            if name in ADDR:
                continue
            assert np.array_equal(got[name], exp[name]), f"job field {name} differs"
        comps = np.array(comps)
        for c in range(3):
            m = comps == c
            if not m.any():
                continue
            ex, ey = exp["x"][m].astype(np.int64), exp["y"][m].astype(np.int64)
            assert np.array_equal(got["dst"][m] - d_dst[c].ptr, ey * pitches[c] + ex * isz) and (got["dst_stride"][m] == pitches[c]).all()
            for l, key in enumerate(("ref0", "ref1")):
                used = m & ((exp["pred_flag"] & (1 << l)) != 0)
                assert (got[key + "_stride"][used] == pitches[c]).all() and (got[key][~used & m] == 0).all()
            assert np.array_equal(exp["dst"][m] - want[c].ctypes.data, ey * dims[c][0] * isz + ex * isz)
        assert np.array_equal((got["rec"] - d_rec.ptr) // 32, (exp["rec"] - h_rec.ctypes.data) // 32)
        # the forward map: on the luma jobs of slice 0's units that are not CIIP, nowhere else
        assert np.array_equal(got["lmcs_lut"] != 0, exp["lmcs_lut"] != 0) and set(np.unique(got["lmcs_lut"])) <= {0, d_lut.ptr}
        if comps[0] == 0:
            assert (exp["lmcs_lut"] != 0).any() and (exp["lmcs_lut"] == 0).any()
        else:
            assert (exp["lmcs_lut"] == 0).all()
        kinds |= {(int(p), int(d), int(b), int(wf)) for p, d, b, wf in zip(exp["pred_flag"], exp["dmvr"], exp["bdof"], exp["weight_flag"])}
    # the case mix: uni / bi, DMVR, BDOF, default / bcw / explicit weights all occur
    if w * h >= 128 * 128:
        assert {k[0] for k in kinds} == {1, 2, 3} and any(k[1] for k in kinds) and any(k[2] for k in kinds) and {k[3] for k in kinds} == {0, 1}
        assert work.n_jobs > len(work.pus)             # units with several sub-blocks / tiles
    for c in range(3):
        got = d_dst[c].to_host(want[c].dtype, (dims[c][1], pitches[c] // isz))[:, :dims[c][0]]
        bad = np.argwhere(got != want[c])
        assert len(bad) == 0, f"component {c}: {len(bad)} samples differ, first at (y, x) = {bad[0].tolist()}"
    assert np.array_equal(d_rec.to_host(np.int32, h_rec.shape)[:, :7], h_rec[:, :7])
    # set_dmvr_info: the refined motion field
    g_dmvr = d_dmvr.to_host(np.uint8, (work.mvf.nbytes,)).view(ifc.MVF_DT).reshape(work.mvf.shape)
    assert np.array_equal(g_dmvr, h_dmvr)
    if mv_range < 100 * 16:
        assert not np.array_equal(h_dmvr, work.mvf)          # some sub-block's motion was refined
