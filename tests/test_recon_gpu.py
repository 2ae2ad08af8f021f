"""GPU parity of the RECON stage driver (vvc355_recon_frame_pass): in-order walk of every CTU's coding units — intra prediction
reading neighbours written by earlier blocks of the same and of neighbouring CTUs, then the residual — with CTUs released in
wavefront order, against the oracle's restatement of ff_vvc_reconstruct that runs the same command lists in decoding order
(vvc_intra.c:188-274, :480-527, :574-714; vvc_thread.c:156-184)."""
import ctypes

import numpy as np
import pytest

import bipred_cases as bc
import recon_cases
from conftest import P
from ffvvc_amd import abi, batch

pytestmark = pytest.mark.gpu


def run_case(dev, orc, rng, bd, w, h, ctb_log2, fmt, ticket_order="raster", workgroups=0, **kw):
    orc.orc_recon_frame_pass.argtypes = [ctypes.c_int, ctypes.POINTER(abi.ReconFrame)]
    orc.orc_recon_frame_pass.restype = None
    hs, vs = fmt
    isz = 1 if bd == 8 else 2
    work = recon_cases.ReconWork(rng, w, h, ctb_log2, hs, vs, **kw)
    raster = work.order                  # the oracle walks the CTUs in decoding order whatever order the device takes its tickets in
    if ticket_order == "critical":
        work.order = recon_cases.critical_order(dev, work.ctus, work.ncx, work.ncy)
        assert sorted(work.order.tolist()) == raster.tolist()
    model = recon_cases.ReconWork.lmcs_model(np.random.default_rng(0x1A5C + bd), bd) if kw.get("lmcs") else None
    d_model = batch.DeviceBuffer.from_host(np.frombuffer(bytes(model), np.uint8)) if model is not None else None
    dims = [(w, h), (w >> hs, h >> vs), (w >> hs, h >> vs)]
    planes = [bc.smooth_picture(rng, ph, pw, bd, scale=16) for (pw, ph) in dims]
    resid = rng.integers(-(1 << (bd - 3)), 1 << (bd - 3), size=max(1, work.resid_len)).astype(np.int32)
    inter = rng.integers(0, 1 << bd, size=max(1, work.ciip_len)).astype(planes[0].dtype)          # inter predictions of the CIIP coding units
    # oracle on host copies
    want = [p.copy() for p in planes]
    hc = work.bind(resid.ctypes.data, inter.ctypes.data, isz)
    hf = work.frame([P(p) for p in want], [d[0] * isz for d in dims], hc.ctypes.data, work.ctus.ctypes.data, raster.ctypes.data, 0,
                    work.slice_idx.ctypes.data, work.col_bd.ctypes.data, work.row_bd.ctypes.data, wpp=kw.get("n_slices", 1) > 2, collocated=int(rng.integers(0, 2)),
                    lmcs_ptr=ctypes.addressof(model) if model is not None else 0)
    orc.orc_recon_frame_pass(bd, ctypes.byref(hf))
    # device
    pitched = [batch.to_pitched(p) for p in planes]
    d_planes = [batch.DeviceBuffer.from_host(p) for p in pitched]
    d_res, d_inter = batch.DeviceBuffer.from_host(resid), batch.DeviceBuffer.from_host(inter)
    dcmd = work.bind(d_res.ptr, d_inter.ptr, isz)
    # the commands' pad bytes belong to the pass (its availability pre-pass writes them): whatever the host leaves there must not matter
    dcmd.view(np.uint8).reshape(len(dcmd), -1)[:, 34:40] = np.random.default_rng(len(dcmd)).integers(0, 256, size=(len(dcmd), 6), dtype=np.uint8)
    d_cmds, d_ctus, d_order = batch.DeviceBuffer.from_host(dcmd.view(np.uint8)), batch.DeviceBuffer.from_host(work.ctus.view(np.uint8)), batch.DeviceBuffer.from_host(work.order if len(work.order) else np.zeros(1, np.int32))
    d_state = batch.DeviceBuffer(dev.vvc355_recon_state_bytes(work.ncx * work.ncy))
    d_slice, d_col, d_row = batch.DeviceBuffer.from_host(work.slice_idx), batch.DeviceBuffer.from_host(work.col_bd), batch.DeviceBuffer.from_host(work.row_bd)
    df = work.frame([b.ptr for b in d_planes], [p.shape[1] * isz for p in pitched], d_cmds.ptr, d_ctus.ptr, d_order.ptr, d_state.ptr,
                    d_slice.ptr, d_col.ptr, d_row.ptr, wpp=hf.wpp, collocated=hf.collocated, lmcs_ptr=d_model.ptr if d_model is not None else 0)
    df.workgroups = workgroups
    d_f = batch.DeviceBuffer.from_host(np.frombuffer(bytes(df), np.uint8))
    for rep in range(2):           # twice: the second pass must find its scheduling state reset (and reproduce the result from the same start)
        for b, p in zip(d_planes, pitched):
            dev.vvc355_upload(b.ptr, p.ctypes.data, p.nbytes)
        dev.vvc355_recon_frame_pass(None, bd, d_f.ptr, ctypes.addressof(df))
        dev.vvc355_stream_sync(None)
        for c in range(3):
            got = d_planes[c].to_host(pitched[c].dtype, pitched[c].shape)[:, :dims[c][0]]
            bad = np.argwhere(got != want[c])
            assert len(bad) == 0, (f"pass {rep} bd={bd} {w}x{h} ctb={1 << ctb_log2} fmt={fmt} component {c}: {len(bad)} samples differ, first at (y, x) = {bad[0].tolist()}; "
                                   f"{len(work.cmds)} commands in {len(work.order)} CTUs")
    changed = sum(int((want[c] != planes[c]).sum()) for c in range(3))
    return work, changed


@pytest.mark.parametrize("bd,fmt,ctb_log2", [(10, (1, 1), 7), (8, (1, 1), 6), (12, (0, 0), 5), (10, (1, 0), 6), (8, (1, 1), 7)])
def test_recon_all_intra_wavefront(dev, orc, bd, fmt, ctb_log2):
    """Every coding unit intra: each block's prediction depends on blocks reconstructed just before it, inside the CTU and across
    CTU borders (left, upper-left, upper, upper-right) — any ordering or visibility error shows up as a mismatch."""
    rng = np.random.default_rng(0x5EED0E00 + bd + 16 * ctb_log2 + fmt[0] + 2 * fmt[1])
    work, changed = run_case(dev, orc, rng, bd, 456, 264, ctb_log2, fmt, intra_frac=1.0, n_slices=3 if ctb_log2 < 7 else 1, tiles=ctb_log2 == 6)
    kinds = work.cmds["kind"]
    assert (kinds == abi.RECON_PRED).sum() > 50 and (kinds == abi.RECON_CCLM).sum() > 3 and (kinds == abi.RECON_RESID).sum() > 50
    assert work.cmds["is_mip"].sum() > 0 and work.cmds["isp_split"].sum() > 0 and (work.cmds["joint"] != 0).sum() > 0
    assert changed > 456 * 264 // 2


@pytest.mark.parametrize("ticket_order", ["raster", "critical"])
@pytest.mark.parametrize("bd", [10])
def test_recon_mixed_picture(dev, orc, bd, ticket_order):
    """Inter and intra coding units mixed, whole CTUs without intra work among them (those are skipped by the scheduler)."""
    rng = np.random.default_rng(0x5EED0E77)
    intra_ctu = rng.random(12 * 7) < 0.4
    work, changed = run_case(dev, orc, rng, bd, 1480, 840, 7, (1, 1), ticket_order, intra_frac=0.5, intra_ctu=None, ciip_frac=0.3)
    assert 0 < len(work.order) <= 12 * 7 and (work.cmds["kind"] == abi.RECON_CIIP).sum() > 20
    work, changed = run_case(dev, orc, rng, bd, 1480, 840, 7, (1, 1), ticket_order, intra_ctu=intra_ctu)
    assert 0 < len(work.order) < 12 * 7


@pytest.mark.parametrize("ticket_order", ["raster", "critical"])
def test_recon_more_ctus_than_workgroups(dev, orc, ticket_order):
    """4:2:0 with 32x32 CTUs on a picture of 576 CTUs: the LDS-tile path with the smallest CTU size, and more CTUs than the pass has
    persistent workgroups (256), so every workgroup walks several CTUs and waits on flags raised by workgroups that took later and
    earlier tickets — in raster order and in the order of vvc355_recon_order (tickets then jump between CTU rows)."""
    rng = np.random.default_rng(0x5EED0E99)
    work, changed = run_case(dev, orc, rng, 10, 1024, 576, 5, (1, 1), ticket_order, workgroups=24 if ticket_order == "critical" else 0, intra_frac=1.0, n_slices=2)
    assert len(work.order) == 32 * 18 and changed > 1024 * 576 // 2
    if ticket_order == "critical":
        assert not np.array_equal(work.order, np.sort(work.order))


def test_recon_ticket_order_sparse_intra_clusters(dev, orc):
    """A picture larger than the pass's 256 workgroups can hold at once (2040 CTUs of 32x32), a fifth of its CTUs intra in clusters, the rest
    LIGHT: the critical-path-first order sends workgroups to clusters far down the picture while the first ones are still waiting."""
    rng = np.random.default_rng(0x5EED0E9A)
    w, h = 1920, 1088
    ncx, ncy = w // 32, h // 32
    seeds = rng.random((ncy, ncx)) < 0.05
    intra_ctu = seeds.copy()
    intra_ctu[:, 1:] |= seeds[:, :-1]
    intra_ctu[1:, :] |= seeds[:-1, :]
    work, changed = run_case(dev, orc, rng, 10, w, h, 5, (1, 1), "critical", intra_ctu=intra_ctu.reshape(-1), lmcs=True, resid_ctu=~intra_ctu.reshape(-1), coded_p=0.6)
    assert len(work.order) > 1000 and not np.array_equal(work.order, np.sort(work.order))


@pytest.mark.parametrize("bd,fmt,min_cu", [(10, (0, 0), 4), (8, (1, 1), 8), (12, (1, 0), 8)])
def test_recon_vertical_isp_narrow_transform_blocks(dev, orc, bd, fmt, min_cu):
    """Vertically split ISP coding units: sub-partitions 1 and 2 samples wide are predicted four columns at a time and get their
    residuals added per sub-partition (get_luma_predict_unit, vvc_intra.c:216-226; add_residual with the transform block's size) —
    the RESID command of a block narrower than four samples, at odd and even columns."""
    rng = np.random.default_rng(0x5EED0EA0 + bd + min_cu)
    work, changed = run_case(dev, orc, rng, bd, 328, 200, 6, fmt, intra_frac=1.0, min_cu=min_cu, split=(0.95, 0.8), coded_p=0.9, isp_p=0.6)
    res = work.cmds[work.cmds["kind"] == abi.RECON_RESID]
    luma = res[res["c_idx"] == 0]
    assert (luma["w"] == 2).sum() > 8
    if min_cu == 4:
        assert (luma["w"] == 1).sum() >= 4 and ((luma["w"] == 1) & (luma["x0"] % 2 == 1)).sum() >= 2


@pytest.mark.parametrize("bd,fmt,ctb_log2", [(10, (1, 1), 7), (8, (1, 1), 6), (12, (0, 0), 5), (10, (1, 0), 6)])
def test_recon_lmcs_chroma_residual_scaling(dev, orc, bd, fmt, ctb_log2):
    """Chroma residual scaling (sh_lmcs_used_flag && ph_chroma_residual_scale_flag): every chroma residual of more than 4 samples is
    scaled by the factor of its coding unit's 64x64 unit — derived in the walk from the reconstructed luma left of and above that unit
    (lmcs_derive_chroma_scale, vvc_intra_template.c:390-429; kept per unit, reset per CTU) — including joint Cb-Cr blocks (sign / shift first,
    vvc_intra.c:180-182), across slices and tiles, at picture edges, and for the inter coding units of CTUs whose chroma residuals the walk adds."""
    rng = np.random.default_rng(0x5EED0EB0 + bd + ctb_log2)
    n_ctb = ((456 + (1 << ctb_log2) - 1) >> ctb_log2) * ((264 + (1 << ctb_log2) - 1) >> ctb_log2)
    work, changed = run_case(dev, orc, rng, bd, 456, 264, ctb_log2, fmt, intra_frac=0.6, n_slices=3 if ctb_log2 < 7 else 1, tiles=ctb_log2 == 6, lmcs=True,
                             resid_ctu=np.ones(n_ctb, bool))
    res = work.cmds[work.cmds["kind"] == abi.RECON_RESID]
    assert ((res["joint"] & 8) != 0).sum() > 100 and ((res["joint"] & 9) == 9).sum() > 3 and ((res["joint"] & 8) == 0).sum() > 50


@pytest.mark.parametrize("bd,ctb_log2,w,h", [(10, 7, 1480, 840), (8, 6, 712, 456), (12, 5, 456, 264)])
def test_recon_lmcs_light_ctus(dev, orc, bd, ctb_log2, w, h):
    """Whole CTUs of inter coding units among intra CTUs, chroma residual scaling on: the inter CTUs are LIGHT (vvc355_recon_ctu.flags: not
    staged, residuals added on the planes, waiting only for the luma of an intra neighbour on their left / above, which that neighbour
    publishes ahead of its chroma), the intra CTUs wait for them like for any neighbour — the result must be the in-order walk's."""
    rng = np.random.default_rng(0x5EED0EE0 + bd)
    ctb = 1 << ctb_log2
    n_ctb = ((w + ctb - 1) // ctb) * ((h + ctb - 1) // ctb)
    intra_ctu = rng.random(n_ctb) < 0.3
    work, changed = run_case(dev, orc, rng, bd, w, h, ctb_log2, (1, 1), "critical" if bd == 10 else "raster", intra_ctu=intra_ctu, lmcs=True, resid_ctu=~intra_ctu, coded_p=0.8)
    fl = work.ctus["flags"]
    assert ((fl & abi.RECON_CTU_LIGHT) != 0).sum() > n_ctb // 3 and ((fl & abi.RECON_CTU_LUMA_LEFT) != 0).sum() > 2 and ((fl & abi.RECON_CTU_LUMA_UP) != 0).sum() > 2
    assert (fl[work.ctus["n_cmd"] > 0] == 0).sum() > 2
