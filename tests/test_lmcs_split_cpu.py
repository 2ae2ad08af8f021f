"""CPU (oracle only): the split the frame-resident path uses for chroma residual scaling is equivalent to the reference's order.

The reference adds every coding unit's residual in the RECON stage, CTU by CTU in decoding order (itransform, vvc_intra.c:431-496), so a
chroma block's scale — derived from the reconstructed luma left of and above its 64x64 unit — always sees finished neighbours.  The
frame-resident path takes the chroma residuals of inter CTUs out of that walk where it can: a CTU whose left and upper neighbours are
not reconstructed by the walk (their luma is final before it starts) gets its scaled residuals from one batched launch
(vvc355_lmcs_chroma_resid_batch); the others keep RESID commands in the walk.  Here both orders run through the oracle on the same
picture and must give the same planes (bench.py builds its frame with this rule)."""
import ctypes

import numpy as np

import bipred_cases as bc
import recon_cases
from conftest import P
from ffvvc_amd import abi


def test_batched_scaling_of_independent_inter_ctus_equals_the_in_order_walk(orc):
    orc.orc_recon_frame_pass.argtypes = [ctypes.c_int, ctypes.POINTER(abi.ReconFrame)]
    orc.orc_recon_frame_pass.restype = None
    orc.orc_lmcs_chroma_resid_block.argtypes = [ctypes.c_int, ctypes.POINTER(abi.LmcsResidJob), ctypes.POINTER(abi.LmcsModel)]
    orc.orc_lmcs_chroma_resid_block.restype = None
    bd, w, h, ctb_log2 = 10, 896, 512, 7
    ctb = 1 << ctb_log2
    ncx, ncy = w // ctb, h // ctb
    rng = np.random.default_rng(0x5EED0ED0)
    intra = rng.random(ncx * ncy) < 0.25
    inter = ~intra
    # the rule: an inter CTU stays in the walk if the CTU to its left or above it is reconstructed by the walk
    g = intra.reshape(ncy, ncx)
    nb = np.zeros_like(g)
    nb[:, 1:] |= g[:, :-1]
    nb[1:, :] |= g[:-1, :]
    dep = inter & nb.reshape(-1)
    assert dep.any() and (inter & ~dep).any()
    work = recon_cases.ReconWork(np.random.default_rng(1), w, h, ctb_log2, 1, 1, intra_ctu=intra, lmcs=True, resid_ctu=inter, split=(0.7, 0.2))
    dims = [(w, h), (w // 2, h // 2), (w // 2, h // 2)]
    planes = [bc.smooth_picture(rng, ph, pw, bd, scale=16) for (pw, ph) in dims]         # "after inter prediction and the luma residuals"
    resid = rng.integers(-(1 << (bd - 2)), 1 << (bd - 2), size=work.resid_len).astype(np.int32)
    model = recon_cases.ReconWork.lmcs_model(rng, bd)

    def walk(pl, cmds, ctus, order):
        f = work.frame([P(p) for p in pl], [d[0] * 2 for d in dims], cmds.ctypes.data, ctus.ctypes.data, order.ctypes.data, 0,
                       work.slice_idx.ctypes.data, work.col_bd.ctypes.data, work.row_bd.ctypes.data, lmcs_ptr=ctypes.addressof(model))
        f.n_work = len(order)
        orc.orc_recon_frame_pass(bd, ctypes.byref(f))

    # ---- the reference's order: every residual in the walk
    truth = [p.copy() for p in planes]
    walk(truth, work.bind(resid.ctypes.data), work.ctus, work.order)

    # ---- the split: independent inter CTUs batched first, then the walk without their RESID commands
    split = [p.copy() for p in planes]
    all_cmds = work.bind(resid.ctypes.data)
    keep = np.ones(len(all_cmds), bool)
    n_batched = 0
    for rs in np.nonzero(inter & ~dep)[0]:
        first, n = int(work.ctus[rs]["first_cmd"]), int(work.ctus[rs]["n_cmd"])
        for k in range(first, first + n):
            c = all_cmds[k]
            if c["kind"] != abi.RECON_RESID:
                continue
            assert c["c_idx"] > 0 and c["joint"] & 8
            j = abi.LmcsResidJob()
            ci = int(c["c_idx"])
            j.dst, j.dst_stride = P(split[ci], (int(c["y0"]) >> 1) * dims[ci][0] + (int(c["x0"]) >> 1)), dims[ci][0] * 2
            j.resid, j.luma, j.luma_stride = int(c["resid"]), P(split[0]), w * 2
            j.w, j.h = int(c["w"]), int(c["h"])
            j.x_vpdu, j.y_vpdu = int(c["cu_x0"]) & ~63, int(c["cu_y0"]) & ~63
            j.pic_w, j.pic_h, j.size_y = w, h, 64
            j.avail_l, j.avail_t, j.joint = int(j.x_vpdu > 0), int(j.y_vpdu > 0), int(c["joint"])
            orc.orc_lmcs_chroma_resid_block(bd, ctypes.byref(j), ctypes.byref(model))
            keep[k] = False
            n_batched += 1
    new_index = np.cumsum(keep) - 1
    cmds2 = np.ascontiguousarray(all_cmds[keep])
    ctus2 = work.ctus.copy()
    for rs in range(ncx * ncy):
        first, n = int(work.ctus[rs]["first_cmd"]), int(work.ctus[rs]["n_cmd"])
        kept = int(keep[first:first + n].sum())
        only_marks = kept and not np.any(all_cmds[first:first + n][keep[first:first + n]]["kind"] != abi.RECON_MARK)
        ctus2[rs]["n_cmd"] = 0 if only_marks else kept          # a CTU left with MARK commands only has nothing to do in the walk
        ctus2[rs]["first_cmd"] = int(new_index[first]) + (0 if keep[first] else 1) if kept else 0
    order2 = np.nonzero(ctus2["n_cmd"])[0].astype(np.int32)
    walk(split, cmds2, ctus2, order2)
    assert n_batched > 50 and len(order2) < len(work.order)
    for c in range(3):
        assert np.array_equal(truth[c], split[c]), f"component {c} differs between the in-order walk and the split"
    assert not np.array_equal(truth[1], planes[1])
