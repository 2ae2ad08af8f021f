"""GPU parity of the regular bi-prediction stage (DMVR search + refined 8-tap MC + BDOF / avg / w_avg, chroma at the refined
motion, edge emulation by clamped reads) vs the oracle's restatement of pred_regular_blk (vvc_inter.c:685-822)."""
import ctypes

import numpy as np
import pytest

import bipred_cases as bc
from conftest import P
from ffvvc_amd import abi, batch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bd,fmt", [(8, (1, 1)), (10, (1, 1)), (12, (1, 1)), (10, (1, 0)), (10, (0, 0)), (8, (0, 0)), (12, (1, 0))])
def test_bipred_frame(dev, orc, bd, fmt):
    bc.bind_oracle(orc)
    hs, vs = fmt                      # 4:2:0, 4:2:2 (hs only) and 4:4:4: the chroma motion fraction and block size follow (vvc_inter.c:209-212)
    rng = np.random.default_rng(0x5EED0800 + bd + 64 * (1 - hs) + 128 * (1 - vs))
    pw, ph = 208, 144
    isz = 1 if bd == 8 else 2
    base = [bc.smooth_picture(rng, ph, pw, bd), bc.smooth_picture(rng, ph >> vs, pw >> hs, bd), bc.smooth_picture(rng, ph >> vs, pw >> hs, bd)]
    # reference 1 = reference 0 displaced by a small whole-sample motion plus noise: the search has something to find
    refs = [[p.copy() for p in base], []]
    for c, p in enumerate(base):
        noise = rng.integers(-3, 4, size=p.shape)
        noise[:, :3 * p.shape[1] // 4] = 0        # left three quarters: an exact displaced copy (zero-cost matches exist)
        q = bc.shifted(p, 2 >> (hs if c else 0), -2 >> (vs if c else 0)).astype(np.int64) + noise
        refs[1].append(np.clip(q, 0, (1 << bd) - 1).astype(p.dtype))
    dims = [(pw, ph), (pw >> hs, ph >> vs), (pw >> hs, ph >> vs)]
    blocks = bc.random_blocks(rng, pw, ph)
    n = len(blocks)

    want = [np.full((d[1], d[0]), 0x21, base[0].dtype) for d in dims]
    want_rec = (abi.BipredResult * n)()
    d_out = [batch.DeviceBuffer.from_host(w_) for w_ in want]
    d_refs = [[batch.DeviceBuffer.from_host(p) for p in r] for r in refs]
    d_rec = batch.DeviceBuffer.from_host(np.zeros(n * 32, np.uint8))
    luma = (abi.BipredJob * n)()
    chroma = (abi.BipredJob * (2 * n))()
    # LMCS: a third of the luma blocks are stored through a forward map (lmcs.filter after predict_inter, vvc_inter.c:888-891)
    lut = np.sort(np.random.default_rng(0x10C5 + bd).integers(0, 1 << bd, size=1 << bd)).astype(base[0].dtype)
    d_lut = batch.DeviceBuffer.from_host(lut)
    host_jobs = []
    for i, (x, y, w, h) in enumerate(blocks):
        kind = i % 5
        if kind == 0:      # motion that matches the displacement between the references (+- a sample): deep minima, early outs
            mv0 = [int(rng.integers(-40, 41)), int(rng.integers(-40, 41))]
            # ref1[y, x] = ref0[y - 2, x + 2]: equal predictions at mv1 = mv0 + (-2, +2) samples; one sample off is found by the search
            mv1 = [mv0[0] - 32 + int(rng.integers(-1, 2)) * 16, mv0[1] + 32 + int(rng.integers(-1, 2)) * 16]
        elif kind == 1:    # far outside the picture: every read is an emulated edge
            mv0 = [int(rng.integers(-4000, 4000)), int(rng.integers(-3000, 3000))]
            mv1 = [int(rng.integers(-4000, 4000)), int(rng.integers(-3000, 3000))]
        else:
            mv0 = [int(v) for v in rng.integers(-200, 201, size=2)]
            mv1 = [int(v) for v in rng.integers(-200, 201, size=2)]
        # whole-sample components (zero fractions) in every combination: the DMVR bilinear stage has one variant per (!!my, !!mx)
        z = i % 9 if kind >= 2 else 0
        if z in (5, 7):
            mv0[0], mv1[0] = mv0[0] // 16 * 16, (mv1[0] // 16 * 16 if z == 7 else mv1[0])
        if z in (6, 7):
            mv0[1], mv1[1] = (mv0[1] // 16 * 16 if z == 7 else mv0[1]), mv1[1] // 16 * 16
        dmvr = int(rng.random() < 0.7 or z in (5, 6, 7))
        bdof = int(rng.random() < 0.6)
        if kind == 0:
            dmvr = bdof = 1
        if w < 8 or h < 8 or h == 12:
            dmvr = bdof = 0                    # the tools need at least 8x8 and multiples of 4 / the DMVR sub-block sizes
        pred_flag = int(rng.choice([1, 2])) if (kind == 2 and rng.random() < 0.5) else 3      # some uni-predicted blocks
        if pred_flag != 3:
            dmvr = bdof = 0
            wf = int(rng.random() < 0.4)
        wf = int(rng.random() < 0.3 and not dmvr)
        for c in range(3):
            j = abi.BipredJob()
            sx, sy = (hs, vs) if c else (0, 0)
            j.x, j.y, j.w, j.h = x >> sx, y >> sy, w >> sx, h >> sy
            j.pic_w, j.pic_h = dims[c]
            for k, v in enumerate(mv0 + mv1):
                j.mv[k] = v
            j.chroma, j.hs, j.vs = int(c > 0), hs, vs
            j.dmvr, j.bdof, j.weight_flag, j.pred_flag = dmvr, bdof, wf, pred_flag
            j.hf_idx = j.vf_idx = int(rng.integers(0, 2)) if c == 0 else 0
            if c:
                j.hf_idx, j.vf_idx = host_jobs[-c][0].hf_idx, host_jobs[-c][0].vf_idx
            j.denom = int(rng.integers(0, 8))
            j.w0, j.w1, j.o0, j.o1 = (int(v) for v in rng.integers(-128, 128, size=4))
            j.dst_stride = j.ref0_stride = j.ref1_stride = dims[c][0] * isz
            host_jobs.append((j, c, i))

    # ---- oracle, in the reference's order: luma block (refines), then its chroma blocks
    for (j, c, i) in host_jobs:
        hj = abi.BipredJob.from_buffer_copy(j)
        hj.dst = P(want[c], hj.y * dims[c][0] + hj.x)
        hj.ref0, hj.ref1 = P(refs[0][c]), P(refs[1][c])
        hj.rec = ctypes.addressof(want_rec[i])
        hj.lmcs_lut = P(lut) if (c == 0 and i % 3 == 0) else 0
        orc.orc_bipred_block(bd, ctypes.byref(hj))

    # ---- device: all luma jobs in one launch, then all chroma jobs
    nl = nc = 0
    for (j, c, i) in host_jobs:
        dj = abi.BipredJob.from_buffer_copy(j)
        dj.dst = d_out[c].ptr + (dj.y * dims[c][0] + dj.x) * isz
        dj.ref0, dj.ref1 = d_refs[0][c].ptr, d_refs[1][c].ptr
        dj.rec = d_rec.ptr + 32 * i
        dj.lmcs_lut = d_lut.ptr if (c == 0 and i % 3 == 0) else 0
        if c == 0:
            luma[nl] = dj; nl += 1
        else:
            chroma[nc] = dj; nc += 1
    if bd == 12:
        # break the (Cb, Cr) pairing of consecutive jobs: the chroma launch must then take its one-job-at-a-time path
        order = rng.permutation(nc)
        shuffled = (abi.BipredJob * (2 * n))()
        for k, o in enumerate(order):
            shuffled[k] = chroma[int(o)]
        chroma = shuffled
    d_l, d_c = batch.jobs_to_device(luma), batch.jobs_to_device(chroma)
    dev.vvc355_bipred_batch(None, bd, d_l.ptr, nl)
    (dev.vvc355_bipred_chroma_batch if bd != 8 else dev.vvc355_bipred_batch)(None, bd, d_c.ptr, nc)     # both entries
    dev.vvc355_stream_sync(None)

    got_rec = d_rec.to_host(np.int32, (n, 8))
    exp_rec = np.frombuffer(bytes(want_rec), np.int32).reshape(n, 8)
    bad = np.argwhere(got_rec[:, :7] != exp_rec[:, :7])
    assert len(bad) == 0, f"record of block {bad[0][0]} {blocks[bad[0][0]]}: got {got_rec[bad[0][0]].tolist()} want {exp_rec[bad[0][0]].tolist()}"
    for c in range(3):
        got = d_out[c].to_host(want[c].dtype, want[c].shape)
        bad = np.argwhere(got != want[c])
        if len(bad):
            # which blocks, with which tools: the pattern of a failure says more than its first sample
            sx, sy = (hs, vs) if c else (0, 0)
            rows = []
            for i, (x, y, w, h) in enumerate(blocks):
                m = got[y >> sy:(y + h) >> sy, x >> sx:(x + w) >> sx] != want[c][y >> sy:(y + h) >> sy, x >> sx:(x + w) >> sx]
                if m.any():
                    j = host_jobs[3 * i][0]
                    ys_, xs_ = np.nonzero(m)
                    rows.append(f"blk{i} {x},{y} {w}x{h} dmvr={j.dmvr} bdof={j.bdof}->{exp_rec[i][4]} wf={j.weight_flag} pf={j.pred_flag} mv={list(j.mv)}->"
                                f"{exp_rec[i][:4].tolist()} n={m.sum()} cols={sorted(set(xs_.tolist()))} rows={sorted(set(ys_.tolist()))}")
            assert False, f"component {c}: {len(bad)} samples differ in {len(rows)} blocks:\n" + "\n".join(rows[:40])
    # the case mix must reach every branch: searches, early terminations, BDOF on and switched off by DMVR
    dm = np.array([hj[0].dmvr for hj in host_jobs[::3]], bool)
    assert np.any(exp_rec[dm, 6] == 1) and np.any(exp_rec[dm, 6] == 0)
    bd_in = np.array([hj[0].bdof for hj in host_jobs[::3]], bool)
    assert np.any(exp_rec[bd_in, 4] == 1) and np.any((exp_rec[:, 4] == 0) & bd_in & dm)
    assert np.any(np.any(exp_rec[:, :4] != np.array([[*hj[0].mv] for hj in host_jobs[::3]]), axis=1))


@pytest.mark.parametrize("bd,fmt", [(10, (1, 1)), (8, (0, 0)), (12, (1, 0))])
def test_gpm_batch(dev, orc, bd, fmt):
    """Geometric-partition blocks (pred_gpm_blk, vvc_inter.c:466-527): two uni-directional predictions blended by a per-sample
    weight mask with signed steps (mirrored masks), luma and chroma, blocks at and beyond the picture edge."""
    orc.orc_gpm_block.argtypes = [ctypes.c_int, ctypes.POINTER(abi.GpmJob)]
    orc.orc_gpm_block.restype = None
    rng = np.random.default_rng(0x5EED0900 + bd)
    hs, vs = fmt
    pw, ph = 176, 112
    isz = 1 if bd == 8 else 2
    dims = [(pw, ph), (pw >> hs, ph >> vs), (pw >> hs, ph >> vs)]
    refs = [[bc.smooth_picture(rng, d[1], d[0], bd) for d in dims] for _ in range(2)]
    mask = rng.integers(0, 9, size=(112, 112)).astype(np.uint8)           # stands for one ff_vvc_gpm_weights[] mask (values 0..8)
    want = [np.full((d[1], d[0]), 0x33, refs[0][0].dtype) for d in dims]
    d_out = [batch.DeviceBuffer.from_host(w_) for w_ in want]
    d_refs = [[batch.DeviceBuffer.from_host(p) for p in r] for r in refs]
    d_mask = batch.DeviceBuffer.from_host(mask)
    lut = np.sort(np.random.default_rng(0x10C5 + bd).integers(0, 1 << bd, size=1 << bd)).astype(refs[0][0].dtype)       # LMCS forward map on every other luma block
    d_lut = batch.DeviceBuffer.from_host(lut)
    jobs = []
    for y in range(0, ph - 15, 16):
        for x in range(0, pw - 15, 16):
            w, h = int(rng.choice([8, 16])), int(rng.choice([8, 16]))
            mv = [int(v) for v in rng.integers(-300, 301, size=4)]
            if rng.random() < 0.2:
                mv = [int(v) for v in rng.integers(-4000, 4001, size=4)]
            mirror = int(rng.integers(0, 3))
            off_x, off_y = int(rng.integers(0, 112 - 64)), int(rng.integers(0, 112 - 64))
            for c in range(3):
                sx, sy = (hs, vs) if c else (0, 0)
                g = abi.GpmJob()
                j = g.base
                j.x, j.y, j.w, j.h = x >> sx, y >> sy, w >> sx, h >> sy
                j.pic_w, j.pic_h = dims[c]
                for k, v in enumerate(mv):
                    j.mv[k] = v
                j.chroma, j.hs, j.vs = int(c > 0), hs, vs
                j.dst_stride = j.ref0_stride = j.ref1_stride = dims[c][0] * isz
                g.step_x, g.step_y = 1 << sx, 112 << sy
                first = off_y * 112 + off_x
                if mirror == 1:
                    g.step_x, first = -g.step_x, off_y * 112 + 111 - off_x
                elif mirror == 2:
                    g.step_y, first = -g.step_y, (111 - off_y) * 112 + off_x
                jobs.append((g, c, first))
    arr = (abi.GpmJob * len(jobs))()
    for i, (g, c, first) in enumerate(jobs):
        hg = abi.GpmJob.from_buffer_copy(g)
        hg.base.dst = P(want[c], hg.base.y * dims[c][0] + hg.base.x)
        hg.base.ref0, hg.base.ref1 = P(refs[0][c]), P(refs[1][c])
        hg.weights = mask.ctypes.data + first
        hg.base.lmcs_lut = P(lut) if (c == 0 and (i // 3) % 2 == 0) else 0
        orc.orc_gpm_block(bd, ctypes.byref(hg))
        g.base.lmcs_lut = d_lut.ptr if (c == 0 and (i // 3) % 2 == 0) else 0
        g.base.dst = d_out[c].ptr + (g.base.y * dims[c][0] + g.base.x) * isz
        g.base.ref0, g.base.ref1 = d_refs[0][c].ptr, d_refs[1][c].ptr
        g.weights = d_mask.ptr + first
        arr[i] = g
    d_jobs = batch.jobs_to_device(arr)
    dev.vvc355_gpm_batch(None, bd, d_jobs.ptr, len(jobs))
    dev.vvc355_stream_sync(None)
    for c in range(3):
        got = d_out[c].to_host(want[c].dtype, want[c].shape)
        bad = np.argwhere(got != want[c])
        assert len(bad) == 0, f"component {c}: {len(bad)} samples differ, first at {bad[0].tolist()}"
    assert np.any(want[0] != 0x33)


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_bdof_rightmost_subblock(dev, orc, bd):
    """BDOF's 6x6 window sums of the right-most 4x4 sub-block of a 16-wide block are the ones whose right-looking pair is anchored on the
    row's last lane (mc_tools.hpp, horizontal box sums by DPP row shifts): pictures that are flat except for texture in a block's last six
    columns make exactly those sums — and nothing else — decide the output."""
    bc.bind_oracle(orc)
    rng = np.random.default_rng(0x5EED0B0F + bd)
    pw, ph, isz = 256, 64, 1 if bd == 8 else 2
    dt = np.uint8 if bd == 8 else np.uint16
    refs = [np.full((ph, pw), 1 << (bd - 1), dt) for _ in range(2)]
    blocks = [(16 * i, 16 * (i % 3)) for i in range(15)]
    for r in refs:
        for (x, y) in blocks:
            r[max(0, y - 4):y + 20, x + 10:x + 20] = rng.integers(0, 1 << bd, size=r[max(0, y - 4):y + 20, x + 10:x + 20].shape)
    want = np.zeros((ph, pw), dt)
    d_refs = [batch.DeviceBuffer.from_host(r) for r in refs]
    d_out = batch.DeviceBuffer.from_host(want)
    arr = (abi.BipredJob * len(blocks))()
    for i, (x, y) in enumerate(blocks):
        j = abi.BipredJob()
        j.x, j.y, j.w, j.h, j.pic_w, j.pic_h, j.bdof, j.pred_flag = x, y, 16, 16, pw, ph, 1, 3
        for k, v in enumerate((int(rng.integers(-3, 4)) * 16, 0, int(rng.integers(-3, 4)) * 16 + int(rng.integers(0, 16)), int(rng.integers(0, 16)))):
            j.mv[k] = v
        j.dst_stride = j.ref0_stride = j.ref1_stride = pw * isz
        hj = abi.BipredJob.from_buffer_copy(j)
        hj.dst, hj.ref0, hj.ref1 = P(want, y * pw + x), P(refs[0]), P(refs[1])
        orc.orc_bipred_block(bd, ctypes.byref(hj))
        j.dst, j.ref0, j.ref1 = d_out.ptr + (y * pw + x) * isz, d_refs[0].ptr, d_refs[1].ptr
        arr[i] = j
    d_jobs = batch.jobs_to_device(arr)
    dev.vvc355_bipred_batch(None, bd, d_jobs.ptr, len(blocks))
    dev.vvc355_stream_sync(None)
    got = d_out.to_host(dt, want.shape)
    bad = np.argwhere(got != want)
    assert len(bad) == 0, f"{len(bad)} samples differ, first at {bad[0].tolist()}"
    # the case is not vacuous: the right-most sub-blocks are not plain averages
    avg = ((refs[0].astype(np.int64) + refs[1] + 1) >> 1)
    assert any(np.any(want[y:y + 16, x + 12:x + 16] != avg[y:y + 16, x + 12:x + 16]) for (x, y) in blocks)
