#!/usr/bin/env python3
"""bench.py — throughput of the MI355X VVC pixel-kernel path on synthetic 8K 10-bit CTU batches.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``; for N > 1 it is launched under
``python -m torch.distributed.run`` with one rank per GPU.  A *step* is one pass of every stage of the hot path over one
synthetic 8K (7680x4320, 4:2:0, 10-bit) random-access frame = 2040 CTUs of 128x128, all inputs resident in HBM:

    inter prediction (bi-prediction with DMVR search + 8-tap luma MC + BDOF, then 4-tap chroma at the refined motion)
    -> intra prediction (intra CTUs) -> dequant + inverse transform + residual add -> LMCS inverse luma map
    -> boundary strengths from the side tables -> deblock (vertical, horizontal; luma + chroma) -> SAO
    -> ALF (luma classify + filter, chroma, cross-component); the loop filters through their table-driven stage drivers

Timing: W warm-up steps; an untimed pass of K steps with HIP events around every stage (the ``stages`` breakdown and which
stage dominates); then the timed region: exactly K steps between barrier + synchronize, HIP events around the dominant
stage only (``roofline``).  ``--graph`` replays the step as one captured hipGraph in the timed region instead.

Frames are independent, so ranks share nothing: weak scaling, no data-path collective and no RCCL (torch.distributed over gloo
carries one barrier and the max-over-ranks of the elapsed time).  ``python bench.py --gpus N`` on its own starts the N ranks
itself (fresh child processes, one per GPU, before anything touches HIP).  Rank 0 prints ONE JSON line.

PyTorch is plumbing here (device memory, streams, events, the rendezvous); every timed kernel is a hand-written HIP
kernel reached through the C ABI (include/vvc_mi355.h).  The CPU oracle is timed separately, as ``cpu_baseline``.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from ffvvc_amd import abi, batch, sharding  # noqa: E402

INTER_FRAC = 0.8               # fraction of inter CTUs (--inter-frac; 0 = an all-intra picture: BASELINE.json configs[1])
ALF_TABLES = True              # ALF through the stage driver (job descriptors built on the device from ALFParams / APS tables); --alf-jobs: host-built jobs
SAO_TABLES = True              # SAO through the stage driver (parameters derived on the device from per-CTB tables); --sao-jobs: host-built jobs
DEBLOCK_JOBS = False           # deblocking through the stage driver (edge parameters derived from side tables); --deblock-jobs: host-built jobs
AFFINE_FRAC = 0.06             # fraction of the inter CTUs predicted as affine (4x4 sub-blocks + PROF on both lists); --affine-frac
GPM_FRAC = 0.05                # fraction of the regular inter blocks coded as geometric partitions (two uni-predictions + mask blend)
CIIP_FRAC = 0.02               # fraction of the CTUs whose coding units are combined inter / intra (inter prediction aside, planar intra + blend in RECON)
# stages of a picture that read nothing a reference picture writes (bench.py GOP scheduler: launched ahead of the reference waits)
REF_FREE_STAGES = {"inter_mvf_fill", "inter_job_build", "itx_job_build", "intra_tb_dequant_lfnst_itx", "side_tables_fill", "deblock_bs", "alf_job_build"}
RECON_FRAMES = []              # the host copies of every picture's vvc355_recon_frame (the launch reads its grid hint from them)
LMCS = True                    # sh_lmcs_used_flag + ph_chroma_residual_scale_flag on: forward luma map on the inter prediction, chroma residual scaling (--no-lmcs)
MC_TOOLS = 3                   # bit 0: DMVR, bit 1: BDOF on the bi-predicted blocks (profiling aid --mc-tools; the metric uses 3)
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
CTB = 128
# what the chain still lacks of BASELINE.json configs[3] (8K random-access, full in-loop filter chain)
MISSING = ["per-frame upload of the decoder's side tables and parse-side work (CABAC, MV derivation) — the decoder's host side",
           "transform blocks of the CIIP coding units (the frame's CIIP CUs carry no residual)"]

TC_TABLE = [0] * 18 + [3, 4, 4, 4, 4, 5, 5, 5, 5, 7, 7, 8, 9, 10, 10, 11, 13, 14, 15, 17, 19, 21, 24, 25, 29, 33, 36, 41, 45,
                       51, 57, 64, 71, 80, 89, 100, 112, 125, 141, 157, 177, 198, 222, 250, 280, 314, 352, 395]
BETA_TABLE = [0] * 16 + list(range(6, 19)) + list(range(20, 90, 2))


_GENERATED = {}                # host-side generator results shared by the frame objects of a run (same seeds, same partitions)


class Stage:
    """One batched launch (or a few launches of the same kernel) of the hot path over the whole frame."""

    def __init__(self, name, kernel, launch, algorithmic_bytes, writes=(), check=None):
        self.name, self.kernel, self.launch, self.algorithmic_bytes = name, kernel, launch, algorithmic_bytes
        self.writes = list(writes)        # device tensors the stage modifies (snapshotted before / after by the `verified` leg)
        self.check = check                # check(fc, orc, before, after, picks) -> (units checked, units that differ); None = not checked


class Frame:
    """Device-resident planes of one synthetic 4:2:0 frame.  Reference pictures are band-limited content (a coarse random grid,
    bilinearly up-sampled) plus per-16x16-block DC steps and a little noise, the second reference a displaced noisy copy of the
    first: prediction + a sparse residual then looks like a decoded picture — deblocking decisions, DMVR searches / early
    terminations and BDOF all see a realistic mix instead of the early-out paths uniform noise would give (--noise restores it)."""

    PAD = 64            # reference planes carry a 64-sample apron so that MC windows never leave the allocation

    def __init__(self, torch, width, height, bd, seed):
        self.torch, self.width, self.height, self.bd = torch, width, height, bd
        self.isz = 1 if bd == 8 else 2
        self.dtype = torch.uint8 if bd == 8 else torch.int16
        self.gen = torch.Generator(device="cuda")
        self.gen.manual_seed(seed)
        self.ncx, self.ncy = (width + CTB - 1) // CTB, (height + CTB - 1) // CTB
        self.n_ctus = self.ncx * self.ncy
        self.keep = []
        self.dims = [(width, height), (width // 2, height // 2), (width // 2, height // 2)]
        self.host = {}        # data_ptr -> host copy of every uploaded table (what the `verified` leg mirrors for the oracle)
        self.derived = {}     # data_ptr -> expected host content of buffers the device writes itself (not part of the per-frame upload)
        self.arena, self.arena_host, self.arena_used, self.pin = None, None, 0, False
        self.torch_of = {}
        self.noise = False
        self.ref_frames = None                 # (Frame, Frame): reference pictures = those frames' decoded output (GOP mode); None: synthetic pictures
        self.out = self.planes(False)          # the decoded picture (ALF output); allocated first so that other frames can refer to it

    def picture(self, c, pad, like=None, shift=(0, 0), sigma=2.0):
        """One plane of a reference picture (component c) with a `pad`-sample apron; `like` = displaced noisy copy of that plane."""
        torch = self.torch
        w, h = self.dims[c]
        pitch_px = batch.plane_pitch(w + 2 * pad, self.isz) // self.isz
        hh, ww = h + 2 * pad, w + 2 * pad
        scale = 1 << (self.bd - 8)
        if self.noise:
            body = torch.randint(0, 1 << self.bd, (hh, ww), device="cuda", generator=self.gen, dtype=torch.int32).float()
        elif like is None:
            cell = 32 >> (c > 0)
            g = torch.rand((1, 1, hh // cell + 3, ww // cell + 3), device="cuda", generator=self.gen) * ((1 << self.bd) - 1)
            body = torch.nn.functional.interpolate(g, size=(hh, ww), mode="bilinear", align_corners=True)[0, 0]
            blk = 16 >> (c > 0)
            steps = torch.randint(-3 * scale, 3 * scale + 1, (hh // blk + 1, ww // blk + 1), device="cuda", generator=self.gen).float()
            body = body + steps.repeat_interleave(blk, 0).repeat_interleave(blk, 1)[:hh, :ww]
        else:
            body = torch.roll(like[:, :ww].float(), shifts=shift, dims=(0, 1))
        if not self.noise:
            body = body + torch.randn((hh, ww), device="cuda", generator=self.gen) * (sigma * scale / 4)
        t = torch.zeros((hh, pitch_px), device="cuda", dtype=self.dtype)
        t[:, :ww] = body.round().clamp(0, (1 << self.bd) - 1).to(torch.int32).to(self.dtype)
        self.keep.append(t)
        return t

    def plane(self, w, h, random=True, pad=0):
        pitch_px = batch.plane_pitch(w + 2 * pad, self.isz) // self.isz
        shape = (h + 2 * pad, pitch_px)
        if random:
            t = self.torch.randint(0, 1 << self.bd, shape, device="cuda", generator=self.gen, dtype=self.torch.int32).to(self.dtype)
        else:
            t = self.torch.zeros(shape, device="cuda", dtype=self.dtype)
        self.keep.append(t)
        return t

    def planes(self, random=True, pad=0):
        return [self.plane(w, h, random, pad) for (w, h) in self.dims]

    def i16_plane(self, w, h):
        t = self.torch.zeros((h, (w + 127) // 128 * 128), device="cuda", dtype=self.torch.int16)
        self.keep.append(t)
        return t

    def upload(self, arr, per_frame=True):
        """A table / descriptor array in HBM.  per_frame: what a decoder sends for every picture (counted and copied by --with-upload);
        False: buffers the device itself writes every step (stage outputs, tables built from records) — kept with their expected host
        content for the `verified` leg only."""
        src = np.ascontiguousarray(arr)
        if per_frame:
            # everything a picture's host side sends lives in one device arena with a pinned host twin: one copy per picture
            if self.arena is None:
                cap = max(1 << 20, int(self.width * self.height * 2.6))
                self.arena = self.torch.zeros(cap, dtype=self.torch.uint8, device="cuda")
                self.arena_host = self.torch.zeros(cap, dtype=self.torch.uint8).pin_memory() if self.pin else self.torch.zeros(cap, dtype=self.torch.uint8)
                self.keep.append(self.arena)
            at = (self.arena_used + 255) & ~255
            if at + src.nbytes <= self.arena.numel():
                self.arena_used = at + src.nbytes
                hv = self.arena_host.numpy()[at:at + src.nbytes]
                hv[:] = src.reshape(-1).view(np.uint8)
                host = hv.view(src.dtype).reshape(src.shape)
                t = self.arena[at:at + src.nbytes].view(getattr(self.torch, str(src.dtype))).view(src.shape) if src.dtype.kind != "V" else self.arena[at:at + src.nbytes]
                t.copy_(self.torch.from_numpy(host) if src.dtype.kind != "V" else self.torch.from_numpy(hv))
                self.host[t.data_ptr()] = host
                self.torch_of[t.data_ptr()] = t
                return t
        host = src.copy()
        t = self.torch.from_numpy(host).cuda()
        self.keep.append(t)
        (self.host if per_frame else self.derived)[t.data_ptr()] = host
        self.torch_of[t.data_ptr()] = t
        return t

    def pitch(self, t):
        return t.stride(0) * t.element_size()


def ifc_mvf_dtype():
    """The reference's MvField (vvc_ctu.h:195-202) as a numpy record."""
    return np.dtype([("mv", "<i4", (2, 2)), ("ref_idx", "i1", (2,)), ("hpel_if_idx", "u1"), ("bcw_idx", "u1"), ("pred_flag", "u1"), ("ciip_flag", "u1"), ("pad_", "u1", (2,))])


def ifc_pu_dtype():
    """vvc355_inter_pu as a numpy record."""
    return np.dtype([("x0", "<i2"), ("y0", "<i2"), ("cb_width", "<i2"), ("cb_height", "<i2"), ("num_sb_x", "u1"), ("num_sb_y", "u1"), ("dmvr_flag", "u1"),
                     ("bdof_flag", "u1"), ("ciip_flag", "u1"), ("hpel_if_idx", "u1"), ("slice", "u1"), ("pad_", "u1"), ("first_job", "<u4")])


def alf_filter_sets(rng, n_sets):
    """APS-like luma filter sets: 25 filters x 12 int8-range coefficients, clip indices 0..3, a random class map."""
    return [(rng.integers(-128, 128, size=(25, 12)).astype(np.int16),
             rng.integers(0, 4, size=(25, 12)).astype(np.uint8),
             rng.permutation(25).astype(np.uint8)) for _ in range(n_sets)]


def build_chain(lib, torch, fr):
    """Build every stage's job descriptors once (host side, numpy) and return the list of Stage objects."""
    if os.path.join(ROOT, "tests") not in sys.path:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
    rng = np.random.default_rng(0x5EED0001)
    bd, isz = fr.bd, fr.isz
    ptr = lambda t: t.data_ptr()          # noqa: E731
    chain = []

    if fr.ref_frames is None:
        # two synthetic reference pictures; the second is the first displaced by (+2, -2) luma samples plus noise: ref1[y, x] = ref0[y + 2, x - 2]
        ref0 = [fr.picture(c, Frame.PAD) for c in range(3)]
        ref1 = [fr.picture(c, Frame.PAD, like=ref0[c], shift=(-2 >> (c > 0), 2 >> (c > 0))) for c in range(3)]
        ref = [ref0, ref1]
        ref_org = lambda r, c: ptr(ref[r][c]) + Frame.PAD * fr.pitch(ref[r][c]) + Frame.PAD * isz          # noqa: E731  sample (0, 0) of the picture
    else:
        # the decoded pictures (ALF output planes) of two other frames of the GOP: what the DPB hands an inter picture.  No apron: every
        # prediction kernel of the chain reads at coordinates clamped to the picture (the reference's edge emulation)
        ref = [fr.ref_frames[0].out, fr.ref_frames[1].out]
        ref_org = lambda r, c: ptr(ref[r][c])          # noqa: E731
    ref_pitch = lambda r, c: fr.pitch(ref[r][c])       # noqa: E731
    fr.refs = ref
    rec = fr.planes(False)                                               # prediction -> reconstruction -> deblocked
    sao = fr.planes(False)
    out = fr.out
    pitches = [fr.pitch(t) for t in rec]
    rec_ptrs = [ptr(t) for t in rec]

    # CTU kinds: 80 % inter, 20 % intra; of the inter CTUs a few are affine and a few combined inter / intra (CIIP)
    ctu_inter = rng.random(fr.n_ctus) < INTER_FRAC
    ctu_ciip = ctu_inter & (rng.random(fr.n_ctus) < CIIP_FRAC / max(INTER_FRAC, 1e-9))
    # LMCS (own generator: the other draws stay what they were): the forward luma map of the inter prediction and the model of the chroma
    # residual scaling.  The scale of a 64x64 unit comes from the reconstructed luma left of and above it, so the chroma residuals of an
    # inter CTU whose left or upper neighbour is reconstructed by the in-order pass (intra, CIIP) are added by that pass, behind the
    # neighbour; the other inter CTUs' chroma residuals are scaled in one batched launch before it.
    rng_l = np.random.default_rng(0x5EED0ECF)
    px_np = np.uint8 if bd == 8 else np.uint16
    d_fwd = fr.upload(np.sort(rng_l.integers(0, 1 << bd, size=1 << bd)).astype(px_np)) if LMCS else None
    fwd_ptr = ptr(d_fwd) if LMCS else 0
    import recon_cases as _rc
    lmcs_model = _rc.ReconWork.lmcs_model(rng_l, bd) if LMCS else None
    d_model = fr.upload(np.frombuffer(bytes(lmcs_model), np.uint8)) if LMCS else None
    in_order = (~ctu_inter | ctu_ciip).reshape(fr.ncy, fr.ncx)              # CTUs whose luma the in-order pass writes
    nb = np.zeros_like(in_order)
    nb[:, 1:] |= in_order[:, :-1]
    nb[1:, :] |= in_order[:-1, :]
    ctu_dep = (ctu_inter & ~ctu_ciip & nb.reshape(-1)) if LMCS else np.zeros(fr.n_ctus, bool)

    # ---------------------------------------------------------------- inter prediction: regular bi-predicted 16x16 luma sub-blocks
    # with DMVR and BDOF switched on (search, parametric refinement, 8-tap MC at the refined motion, BDOF), then their 8x8
    # chroma blocks (4-tap) at the refined motion.  Motion is random within +-24 samples, so part of the blocks at the picture
    # border exercise the edge emulation; uniformly random references make every DMVR search run to the end (worst case).
    # optionally (--affine-frac, not part of the metric's workload) some of the inter CTUs are affine instead: 4x4 luma
    # sub-blocks, bi-predicted with PROF on both lists, chroma at the sub-block motion
    ctu_affine = ctu_inter & ~ctu_ciip & (rng.random(fr.n_ctus) < AFFINE_FRAC)
    bs = 16
    x0, y0 = batch.block_grid(fr.width // bs * bs, fr.height // bs * bs, bs, bs)
    ctu_of = (y0 // CTB) * fr.ncx + (x0 // CTB)
    aff_blk = ctu_affine[ctu_of]
    inter = ctu_inter[ctu_of] & ~aff_blk & ~ctu_ciip[ctu_of]
    gpm_blk = inter & (rng.random(len(x0)) < GPM_FRAC)
    inter &= ~gpm_blk
    xa0, ya0 = x0[aff_blk], y0[aff_blk]
    xg0, yg0 = x0[gpm_blk], y0[gpm_blk]
    ctu_of_gpm = ctu_of[gpm_blk]
    x0, y0 = x0[inter], y0[inter]
    n_blk = len(x0)
    if n_blk:            # (an all-intra picture has no regular inter blocks: --inter-frac 0)
        d_rec = fr.upload(np.zeros(n_blk * 32, np.uint8), per_frame=False)          # device scratch: the DMVR records
        mv = rng.integers(-24 * 16, 24 * 16 + 1, size=(n_blk, 4))
        if not fr.noise:
            # half of the blocks carry motion that is consistent with the displacement between the two references, up to one sample
            # and a fraction: their search finds a real minimum (early terminations, BDOF switched off by a low cost); the others
            # point at unrelated content (full searches, BDOF on)
            match = rng.random(n_blk) < 0.5
            mv[match, 2] = mv[match, 0] - 32 + rng.integers(-1, 2, size=match.sum()) * 16 + rng.integers(-3, 4, size=match.sum())
            mv[match, 3] = mv[match, 1] + 32 + rng.integers(-1, 2, size=match.sum()) * 16 + rng.integers(-3, 4, size=match.sum())
        ctu_of_blk = ctu_of[inter]
        bj = []
        for c, (w, h) in enumerate(fr.dims):
            sh = 1 if c else 0
            j = batch.job_array(abi.BipredJob, n_blk)
            j["dst"] = ptr(rec[c]) + (y0 >> sh) * fr.pitch(rec[c]) + (x0 >> sh) * isz
            j["dst_stride"] = fr.pitch(rec[c])
            for r, key in enumerate(("ref0", "ref1")):
                j[key] = ref_org(r, c)        # sample (0, 0) of the picture
                j[key + "_stride"] = ref_pitch(r, c)
            j["rec"] = ptr(d_rec) + np.arange(n_blk, dtype=np.int64) * 32
            j["mv"] = mv
            j["x"], j["y"], j["w"], j["h"] = x0 >> sh, y0 >> sh, bs >> sh, bs >> sh
            j["pic_w"], j["pic_h"] = w, h
            j["chroma"], j["hs"], j["vs"] = int(c > 0), 1, 1
            j["dmvr"], j["bdof"] = MC_TOOLS & 1, ((MC_TOOLS >> 1) & 1) if c == 0 else 0
            j["pred_flag"] = 3
            j["lmcs_lut"] = fwd_ptr if c == 0 else 0
            bj.append(j)
        # chroma jobs interleaved Cb, Cr, Cb, Cr, ...: the chroma launch predicts the two planes of a sub-block in one wave
        luma_jobs = bj[0]
        chroma_jobs = np.empty(2 * n_blk, dtype=bj[1].dtype)
        chroma_jobs[0::2], chroma_jobs[1::2] = bj[1], bj[2]
        n_bl, n_bc = len(luma_jobs), len(chroma_jobs)
        # The job arrays above are the host's expectation only.  What the device runs is written by vvc355_inter_frame_build from the
        # decoder-side tables: the MvField table (one entry per 4x4 luma block), the reference picture lists, the slice's weight tables
        # and one 20-byte record per coding unit (here: every 16x16 block is a bi-predicted coding unit of one sub-block).
        pic_w, pic_h = fr.dims[0]
        mvf = np.zeros((pic_h // 4 + 1, pic_w // 4), ifc_mvf_dtype())
        for dy in range(bs // 4):
            for dx in range(bs // 4):
                e = mvf[y0 // 4 + dy, x0 // 4 + dx]
                e["mv"] = mv.reshape(-1, 2, 2)
                e["pred_flag"] = 3
                mvf[y0 // 4 + dy, x0 // 4 + dx] = e
        pus = np.zeros(n_blk, ifc_pu_dtype())
        pus["x0"], pus["y0"], pus["cb_width"], pus["cb_height"] = x0, y0, bs, bs
        pus["num_sb_x"] = pus["num_sb_y"] = 1
        pus["dmvr_flag"], pus["bdof_flag"] = MC_TOOLS & 1, (MC_TOOLS >> 1) & 1
        pus["first_job"] = np.arange(n_blk)
        reft = (abi.RefPic * 32)()
        for l in range(2):
            for c in range(3):
                reft[l * 16].plane[c] = ref_org(l, c)
                reft[l * 16].stride[c] = ref_pitch(l, c)
        # the MvField table itself is written on the device from one 32-byte record per prediction unit (vvc355_tab_fill_pass)
        d_mvf, d_pus = fr.upload(mvf.view(np.uint8).reshape(-1), per_frame=False), fr.upload(pus.view(np.uint8))
        mvrec = np.zeros(n_blk, np.dtype([("x0", "<i2"), ("y0", "<i2"), ("w", "u1"), ("h", "u1"), ("pad_", "u1", (2,)), ("mvf", "V24")]))
        mvrec["x0"], mvrec["y0"], mvrec["w"], mvrec["h"] = x0, y0, bs, bs
        mvrec["mvf"] = np.ascontiguousarray(mvf[y0 // 4, x0 // 4]).view("V24").reshape(-1)
        import bs_cases as _bsc
        mvrec, mv_first = _bsc.BsTables.group_per_ctu(mvrec, 7, fr.ncx, fr.n_ctus)
        d_mvrec, d_mvfirst = fr.upload(mvrec.view(np.uint8)), fr.upload(mv_first)
        mfill = abi.TabFill()
        mfill.mv, mfill.n_mv, mfill.mvf, mfill.mvf_pitch, mfill.unit_pitch = ptr(d_mvrec), n_blk, ptr(d_mvf), pic_w // 4, pic_w // 4
        mfill.ctu_first_mv, mfill.ctb_log2, mfill.width, mfill.height, mfill.ctb_width, mfill.ctb_height = ptr(d_mvfirst), 7, fr.width, fr.height, fr.ncx, fr.ncy
        d_mfill = fr.upload(np.frombuffer(bytes(mfill), np.uint8))
        fr.keep.append(mfill)
        d_mvf.zero_()
        lib.vvc355_tab_fill_pass(None, ptr(d_mfill), ctypes.addressof(mfill))

        def check_mvf_fill(fc, orc, env):
            return n_blk, int(not np.array_equal(env.after[ptr(d_mvf)], fr.derived[ptr(d_mvf)]))

        chain.append(Stage("inter_mvf_fill", "tabfill_kernel", lambda st: lib.vvc355_tab_fill_pass(st, ptr(d_mfill), ctypes.addressof(mfill)),
                           n_blk * 16 * 24, writes=[d_mvf], check=check_mvf_fill))
        islice = abi.InterSlice()
        islice.lmcs_used = int(LMCS)
        d_reft, d_slices = fr.upload(np.frombuffer(bytes(reft), np.uint8)), fr.upload(np.frombuffer(bytes(islice), np.uint8))
        d_bl = torch.zeros(luma_jobs.nbytes, dtype=torch.uint8, device="cuda")
        d_bc = torch.zeros(chroma_jobs.nbytes, dtype=torch.uint8, device="cuda")
        fr.keep += [d_bl, d_bc]
        inf = abi.InterFrame()
        for c in range(3):
            inf.dst[c], inf.dst_stride[c] = ptr(rec[c]), fr.pitch(rec[c])
        inf.mvf, inf.refs, inf.pus, inf.slices = ptr(d_mvf), ptr(d_reft), ptr(d_pus), ptr(d_slices)
        inf.jobs_luma, inf.jobs_chroma, inf.records = ptr(d_bl), ptr(d_bc), ptr(d_rec)
        inf.mvf_stride, inf.n_pus, inf.n_jobs = pic_w // 4, n_blk, n_blk
        inf.width, inf.height = pic_w, pic_h
        inf.hs, inf.vs, inf.chroma_format_idc, inf.pixel_shift = 1, 1, 1, int(isz == 2)
        inf.lmcs_fwd_lut = fwd_ptr
        d_inf = fr.upload(np.frombuffer(bytes(inf), np.uint8))
        fr.keep.append(inf)

        def check_build(fc, orc, env):
            got_l, got_c = env.after[ptr(d_bl)].view(luma_jobs.dtype), env.after[ptr(d_bc)].view(chroma_jobs.dtype)
            return n_bl + n_bc, int((got_l != luma_jobs).sum() + (got_c != chroma_jobs).sum())

        # filled once here as well, so that a profiling run with --only on the prediction stages alone finds valid jobs
        lib.vvc355_inter_frame_build(None, ptr(d_inf), ctypes.addressof(inf))
        lib.vvc355_stream_sync(None)
        chain.append(Stage("inter_job_build", "inter_build_kernel", lambda st: lib.vvc355_inter_frame_build(st, ptr(d_inf), ctypes.addressof(inf)),
                           n_blk * 3 * ctypes.sizeof(abi.BipredJob), writes=[d_bl, d_bc], check=check_build))
        inter_samples = n_blk * (bs * bs + 2 * (bs // 2) ** 2)

        # luma refines the motion (DMVR) and writes the records; chroma of both planes follows at the refined motion.
        # algorithmic bytes: two reference samples read + one sample written
        def check_luma(fc, orc, env):
            idx = np.nonzero(np.isin(ctu_of_blk, env.picks))[0]
            recs = env.after[ptr(d_rec)].view(np.int32).reshape(-1, 8)
            bad = fc.check_bipred(orc, bd, luma_jobs, idx, env.mirror, [env.after[rec_ptrs[0]], None, None], rec_ptrs, pitches, recs, ptr(d_rec))
            env.stats["dmvr_searched_fraction"] = float(recs[:, 6].mean())
            env.stats["bdof_applied_fraction"] = float(recs[:, 4].mean())
            return len(idx), bad

        def check_chroma(fc, orc, env):
            idx = np.nonzero(np.isin(np.repeat(ctu_of_blk, 2), env.picks))[0]
            recs = env.snap(d_rec).view(np.int32).reshape(-1, 8)
            bad = fc.check_bipred(orc, bd, chroma_jobs, idx, env.mirror, [None, env.after[rec_ptrs[1]], env.after[rec_ptrs[2]]], rec_ptrs, pitches, recs, ptr(d_rec))
            return len(idx), bad

        chain.append(Stage("inter_pred_luma_dmvr_bdof", f"bipred_kernel<{bd}, true>", lambda st: lib.vvc355_bipred_batch(st, bd, ptr(d_bl), n_bl),
                           n_blk * bs * bs * 3 * isz, writes=[rec[0], d_rec], check=check_luma))
        chain.append(Stage("inter_pred_chroma", f"bipred_chroma_pair_kernel<{bd}>", lambda st: lib.vvc355_bipred_chroma_batch(st, bd, ptr(d_bc), n_bc),
                           n_blk * 2 * (bs // 2) ** 2 * 3 * isz, writes=[rec[1], rec[2]], check=check_chroma))

    if len(xg0):
        # geometric-partition blocks: two uni-directional predictions per component, blended by a 112 x 112 weight mask the job
        # addresses with signed steps (the reference's mirrored masks); one mask with smooth diagonal weights stands for the table
        gy, gx = np.mgrid[0:112, 0:112]
        gmask = np.clip((gx - gy) // 4 + 4, 0, 8).astype(np.uint8)
        d_gmask = fr.upload(gmask)
        n_g = len(xg0)
        gmv = rng.integers(-24 * 16, 24 * 16 + 1, size=(n_g, 4))
        goff = rng.integers(0, 112 - 32, size=(n_g, 2))
        gmir = rng.integers(0, 3, size=n_g)
        gj = []
        for c, (w, h) in enumerate(fr.dims):
            sh = 1 if c else 0
            j = batch.job_array(abi.GpmJob, n_g)
            b = j["base"]
            b["dst"] = ptr(rec[c]) + (yg0 >> sh) * fr.pitch(rec[c]) + (xg0 >> sh) * isz
            b["dst_stride"] = fr.pitch(rec[c])
            for r, key in enumerate(("ref0", "ref1")):
                b[key] = ref_org(r, c)
                b[key + "_stride"] = ref_pitch(r, c)
            b["mv"] = gmv
            b["x"], b["y"], b["w"], b["h"], b["pic_w"], b["pic_h"] = xg0 >> sh, yg0 >> sh, 16 >> sh, 16 >> sh, w, h
            b["chroma"], b["hs"], b["vs"] = int(c > 0), 1, 1
            b["lmcs_lut"] = fwd_ptr if c == 0 else 0
            first = goff[:, 1] * 112 + goff[:, 0]
            j["step_x"], j["step_y"] = 1 << sh, 112 << sh
            m1, m2 = gmir == 1, gmir == 2
            j["step_x"][m1] = -(1 << sh)
            first = np.where(m1, goff[:, 1] * 112 + 111 - goff[:, 0], first)
            j["step_y"][m2] = -(112 << sh)
            first = np.where(m2, (111 - goff[:, 1]) * 112 + goff[:, 0], first)
            j["weights"] = ptr(d_gmask) + first
            gj.append(j)
        gpm_jobs = np.concatenate(gj)
        d_gj = fr.upload(gpm_jobs.view(np.uint8))
        n_gj = len(gpm_jobs)

        def check_gpm(fc, orc, env):
            orc.orc_gpm_block.argtypes = [ctypes.c_int, ctypes.POINTER(abi.GpmJob)]
            orc.orc_gpm_block.restype = None
            idx = np.nonzero(np.isin(np.tile(ctu_of_gpm, 3), env.picks) | (np.arange(n_gj) % 53 == 0))[0]
            dt = np.uint8 if bd == 8 else np.uint16
            bad = 0
            for i in idx:
                g = abi.GpmJob.from_buffer_copy(gpm_jobs[i].tobytes())
                c = next(k for k, p_ in enumerate(rec_ptrs) if p_ <= g.base.dst < p_ + env.after[p_].nbytes)
                off = int(g.base.dst) - rec_ptrs[c]
                x_, y_ = (off % pitches[c]) // isz, off // pitches[c]
                blk = np.zeros((g.base.h, g.base.w), dt)
                g.base.dst, g.base.dst_stride = blk.ctypes.data, g.base.w * isz
                g.base.ref0, g.base.ref1 = env.mirror.host_addr(g.base.ref0), env.mirror.host_addr(g.base.ref1)
                g.base.lmcs_lut = env.mirror.host_addr(g.base.lmcs_lut) if g.base.lmcs_lut else 0
                g.weights = env.mirror.host_addr(g.weights)
                orc.orc_gpm_block(bd, ctypes.byref(g))
                bad += not np.array_equal(env.after[rec_ptrs[c]][y_:y_ + g.base.h, x_:x_ + g.base.w], blk)
            return len(idx), bad

        chain.append(Stage("inter_pred_gpm", f"gpm_kernel<{bd}>", lambda st: lib.vvc355_gpm_batch(st, bd, ptr(d_gj), n_gj),
                           n_g * 384 * 3 * isz, writes=rec, check=check_gpm))

    if len(xa0):
        # affine CTUs: every 16x16 area = 16 luma sub-blocks of 4x4 (own motion, PROF on both lists) + its 8x8 chroma blocks
        sx, sy = np.meshgrid(np.arange(0, 16, 4), np.arange(0, 16, 4))
        ax = (xa0[:, None] + sx.ravel()[None, :]).ravel()
        ay = (ya0[:, None] + sy.ravel()[None, :]).ravel()
        n_sb = len(ax)
        base_mv = np.repeat(rng.integers(-24 * 16, 24 * 16 + 1, size=(len(xa0), 4)), 16, axis=0)
        afj = batch.job_array(abi.AffineJob, n_sb)
        afj["dst"] = ptr(rec[0]) + ay * fr.pitch(rec[0]) + ax * isz
        afj["dst_stride"] = fr.pitch(rec[0])
        for r, key in enumerate(("ref0", "ref1")):
            afj[key] = ref_org(r, 0)
            afj[key + "_stride"] = ref_pitch(r, 0)
        d_dmv = fr.upload(rng.integers(-32, 33, size=(2, 2, 16)).astype(np.int16))
        afj["diff_mv"] = ptr(d_dmv)
        afj["mv"] = base_mv + rng.integers(-8, 9, size=(n_sb, 4))
        afj["x"], afj["y"], afj["pic_w"], afj["pic_h"] = ax, ay, fr.width, fr.height
        afj["pred_flag"], afj["prof0"], afj["prof1"] = 3, 1, 1
        afj["lmcs_lut"] = fwd_ptr
        acj = []
        for c in (1, 2):
            j = batch.job_array(abi.BipredJob, len(xa0))
            j["dst"] = ptr(rec[c]) + (ya0 >> 1) * fr.pitch(rec[c]) + (xa0 >> 1) * isz
            j["dst_stride"] = fr.pitch(rec[c])
            for r, key in enumerate(("ref0", "ref1")):
                j[key] = ref_org(r, c)
                j[key + "_stride"] = ref_pitch(r, c)
            j["mv"] = base_mv[::16]
            j["x"], j["y"], j["w"], j["h"] = xa0 >> 1, ya0 >> 1, 8, 8
            j["pic_w"], j["pic_h"] = fr.dims[c]
            j["chroma"], j["hs"], j["vs"] = 1, 1, 1
            acj.append(j)
        afc = np.empty(2 * len(xa0), dtype=acj[0].dtype)
        afc[0::2], afc[1::2] = acj[0], acj[1]
        d_afj, d_afc = fr.upload(afj.view(np.uint8)), fr.upload(afc.view(np.uint8))
        n_afc = len(afc)

        def launch_affine(st):
            lib.vvc355_affine_batch(st, bd, ptr(d_afj), n_sb)
            lib.vvc355_bipred_chroma_batch(st, bd, ptr(d_afc), n_afc)

        ctu_of_aff = (ay // CTB) * fr.ncx + ax // CTB

        def check_affine(fc, orc, env):
            dt = np.uint8 if bd == 8 else np.uint16
            idx = np.nonzero(np.isin(ctu_of_aff, env.picks))[0]
            if len(idx) > 4096:
                idx = idx[::len(idx) // 4096]
            bad = 0
            for i in idx:
                j = abi.AffineJob.from_buffer_copy(afj[i].tobytes())
                blk = np.zeros((4, 4), dt)
                off = int(j.dst) - rec_ptrs[0]
                x_, y_ = (off % pitches[0]) // isz, off // pitches[0]
                j.dst, j.dst_stride = blk.ctypes.data, 4 * isz
                j.ref0, j.ref1, j.diff_mv = env.mirror.host_addr(j.ref0), env.mirror.host_addr(j.ref1), env.mirror.host_addr(j.diff_mv)
                j.lmcs_lut = env.mirror.host_addr(j.lmcs_lut) if j.lmcs_lut else 0
                orc.orc_affine_block(bd, ctypes.byref(j))
                bad += not np.array_equal(env.after[rec_ptrs[0]][y_:y_ + 4, x_:x_ + 4], blk)
            cidx = np.nonzero(np.isin(np.repeat((ya0 // CTB) * fr.ncx + xa0 // CTB, 2), env.picks))[0]
            bad += fc.check_bipred(orc, bd, afc, cidx, env.mirror, [None, env.after[rec_ptrs[1]], env.after[rec_ptrs[2]]], rec_ptrs, pitches)
            return len(idx) + len(cidx), bad

        chain.append(Stage("inter_pred_affine_prof", f"affine_kernel<{bd}>", launch_affine, len(xa0) * (256 + 128) * 3 * isz, writes=rec, check=check_affine))

    # ---------------------------------------------------------------- the intra CTUs: a random partition into coding units, flattened into
    # the RECON stage driver's per-CTU command lists (tests/recon_cases.py mirrors what the parse stage leaves per CTU); their
    # transform blocks go through the batched transform stage with store_coeffs (residuals in place), the predictions and
    # the residual adds through the in-order wavefront pass further down
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import recon_cases
    # (every frame object of a run describes the same partition — the generators are seeded with constants — so the host-side generation
    # is done once per process and shared; the per-frame arrays with addresses in them are bound per object below)
    wkey = ("work", fr.width, fr.height, LMCS, ctu_inter.tobytes(), ctu_ciip.tobytes())
    if wkey not in _GENERATED:
        _GENERATED[wkey] = recon_cases.ReconWork(np.random.default_rng(0x5EED0EC0), fr.width, fr.height, 7, 1, 1, intra_ctu=~ctu_inter, split=(0.6, 0.1), ciip_ctu=ctu_ciip,
                                                 lmcs=LMCS, resid_ctu=ctu_dep if LMCS else None)
    work = _GENERATED[wkey]

    # the inter half of the CIIP coding units: plain bi-prediction (CIIP switches DMVR / BDOF off) of every 16x16 (chroma 8x8 or 16x16)
    # tile into packed per-CU blocks that the RECON pass blends with the planar intra prediction
    d_ciip = fr.upload(np.zeros(max(1, work.ciip_len), np.uint8 if bd == 8 else np.uint16), per_frame=False)     # device scratch: inter part of the CIIP units
    cj = []
    for (c, x, y, w, h, off, _k) in work.ciip:
        sh = 1 if c else 0
        wc, hc = w >> sh, h >> sh
        for ty in range(0, hc, 16):
            for tx in range(0, wc, 16):
                cj.append((c, (x >> sh) + tx, (y >> sh) + ty, min(16, wc - tx), min(16, hc - ty), off + ty * wc + tx, wc))
    cj = np.array(cj, np.int64).reshape(-1, 7)
    n_cj = len(cj)
    if n_cj:
        cjobs = batch.job_array(abi.BipredJob, n_cj)
        cc = cj[:, 0]
        cjobs["dst"] = ptr(d_ciip) + cj[:, 5] * isz
        cjobs["dst_stride"] = cj[:, 6] * isz
        for r, key in enumerate(("ref0", "ref1")):
            cjobs[key] = np.array([ref_org(r, c) for c in range(3)], np.int64)[cc]
            cjobs[key + "_stride"] = np.array([ref_pitch(r, c) for c in range(3)], np.int64)[cc]
        cmv = rng.integers(-24 * 16, 24 * 16 + 1, size=(len(work.ciip), 4))
        # every tile of a coding unit shares its motion: index of the CU each tile came from
        cu_of_tile = np.repeat(np.arange(len(work.ciip)), [((w >> (1 if c else 0)) + 15) // 16 * (((h >> (1 if c else 0)) + 15) // 16) for (c, x, y, w, h, off, _k) in work.ciip])
        cjobs["mv"] = cmv[cu_of_tile // 3 * 3]            # the three components of a CU are consecutive entries
        cjobs["x"], cjobs["y"], cjobs["w"], cjobs["h"] = cj[:, 1], cj[:, 2], cj[:, 3], cj[:, 4]
        cjobs["pic_w"], cjobs["pic_h"] = np.array([d[0] for d in fr.dims])[cc], np.array([d[1] for d in fr.dims])[cc]
        cjobs["chroma"], cjobs["hs"], cjobs["vs"] = (cc > 0).astype(np.int64), 1, 1
        cjobs["lmcs_lut"] = np.where(cc == 0, fwd_ptr, 0)          # the inter part of a CIIP block is mapped before the blend (vvc_inter.c:573-574)
        cl, cch = cjobs[cc == 0], cjobs[cc > 0]
        d_cl, d_cch = fr.upload(cl.view(np.uint8)), fr.upload(cch.view(np.uint8))
        n_cl, n_cch = len(cl), len(cch)

        def launch_ciip(st):
            lib.vvc355_bipred_batch(st, bd, ptr(d_cl), n_cl)
            lib.vvc355_bipred_batch(st, bd, ptr(d_cch), n_cch)

        def check_ciip_inter(fc, orc, env):
            dt = np.uint8 if bd == 8 else np.uint16
            got = env.after[ptr(d_ciip)]
            bad = 0
            idx = range(0, n_cj, max(1, n_cj // 400))
            for i in idx:
                j = abi.BipredJob.from_buffer_copy(cjobs[i].tobytes())
                blk = np.zeros((j.h, j.w), dt)
                first, pitch_px = (int(j.dst) - ptr(d_ciip)) // isz, j.dst_stride // isz
                j.dst, j.dst_stride = blk.ctypes.data, j.w * isz
                j.ref0, j.ref1 = env.mirror.host_addr(j.ref0), env.mirror.host_addr(j.ref1)
                j.lmcs_lut = env.mirror.host_addr(j.lmcs_lut) if j.lmcs_lut else 0
                orc.orc_bipred_block(bd, ctypes.byref(j))
                rows = np.stack([got[first + r * pitch_px:first + r * pitch_px + j.w] for r in range(j.h)])
                bad += not np.array_equal(rows, blk)
            return len(idx), bad

        chain.append(Stage("inter_pred_ciip", f"bipred_kernel<{bd}, true>", launch_ciip, int(work.ciip_len) * 3 * isz, writes=[d_ciip], check=check_ciip_inter))

    # ---------------------------------------------------------------- inverse transform + residual add, every sample of the frame
    by_shape = {}              # log2 size -> job arrays of all three planes: one launch per block shape
    windows = []               # (first coefficient, block size, nzw[], nzh[]) of every group of blocks laid out back to back
    ctu_by_shape = {}
    pos_by_shape = {}          # (x0, y0, component) of every job, in by_shape's order
    coeff_off = 0
    chroma_first = None        # with LMCS: first coefficient (int32 index) of the chroma blocks, whose residuals stay in the buffer for the scaling stage
    cres = []                  # (component, x0, y0, size, first coefficient byte offset) per group of chroma blocks
    for c, (w, h) in enumerate(fr.dims):
        if c == 1 and LMCS:
            chroma_first = coeff_off // 4
        if c == 0:
            # per 128x128 CTU: one 64x64, four 32x32, sixteen 16x16, sixty-four 8x8 (one size per quadrant)
            parts = [(0, 0, 64), (64, 0, 32), (0, 64, 16), (64, 64, 8)]
        else:
            parts = [(0, 0, 32), (32, 0, 16), (0, 32, 8), (32, 32, 4)]
        cs = CTB if c == 0 else CTB // 2
        cx0, cy0, _, _ = batch.ctb_grid(w // cs * cs, h // cs * cs, cs)
        for (qx, qy, n) in parts:
            q = cs // 2
            ox, oy = np.meshgrid(np.arange(0, q, n), np.arange(0, q, n))
            x0 = (cx0[:, None] + qx + ox.ravel()[None, :]).ravel()
            y0 = (cy0[:, None] + qy + oy.ravel()[None, :]).ravel()
            kc = (y0 // cs) * fr.ncx + (x0 // cs)
            keep = ctu_inter[kc] & ~ctu_ciip[kc]       # the intra CTUs have their own transform blocks (below); the CIIP coding units carry no residual here
            if c and LMCS:
                keep &= ~ctu_dep[kc]                   # their chroma residuals are scaled and added by the in-order pass (RESID commands of `work`)
            x0, y0 = x0[keep], y0[keep]
            if not len(x0):
                continue
            j = batch.job_array(abi.ItxJob, len(x0))
            lg = int(np.log2(n))
            j["coeffs"] = coeff_off + np.arange(len(x0), dtype=np.int64) * (n * n * 4)
            coeff_off += len(x0) * n * n * 4
            j["dst"] = ptr(rec[c]) + y0 * fr.pitch(rec[c]) + x0 * isz
            j["dst_stride"] = fr.pitch(rec[c])
            j["log2_w"] = j["log2_h"] = lg
            dxt_ok = 4 <= n <= 32
            use_dxt = (rng.random(len(x0)) < 0.3) & dxt_ok
            j["trh"] = np.where(use_dxt, rng.integers(1, 3, size=len(x0)), 0)
            j["trv"] = np.where(use_dxt, rng.integers(1, 3, size=len(x0)), 0)
            lim_h = np.where(j["trh"] == 0, min(32, n), min(16, n))
            lim_v = np.where(j["trv"] == 0, min(32, n), min(16, n))
            j["nzw"] = 1 + (rng.random(len(x0)) * lim_h).astype(np.int64)
            j["nzh"] = 1 + (rng.random(len(x0)) * lim_v).astype(np.int64)
            j["range"], j["bd"], j["store_coeffs"] = 15, bd, 0
            if c and LMCS:
                # chroma residual scaling: the transform leaves the residual in the buffer, the scaling stage adds it
                j["dst"], j["store_coeffs"] = 0, 1
                cres.append((c, x0.copy(), y0.copy(), n, j["coeffs"].copy()))
            by_shape.setdefault(lg, []).append(j)
            pos_by_shape.setdefault(lg, []).append(np.stack([x0, y0, np.full_like(x0, c)], axis=1))
            ctu_by_shape.setdefault(lg, []).append((y0 // cs) * fr.ncx + (x0 // cs))
            windows.append((int(j["coeffs"][0]) // 4, n, j["nzw"].copy(), j["nzh"].copy()))
    if by_shape:         # (no transform blocks outside the intra CTUs in an all-intra picture)
        tj, itx_launches = [], []  # (first job, count, log2 size)
        for lg in sorted(by_shape, reverse=True):
            j = np.concatenate(by_shape[lg])
            itx_launches.append((sum(len(t) for t in tj), len(j), lg))
            tj.append(j)
        if fr.noise:
            coeffs = torch.randint(-(1 << 8), 1 << 8, (coeff_off // 4,), device="cuda", generator=fr.gen, dtype=torch.int32)
        else:
            # sparse levels with a Laplacian magnitude distribution (about half of them zero): the residual of a coded picture
            mag = (-torch.log(torch.rand(coeff_off // 4, device="cuda", generator=fr.gen).clamp_min(1e-9)) * 1.1).floor()
            sign = torch.randint(0, 2, (coeff_off // 4,), device="cuda", generator=fr.gen, dtype=torch.int32) * 2 - 1
            coeffs = mag.to(torch.int32) * sign
            del mag, sign
        # levels exist only inside each block's [0, nzw) x [0, nzh) window (the contract of the nz arguments, vvcdsp.h:118)
        for (first, n, nzw, nzh) in windows:
            v = coeffs[first:first + len(nzw) * n * n].view(len(nzw), n, n)
            ar = torch.arange(n, device="cuda")
            v *= ((ar[None, None, :] < torch.from_numpy(nzw.astype(np.int64)).cuda()[:, None, None]) &
                  (ar[None, :, None] < torch.from_numpy(nzh.astype(np.int64)).cuda()[:, None, None])).to(torch.int32)
        fr.keep.append(coeffs)
        itx_all = np.concatenate(tj)
        ctu_of_itx = np.concatenate([np.concatenate(ctu_by_shape[lg]) for lg in sorted(by_shape, reverse=True)])
        itx_all["coeffs"] += coeffs.data_ptr()
        n_itx = len(itx_all)
        n_samples = coeff_off // 4
        jsz = itx_all.dtype.itemsize

        # the scaling process (dequant, vvc_intra.c:277-417) is fused into the transform's load stage: flat scaling matrix,
        # qp 22..37, dependent quantisation on for half of the blocks
        itx_all["dq_flags"] = 1 | (rng.integers(0, 2, size=n_itx) << 1)
        itx_all["dq_qp"] = rng.integers(22, 38, size=n_itx)
        itx_all["log2_matrix_size"], itx_all["dc"] = 1, -1
        # The 48-byte jobs above are the host's expectation only.  What the device runs is written by vvc355_itx_frame_build from one 16-byte
        # record per transform block (what the parser leaves in a TransformBlock); the job array is device scratch.
        pos_all = np.concatenate([np.concatenate(pos_by_shape[lg]) for lg in sorted(by_shape, reverse=True)])
        tus = np.zeros(n_itx, np.dtype(abi.ItxTu, align=True))
        tus["coeff_off"] = (itx_all["coeffs"] - coeffs.data_ptr()) // 4
        tus["x0"], tus["y0"], tus["c_idx"] = pos_all[:, 0], pos_all[:, 1], pos_all[:, 2]
        for k_ in ("log2_w", "log2_h", "nzw", "nzh"):
            tus[k_] = itx_all[k_]
        tus["qp"] = itx_all["dq_qp"]
        # (with LMCS the chroma blocks' residuals stay in the arena and are scaled by the next stage: bit 6, every 64x64 unit's neighbours exist
        # except at the picture's left / upper edge — one slice, one tile)
        cx_l, cy_l = pos_all[:, 0] << (pos_all[:, 2] > 0), pos_all[:, 1] << (pos_all[:, 2] > 0)
        tus["flags"] = ((itx_all["dq_flags"] & 3) | (itx_all["store_coeffs"] << 2) |
                        np.where(itx_all["store_coeffs"] != 0, 64 | (((cx_l & ~63) > 0) << 4) | (((cy_l & ~63) > 0) << 5), 0))
        tus["tr"] = itx_all["trh"] | (itx_all["trv"] << 4)
        itx_all["c_idx"] = pos_all[:, 2]
        d_tus = fr.upload(tus.view(np.uint8))
        d_itx = fr.upload(itx_all.view(np.uint8), per_frame=False)
        itf = abi.ItxFrame()
        itf.tus, itf.jobs, itf.coeffs, itf.n_tus = ptr(d_tus), ptr(d_itx), coeffs.data_ptr(), n_itx
        for c in range(3):
            itf.plane[c], itf.stride[c] = ptr(rec[c]), fr.pitch(rec[c])
        itf.range, itf.bd, itf.pixel_shift = 15, bd, int(isz == 2)
        itf.width, itf.height, itf.hs, itf.vs, itf.size_y = fr.width, fr.height, 1, 1, 64
        d_rjall = fr.upload(np.zeros(n_itx * ctypes.sizeof(abi.LmcsResidJob), np.uint8), per_frame=False) if (LMCS and chroma_first is not None) else None
        itf.resid_jobs = ptr(d_rjall) if d_rjall is not None else 0
        # every 64x64 unit's chroma residual scale, one table per picture (vvc355_lmcs_vpdu_scale_pass); the residual jobs point into it
        vux, vuy = (fr.width + 63) // 64, (fr.height + 63) // 64
        d_vscale = fr.upload(np.zeros(vux * vuy, np.int16), per_frame=False) if d_rjall is not None else None
        itf.scale_table = ptr(d_vscale) if d_vscale is not None else 0
        d_itf = fr.upload(np.frombuffer(bytes(itf), np.uint8))
        fr.keep.append(itf)
        d_itx.zero_()
        lib.vvc355_itx_frame_build(None, ptr(d_itf), ctypes.addressof(itf))

        def check_itx_build(fc, orc, env):
            got = env.after[ptr(d_itx)].view(itx_all.dtype)
            return n_itx, int(sum(int((got[name] != itx_all[name]).sum()) for name in itx_all.dtype.names if not name.startswith("pad")))

        chain.append(Stage("itx_job_build", "itx_build_kernel", lambda st: lib.vvc355_itx_frame_build(st, ptr(d_itf), ctypes.addressof(itf)),
                           n_itx * (16 + 48), writes=[d_itx] + ([d_rjall] if d_rjall is not None else []), check=check_itx_build))

        scaled = LMCS and chroma_first is not None and chroma_first < n_samples
        if scaled:
            coeffs_c = coeffs[chroma_first:]                 # the chroma blocks' part: levels in, residuals out (restored from the pristine copy every step)
            coeffs_c0 = coeffs_c.clone()
            fr.keep += [coeffs_c, coeffs_c0]

        def launch_itx(st):
            if scaled:
                lib.vvc355_copy_async(st, coeffs_c.data_ptr(), coeffs_c0.data_ptr(), coeffs_c.numel() * 4)
            for (first, count, lg) in itx_launches:
                lib.vvc355_itx_shape_batch(st, bd, ptr(d_itx) + first * jsz, count, lg, lg)

        def check_itx(fc, orc, env):
            levels = env.snap(coeffs)
            if scaled:
                levels[chroma_first:] = env.snap(coeffs_c0)
            env.mirror.add(coeffs.data_ptr(), levels)
            idx = np.nonzero(np.isin(ctu_of_itx, env.picks))[0]
            bad = fc.check_itx(orc, bd, itx_all, idx, env.mirror, [env.before[p_] for p_ in rec_ptrs], [env.after[p_] for p_ in rec_ptrs], rec_ptrs, pitches,
                               coeffs_after=(coeffs_c.data_ptr(), env.after[coeffs_c.data_ptr()]) if scaled else None)
            return len(idx), bad

        chain.append(Stage("dequant_itx_add_residual", f"itx_shape_kernel<{bd}, *>", launch_itx, n_samples * (4 + 2 * isz), writes=rec + ([coeffs_c] if scaled else []), check=check_itx))

        if scaled:
            # ---------------------------------------------------------------- chroma residual scaling of the inter CTUs that do not depend on the in-order
            # pass: per block the 64x64 unit's scale from the reconstructed luma (prediction + residual: final for these CTUs), then
            # lmcs_scale_chroma + add_residual
            rjs = []
            for (c, x0c, y0c, n, cofs) in cres:
                rj = batch.job_array(abi.LmcsResidJob, len(x0c))
                rj["dst"] = ptr(rec[c]) + y0c * fr.pitch(rec[c]) + x0c * isz
                rj["dst_stride"], rj["luma"], rj["luma_stride"] = fr.pitch(rec[c]), ptr(rec[0]), fr.pitch(rec[0])
                rj["resid"] = cofs + coeffs.data_ptr()
                rj["w"] = rj["h"] = n
                rj["x_vpdu"], rj["y_vpdu"] = (2 * x0c) & ~63, (2 * y0c) & ~63
                rj["pic_w"], rj["pic_h"], rj["size_y"] = fr.width, fr.height, 64
                rj["avail_l"], rj["avail_t"] = rj["x_vpdu"] > 0, rj["y_vpdu"] > 0          # one slice, one tile: every neighbour inside the picture exists
                rj["joint"] = 8
                rjs.append(rj)
            rj_all = np.concatenate(rjs)
            ctu_of_rj = np.concatenate([((2 * y0c) // CTB) * fr.ncx + (2 * x0c) // CTB for (_c, x0c, y0c, _n, _o) in cres])
            # (rj_all is the host's expectation for the check; the device runs the job array the transform-block builder wrote: one slot per
            # transform block, empty for the blocks that are not scaled here)

            def check_lmcs_resid(fc, orc, env):
                orc.orc_lmcs_chroma_resid_block.argtypes = [ctypes.c_int, ctypes.POINTER(abi.LmcsResidJob), ctypes.POINTER(abi.LmcsModel)]
                orc.orc_lmcs_chroma_resid_block.restype = None
                idx = np.nonzero(np.isin(ctu_of_rj, env.picks))[0]
                luma_h, res_h = env.snap(rec[0]), env.snap(coeffs_c)
                dt = np.uint8 if bd == 8 else np.uint16
                bad = 0
                for i in idx:
                    j = abi.LmcsResidJob.from_buffer_copy(rj_all[i].tobytes())
                    c = next(k for k in (1, 2) if rec_ptrs[k] <= j.dst < rec_ptrs[k] + env.after[rec_ptrs[k]].nbytes)
                    off = int(j.dst) - rec_ptrs[c]
                    x_, y_ = (off % pitches[c]) // isz, off // pitches[c]
                    blk = np.ascontiguousarray(env.before[rec_ptrs[c]][y_:y_ + j.h, x_:x_ + j.w]).astype(dt)
                    r = np.ascontiguousarray(res_h[(int(j.resid) - coeffs_c.data_ptr()) // 4:][:j.w * j.h])
                    j.dst, j.dst_stride, j.resid, j.luma = blk.ctypes.data, j.w * isz, r.ctypes.data, luma_h.ctypes.data
                    orc.orc_lmcs_chroma_resid_block(bd, ctypes.byref(j), ctypes.byref(lmcs_model))
                    bad += not np.array_equal(env.after[rec_ptrs[c]][y_:y_ + j.h, x_:x_ + j.w], blk)
                return len(idx), bad

            lsf = abi.LmcsScaleFrame()
            d_vtabs = [fr.upload(work.slice_idx), fr.upload(work.col_bd), fr.upload(work.row_bd)]
            lsf.luma, lsf.scale, lsf.model, lsf.luma_stride = ptr(rec[0]), ptr(d_vscale), ptr(d_model), fr.pitch(rec[0])
            lsf.slice_idx, lsf.ctb_to_col_bd, lsf.ctb_to_row_bd = (ptr(t) for t in d_vtabs)
            lsf.width, lsf.height, lsf.ctb_width, lsf.ctb_log2, lsf.size_y = fr.width, fr.height, fr.ncx, int(np.log2(CTB)), 64
            d_lsf = fr.upload(np.frombuffer(bytes(lsf), np.uint8))
            fr.keep.append(lsf)

            def check_vpdu_scale(fc, orc, env):
                # the whole table against the oracle's on the same luma plane (units whose neighbours the in-order pass has yet to write
                # hold a value nobody reads; they are compared all the same)
                orc.orc_lmcs_vpdu_scale_pass.argtypes = [ctypes.c_int, ctypes.POINTER(abi.LmcsScaleFrame)]
                orc.orc_lmcs_vpdu_scale_pass.restype = None
                luma_h = env.snap(rec[0])
                want = np.zeros(vux * vuy, np.int16)
                tabs = [np.ascontiguousarray(t) for t in (work.slice_idx, work.col_bd, work.row_bd)]
                hf = abi.LmcsScaleFrame.from_buffer_copy(bytes(lsf))
                hf.luma, hf.scale, hf.model = luma_h.ctypes.data, want.ctypes.data, ctypes.addressof(lmcs_model)
                hf.slice_idx, hf.ctb_to_col_bd, hf.ctb_to_row_bd = (t.ctypes.data for t in tabs)
                orc.orc_lmcs_vpdu_scale_pass(bd, ctypes.byref(hf))
                got = env.after[ptr(d_vscale)].view(np.int16)[:vux * vuy]
                return vux * vuy, int((got != want).sum())

            chain.append(Stage("lmcs_vpdu_scale_table", f"lmcs_vpdu_scale_kernel<{bd}>",
                               lambda st: lib.vvc355_lmcs_vpdu_scale_pass(st, bd, ptr(d_lsf), ctypes.addressof(lsf)),
                               vux * vuy * (128 * isz + 2), writes=[d_vscale], check=check_vpdu_scale))
            n_cs = int(sum(len(x0c) * n * n for (_c, x0c, _y, n, _o) in cres))
            chain.append(Stage("lmcs_chroma_residual_scale", f"lmcs_chroma_resid_kernel<{bd}>",
                               lambda st: lib.vvc355_lmcs_chroma_resid_batch(st, bd, ptr(d_rjall), n_itx, ptr(d_model)),
                               n_cs * (4 + 2 * isz), writes=[rec[1], rec[2]], check=check_lmcs_resid))

    # ---------------------------------------------------------------- transform blocks of the intra CTUs: scaling process (+ LFNST on a
    # fifth of the luma blocks) + inverse transform with the transform types derived on the device (implicit MTS), residuals left
    # in place for the RECON pass.  Levels are restored from a pristine copy at the start of every step (the stage works in place).
    itb = np.array(work.tbs, dtype=np.int64).reshape(-1, 6)        # c_idx, x0, y0, w, h, element offset
    n_tb = len(itb)
    rng_t = np.random.default_rng(0x5EED0EC1)
    lev_mag = (-torch.log(torch.rand(max(1, work.resid_len), device="cuda", generator=fr.gen).clamp_min(1e-9)) * 1.1).floor().to(torch.int32)
    levels0 = lev_mag * (torch.randint(0, 2, (max(1, work.resid_len),), device="cuda", generator=fr.gen, dtype=torch.int32) * 2 - 1)
    res = torch.zeros_like(levels0)
    fr.keep += [levels0, res]
    tw, th, tc = itb[:, 3], itb[:, 4], itb[:, 0]
    tlw, tlh = np.log2(tw).astype(np.int64), np.log2(th).astype(np.int64)
    use_lfnst = (tc == 0) & (tw >= 4) & (th >= 4) & (rng_t.random(n_tb) < 0.2)
    itj = batch.job_array(abi.ItxJob, n_tb)
    itj["coeffs"] = ptr(res) + itb[:, 5] * 4
    itj["log2_w"], itj["log2_h"], itj["range"], itj["bd"], itj["store_coeffs"], itj["c_idx"] = tlw, tlh, 15, bd, 1, tc
    inzw = 1 + (rng_t.random(n_tb) * np.minimum(tw, 16)).astype(np.int64)
    inzh = 1 + (rng_t.random(n_tb) * np.minimum(th, 16)).astype(np.int64)
    lf_n = np.where((tw >= 8) & (th >= 8), 8, 4)
    itj["nzw"], itj["nzh"] = np.where(use_lfnst, lf_n, inzw), np.where(use_lfnst, lf_n, inzh)
    itj["dq_flags"] = np.where(use_lfnst, 0, 1 | (rng_t.integers(0, 2, size=n_tb) << 1))
    itj["dq_qp"], itj["log2_matrix_size"], itj["dc"] = rng_t.integers(22, 38, size=n_tb), 1, -1
    itj["mts_flags"], itj["tu_flags"] = abi.ITX_DERIVE_TYPE, abi.TU_MTS_ENABLED | abi.TU_INTRA
    itj["lfnst_idx"] = np.where(use_lfnst, rng_t.integers(1, 3, size=n_tb), 0)
    # the levels of a block live inside its scan window (LFNST blocks: the ifirst 16 positions of the 4x4 diagonal scan)
    win_w, win_h = np.where(use_lfnst, np.minimum(tw, 4), inzw), np.where(use_lfnst, np.minimum(th, 4), inzh)
    lfj = batch.job_array(abi.LfnstJob, int(use_lfnst.sum()))
    lfi = np.nonzero(use_lfnst)[0]
    lfj["coeffs"], lfj["log2_w"], lfj["log2_h"] = itj["coeffs"][lfi], tlw[lfi], tlh[lfi]
    lfj["max_x"], lfj["max_y"], lfj["qp"], lfj["dequant"], lfj["dep_quant"] = win_w[lfi] - 1, win_h[lfi] - 1, itj["dq_qp"][lfi], 1, rng_t.integers(0, 2, size=len(lfi))
    lfj["bit_depth"], lfj["range"], lfj["log2_matrix_size"], lfj["dc"] = bd, 15, 1, -1
    lfj["pred_mode_intra"], lfj["lfnst_idx"] = rng_t.integers(-14, 81, size=len(lfi)), itj["lfnst_idx"][lfi]
    # zero the levels outside the windows (one pass over the pristine copy, host-built imask)
    imask = np.zeros(max(1, work.resid_len), np.int32)
    for k in range(n_tb):
        w_, o_ = int(tw[k]), int(itb[k, 5])
        m2 = imask[o_:o_ + int(tw[k] * th[k])].reshape(int(th[k]), w_)
        m2[:int(win_h[k]), :int(win_w[k])] = 1
    levels0 *= torch.from_numpy(imask).cuda()
    area_class = np.select([tlw + tlh <= 4, tlw + tlh <= 6, tlw + tlh <= 8, tlw + tlh <= 10], [4, 6, 8, 10], 12)
    tb_order = np.argsort(area_class, kind="stable")
    itj_sorted = itj[tb_order]
    d_tj, d_lj = fr.upload(itj_sorted.view(np.uint8)), fr.upload(lfj.view(np.uint8) if len(lfj) else np.zeros(32, np.uint8))
    tb_launches, ifirst = [], 0
    for cls in (4, 6, 8, 10, 12):
        cnt = int((area_class == cls).sum())
        if cnt:
            tb_launches.append((ifirst, cnt, cls))
        ifirst += cnt
    n_lj, tjsz = len(lfj), itj.dtype.itemsize

    def launch_intra_tb(st):
        lib.vvc355_copy_async(st, ptr(res), ptr(levels0), res.numel() * 4)
        if n_lj:
            lib.vvc355_lfnst_batch(st, ptr(d_lj), n_lj)
        for (first_, cnt_, cls_) in tb_launches:
            lib.vvc355_itx_batch(st, bd, ptr(d_tj) + first_ * tjsz, cnt_, cls_)

    tb_ctu = (itb[:, 2] // CTB) * fr.ncx + itb[:, 1] // CTB

    def check_intra_tb(fc, orc, env):
        lev = env.snap(levels0)
        got = env.after[ptr(res)]
        pick = np.nonzero(np.isin(tb_ctu, env.picks) | (np.arange(n_tb) % 97 == 0))[0]
        lf_of = {int(i): k for k, i in enumerate(lfi)}
        bad = 0
        for k in pick:
            w_, h_, o_ = int(tw[k]), int(th[k]), int(itb[k, 5])
            co = lev[o_:o_ + w_ * h_].copy()
            j = itj[k]
            if int(k) in lf_of:
                l = lfj[lf_of[int(k)]]
                orc.orc_dequant(co.ctypes.data, int(tlw[k]), int(tlh[k]), 0, 0, int(l["max_x"]), int(l["max_y"]), int(l["qp"]), 0, int(l["dep_quant"]), bd, 15, None, 1, -1)
                orc.orc_ilfnst_transform(co.ctypes.data, w_, h_, int(l["pred_mode_intra"]), int(l["lfnst_idx"]), 15)
            else:
                orc.orc_dequant(co.ctypes.data, int(tlw[k]), int(tlh[k]), 0, 0, int(j["nzw"]) - 1, int(j["nzh"]) - 1, int(j["dq_qp"]), 0, (int(j["dq_flags"]) >> 1) & 1, bd, 15, None, 1, -1)
            t = orc.orc_derive_transform_type(int(j["tu_flags"]), int(j["mts_idx"]), int(j["lfnst_idx"]), int(j["c_idx"]), w_, h_)
            orc.orc_itx(t & 15, t >> 4, int(tlw[k]), int(tlh[k]), co.ctypes.data, int(j["nzw"]), int(j["nzh"]), 15, bd)
            bad += not np.array_equal(co, got[o_:o_ + w_ * h_])
        return len(pick), bad

    if n_tb:
        chain.append(Stage("intra_tb_dequant_lfnst_itx", f"itx_kernel<{bd}, *> + lfnst_batch_kernel", launch_intra_tb, int(work.resid_len) * (4 + 4),
                           writes=[res], check=check_intra_tb))

    # ---------------------------------------------------------------- RECON: the intra CTUs' coding units in decoding order (prediction from
    # what earlier blocks wrote, then the residual), CTUs released in wavefront order
    if len(work.order):
        cmds_dev = work.bind(ptr(res), ptr(d_ciip), isz)
        # ticket order: longest remaining dependency chain first (host helper of the C ABI; the oracle walks in raster = decoding order)
        ticket_order = np.zeros(fr.n_ctus, np.int32)
        n_tk = lib.vvc355_recon_order(np.ascontiguousarray(work.ctus).ctypes.data, fr.ncx, fr.ncy, ticket_order.ctypes.data)
        assert n_tk == len(work.order)
        ticket_order = ticket_order[:n_tk].copy() if not os.environ.get("VVC355_RECON_RASTER") else work.order
        d_cmds, d_ctus, d_order = fr.upload(cmds_dev.view(np.uint8)), fr.upload(work.ctus.view(np.uint8)), fr.upload(ticket_order)
        d_rstate = fr.upload(np.zeros(lib.vvc355_recon_state_bytes(fr.n_ctus), np.uint8), per_frame=False)
        d_rslice, d_rcol, d_rrow = fr.upload(work.slice_idx), fr.upload(work.col_bd), fr.upload(work.row_bd)
        rf = work.frame(rec_ptrs, pitches, ptr(d_cmds), ptr(d_ctus), ptr(d_order), ptr(d_rstate), ptr(d_rslice), ptr(d_rcol), ptr(d_rrow),
                        lmcs_ptr=ptr(d_model) if LMCS else 0)
        d_rf = fr.upload(np.frombuffer(bytes(rf), np.uint8))
        fr.keep.append(rf)
        RECON_FRAMES.append(rf)
        n_pred = int((work.cmds["kind"] == abi.RECON_PRED).sum() + (work.cmds["kind"] == abi.RECON_CCLM).sum())
        intra_px = int(((work.cmds["kind"] == abi.RECON_MARK) & (work.cmds["c_idx"] == 0) * 1).astype(bool).sum())    # noqa: F841

        def check_recon(fc, orc, env):
            # whole picture: the oracle walks the same command lists in decoding order on host copies of the planes
            orc.orc_recon_frame_pass.argtypes = [ctypes.c_int, ctypes.POINTER(abi.ReconFrame)]
            orc.orc_recon_frame_pass.restype = None
            work_p = [env.before[p_].copy() for p_ in rec_ptrs]
            for p_, w_ in zip(rec_ptrs, work_p):
                env.mirror.add(p_, w_)
            env.mirror.add(ptr(res), env.snap(res))
            env.mirror.add(ptr(d_ciip), env.snap(d_ciip))
            hc = work.bind(env.mirror.host_addr(ptr(res)), env.mirror.host_addr(ptr(d_ciip)), isz)
            env.mirror.add(ptr(d_cmds), hc.view(np.uint8))
            f = fc.translate(rf, env.mirror, ("plane", "cmds", "ctus", "order", "slice_idx", "ctb_to_col_bd", "ctb_to_row_bd") + (("lmcs_model",) if LMCS else ()))
            f.state = 0
            raster_order = np.ascontiguousarray(work.order)
            f.order = raster_order.ctypes.data
            orc.orc_recon_frame_pass(bd, ctypes.byref(f))
            bad = sum(int(not np.array_equal(w_, env.after[p_])) for p_, w_ in zip(rec_ptrs, work_p))
            env.stats["recon_commands"] = int(len(work.cmds))
            env.stats["recon_intra_ctus"] = int(len(work.order))
            return len(work.order), bad

        intra_samples = int(sum(int(c[3]) * int(c[4]) for c in work.cmds[work.cmds["kind"] == abi.RECON_MARK] if c[11] == 0)) * 3 // 2
        # what bounds this stage is not bandwidth but the longest chain of dependent CTUs (left, upper-left, upper, upper-right
        # neighbours that have commands), each walked by one wave per channel type: its length in CTUs and in commands
        ncx_ = fr.ncx if hasattr(fr, "ncx") else (fr.width + CTB - 1) // CTB
        ncmd = work.ctus["n_cmd"].astype(np.int64)
        depth_c, depth_k = np.zeros(len(ncmd), np.int64), np.zeros(len(ncmd), np.int64)
        cflags = work.ctus["flags"].astype(np.int64)
        for rs in work.order:
            rs = int(rs)
            rx_, best_c, best_k = rs % ncx_, 0, 0
            if cflags[rs] & abi.RECON_CTU_LIGHT:      # waits for the luma of the flagged neighbours only
                deps_ = ((rs - 1) if (rx_ and cflags[rs] & abi.RECON_CTU_LUMA_LEFT) else -1, (rs - ncx_) if cflags[rs] & abi.RECON_CTU_LUMA_UP else -1)
            else:
                deps_ = ((rs - 1) if rx_ else -1, (rs - ncx_ - 1) if rx_ else -1, rs - ncx_, (rs - ncx_ + 1) if rx_ + 1 < ncx_ else -1)
            for d_ in deps_:
                if d_ >= 0 and ncmd[d_]:
                    best_c, best_k = max(best_c, int(depth_c[d_])), max(best_k, int(depth_k[d_]))
            light_ = bool(cflags[rs] & abi.RECON_CTU_LIGHT)
            depth_c[rs], depth_k[rs] = best_c + (0 if light_ else 1), best_k + (int(ncmd[rs]) + 3) // 4 if light_ else best_k + int(ncmd[rs])
        recon_chain = {"ctus_with_commands": int(len(work.order)), "commands": int(len(work.cmds)),
                       "longest_dependency_chain_ctus": int(depth_c.max()), "longest_dependency_chain_commands": int(depth_k.max()),
                       "chain_note": "chains follow what the pass waits for (LIGHT CTUs: flagged luma neighbours only, weighted a quarter); tickets in vvc355_recon_order() order"}
        chain.append(Stage("intra_recon_wavefront", f"recon_wavefront_kernel<{bd}>", lambda st: lib.vvc355_recon_frame_pass(st, bd, ptr(d_rf), ctypes.addressof(rf)),
                           intra_samples * isz, writes=rec, check=check_recon))
        chain[-1].extra = {"bound": "dependency chain (one wave per CTU and channel type), not bandwidth", **recon_chain}

    # ---------------------------------------------------------------- LMCS inverse luma mapping
    lut = fr.upload(np.sort(rng.integers(0, 1 << bd, size=1 << bd)).astype(np.uint8 if bd == 8 else np.uint16))
    x0, y0, cw, ch = batch.ctb_grid(fr.width, fr.height, CTB)
    lj = batch.job_array(abi.BlendJob, len(x0))
    lj["dst"] = ptr(rec[0]) + y0 * fr.pitch(rec[0]) + x0 * isz
    lj["dst_stride"], lj["src0"], lj["w"], lj["h"] = fr.pitch(rec[0]), ptr(lut), cw, ch
    d_lmcs = fr.upload(lj.view(np.uint8))
    n_lmcs = len(lj)
    def check_lmcs(fc, orc, env, x0=x0, y0=y0, cw=cw, ch=ch):
        rects = [(int(x0[i]), int(y0[i]), int(cw[i]), int(ch[i])) for i in env.picks]
        return len(rects), fc.check_lmcs(orc, bd, rects, fr.host[ptr(lut)], env.before[rec_ptrs[0]], env.after[rec_ptrs[0]])

    chain.append(Stage("lmcs_inverse_luma", f"lmcs_kernel<{bd}>", lambda st: lib.vvc355_lmcs_batch(st, bd, ptr(d_lmcs), n_lmcs, CTB, CTB),
                       fr.width * fr.height * isz * 2, writes=[rec[0]], check=check_lmcs))

    # ---------------------------------------------------------------- deblocking: every vertical edge, then every horizontal edge
    def deblock_jobs(direction):
        js = []
        for c, (w, h) in enumerate(fr.dims):
            grid = 8
            if direction == 1:      # vertical edges at x = 8, 16, ...; 8 rows per job
                ex, ey = np.meshgrid(np.arange(grid, w, grid), np.arange(0, h - 7, 8))
            else:                   # horizontal edges at y = 8, 16, ...; 8 columns per job
                ex, ey = np.meshgrid(np.arange(0, w - 7, 8), np.arange(grid, h, grid))
            ex, ey = ex.ravel(), ey.ravel()
            j = batch.job_array(abi.DeblockJob, len(ex))
            j["pix"] = ptr(rec[c]) + ey * fr.pitch(rec[c]) + ex * isz
            j["stride"], j["dir"], j["chroma"] = fr.pitch(rec[c]), direction, int(c > 0)
            qp = rng.integers(22, 43, size=(len(ex), 4))
            bs_on = rng.random((len(ex), 4)) < 0.6                      # boundary strength > 0 on 60 % of the segments
            j["tc"] = np.where(bs_on, np.array(TC_TABLE)[qp + 2], 0)
            j["beta"] = np.array(BETA_TABLE)[qp]
            j["max_len_p"] = rng.choice([1, 3] if c else [1, 2, 3], size=(len(ex), 4))
            j["max_len_q"] = rng.choice([1, 3] if c else [1, 2, 3], size=(len(ex), 4))
            j["flag"] = 0
            js.append(j)
        return np.concatenate(js)

    frame_bytes = sum(w * h for (w, h) in fr.dims) * isz

    # ---------------------------------------------------------------- boundary strengths: the decoder's side tables -> bS / max filter lengths
    bs_dev = None
    if not DEBLOCK_JOBS:
        # a self-consistent random partition (coding blocks 8..64, transform-unit strips, 80 % inter with sub-block blocks, coded
        # flags) of a quarter-width picture, repeated four times across: about 70 % of the 4-sample segments on the 8-sample
        # grid end up filtered
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests"))
        import bs_cases
        reps = next(r for r in (4, 3, 2, 1) if fr.width % (r * CTB) == 0 or r == 1)
        bkey = ("bs", fr.width // reps, fr.height)
        if bkey not in _GENERATED:
            bt0 = bs_cases.BsTables(np.random.default_rng(0x5EED0B5), fr.width // reps, fr.height, 7, split=(0.95, 0.45), cbf_p=0.4)
            _GENERATED[bkey] = (bt0, bt0.records())
        import copy
        bt, bt_records = copy.copy(_GENERATED[bkey][0]), _GENERATED[bkey][1]
        bs_dev = {}
        for name in bt.IN + bt.OUT:
            a = getattr(bt, name)
            if name in ("slice_idx",):
                a = np.zeros(fr.n_ctus, np.int16)
            elif name in ("col_bd",):
                a = np.zeros(fr.ncx + 1, np.int16)
            elif a.ndim == 2 and a.shape == (bt.th, bt.tw):
                a = np.tile(a, (1, reps))
                if name in ("tbx0", "tbx1", "cbx"):
                    a = a + (np.arange(reps * bt.tw) // bt.tw * (fr.width // reps)).astype(a.dtype)[None, :]
            a = np.ascontiguousarray(a)
            # the per-unit tables are written on the device from records (below), the ten output tables by the bS pass itself: only the
            # slice / tile maps and the POC lists travel as they are
            bs_dev[name] = fr.upload(a.view(np.uint8) if a.dtype.kind == "V" else a, per_frame=name not in bt.FILLED + bt.OUT)
        # what the parser knows per coding unit / transform unit / rectangle of equal motion, repeated across like the tables
        recs = []
        for r_ in bt_records:
            tiled = np.tile(r_, reps)
            tiled["x0"] += (np.arange(len(tiled)) // len(r_) * (fr.width // reps)).astype(np.int16)
            recs.append(tiled)
        grouped = [bt.group_per_ctu(r_, 7, fr.ncx, fr.n_ctus) for r_ in recs]           # per CTU, raster order: how a parser produces them
        recs = [g_[0] for g_ in grouped]
        d_recs = [fr.upload(r_.view(np.uint8)) for r_ in recs]
        d_firsts = [fr.upload(g_[1]) for g_ in grouped]
        bt.width, bt.tw, bt.cw = fr.width, bt.tw * reps, fr.ncx
        tfill = bt.fill_frame(ptr(d_recs[0]), ptr(d_recs[1]), ptr(d_recs[2]), tuple(len(r_) for r_ in recs), lambda name: ptr(bs_dev[name]),
                              tuple(ptr(d_) for d_ in d_firsts))
        d_tfill = fr.upload(np.frombuffer(bytes(tfill), np.uint8))
        fr.keep.append(tfill)
        filled = [bs_dev[name] for name in bt.FILLED]
        for t_ in filled:
            t_.zero_()                                     # the device writes them: nothing of the host's copy is left in HBM
        lib.vvc355_tab_fill_pass(None, ptr(d_tfill), ctypes.addressof(tfill))

        def check_fill(fc, orc, env):
            bad = sum(int(not np.array_equal(fr.derived[ptr(t_)].reshape(-1).view(np.uint8), env.after[ptr(t_)].reshape(-1).view(np.uint8))) for t_ in filled)
            return sum(len(r_) for r_ in recs), bad

        chain.append(Stage("side_tables_fill", "tabfill_kernel", lambda st: lib.vvc355_tab_fill_pass(st, ptr(d_tfill), ctypes.addressof(tfill)),
                           sum(fr.derived[ptr(t_)].nbytes for t_ in filled), writes=filled, check=check_fill))
        bsf = bt.frame(lambda name: ptr(bs_dev[name]))
        d_bsf = fr.upload(np.frombuffer(bytes(bsf), np.uint8))
        fr.keep.append(bsf)
        n_units = (fr.width // 4) * (fr.height // 4)
        # the deblocking passes read these tables: fill them once here, so that they are valid even when --only drops the stage
        lib.vvc355_deblock_bs_pass(None, ptr(d_bsf), ctypes.addressof(bsf))
        lib.vvc355_stream_sync(None)
        def check_bs(fc, orc, env):
            # whole picture: the oracle fills host copies of the ten output tables, compared entry by entry
            outs = [bs_dev[name] for name in bt.OUT]
            for t in outs:
                fr.derived[ptr(t)][...] = 0xEE
            hf = fc.translate(bsf, env.mirror, fc.BS_IN + fc.BS_OUT)
            orc.orc_deblock_bs_pass(ctypes.byref(hf))
            bad = sum(int(not np.array_equal(fr.derived[ptr(t)].ravel(), env.after[ptr(t)].ravel())) for t in outs)
            return n_units, bad

        chain.append(Stage("deblock_bs", "deblock_bs_kernel", lambda st: lib.vvc355_deblock_bs_pass(st, ptr(d_bsf), ctypes.addressof(bsf)),
                           n_units * (24 + 2 * 10 + 6 + 10), writes=[bs_dev[name] for name in bt.OUT], check=check_bs))
                           # MvField + both trees' TU tables + flags read, 10 table bytes written

    def deblock_tables(vertical):
        """The side tables one pass of the stage driver reads (vvc355_deblock_frame): boundary strengths and luma filter lengths
        as the deblock_bs stage leaves them, the chroma transform sizes of the same partition, QP 22..42, per-CTU beta / tc
        offsets, LADF on."""
        tw, th = fr.width // 4, fr.height // 4
        f = abi.DeblockFrame()
        tabs = []
        for c in range(3):
            f.plane[c], f.stride[c], f.bs[c] = ptr(rec[c]), fr.pitch(rec[c]), ptr(bs_dev[f"bs{vertical}{c}"])
        f.max_len_p, f.max_len_q = ptr(bs_dev[f"p{vertical}"]), ptr(bs_dev[f"q{vertical}"])
        f.tb_size_c = ptr(bs_dev["tbw1" if vertical else "tbh1"])
        tabs.append(fr.upload(rng.integers(22, 43, size=(fr.height // 8, fr.width // 8)).astype(np.int8)))
        f.qp_y = ptr(tabs[-1])
        for k in range(2):
            tabs.append(fr.upload(rng.integers(22, 43, size=(th, tw)).astype(np.int8) + 6 * (bd - 8)))
            f.qp_c[k] = ptr(tabs[-1])
        tabs.append(fr.upload(rng.integers(-6, 7, size=(fr.n_ctus, 6)).astype(np.int8)))
        f.db_params = ptr(tabs[-1])
        f.width, f.height, f.min_tu_width, f.min_cb_width, f.ctb_width = fr.width, fr.height, tw, fr.width // 8, fr.ncx
        f.min_cb_log2, f.ctb_log2, f.hs, f.vs, f.n_comp, f.vertical, f.qp_bd_offset = 3, 7, 1, 1, 3, vertical, 6 * (bd - 8)
        f.ladf_enabled, f.num_ladf_intervals, f.ladf_lowest_qp_offset = 1, 3, -2
        f.ladf_qp_offset[0], f.ladf_qp_offset[1] = 1, 3
        f.ladf_lower_bound[1], f.ladf_lower_bound[2] = 1 << (bd - 2), 1 << (bd - 1)
        return f, fr.upload(np.frombuffer(bytes(f), np.uint8))

    for direction, name in ((1, "deblock_vertical"), (0, "deblock_horizontal")):
        if DEBLOCK_JOBS:
            dj = deblock_jobs(direction)
            d_dj = fr.upload(dj.view(np.uint8))
            n_dj = len(dj)
            chain.append(Stage(name, f"deblock_kernel<{bd}>", (lambda p, n: (lambda st: lib.vvc355_deblock_batch(st, bd, p, n)))(ptr(d_dj), n_dj),
                               frame_bytes * 2))
        else:
            hf, d_f = deblock_tables(direction)
            fr.keep.append(hf)

            def check_deblock(fc, orc, env, hf=hf, name=name):
                # whole picture, in place on host copies of the planes as they were before the pass
                work = [env.before[p_].copy() for p_ in rec_ptrs]
                for p_, w_ in zip(rec_ptrs, work):
                    env.mirror.add(p_, w_)
                f = fc.translate(hf, env.mirror, fc.DEBLOCK_PTRS)
                orc.orc_deblock_frame_pass(bd, ctypes.byref(f))
                bad = sum(int(not np.array_equal(w_, env.after[p_])) for p_, w_ in zip(rec_ptrs, work))
                env.stats[name + "_changed_luma_sample_fraction"] = float((env.before[rec_ptrs[0]] != env.after[rec_ptrs[0]]).mean())
                return fr.n_ctus, bad

            chain.append(Stage(name, f"deblock_frame_kernel<{bd}>",
                               (lambda p, hp: (lambda st: lib.vvc355_deblock_frame_pass(st, bd, p, hp)))(ptr(d_f), ctypes.addressof(hf)),
                               frame_bytes * 2, writes=rec, check=check_deblock))

    # ---------------------------------------------------------------- SAO: edge (+ restore at picture borders) or band per CTB
    sj = []
    for c, (w, h) in enumerate(fr.dims):
        cs = CTB if c == 0 else CTB // 2
        x0, y0, cw, ch = batch.ctb_grid(w, h, cs)
        j = batch.job_array(abi.SaoJob, len(x0))
        j["dst"] = ptr(sao[c]) + y0 * fr.pitch(sao[c]) + x0 * isz
        j["src"] = ptr(rec[c]) + y0 * fr.pitch(rec[c]) + x0 * isz
        j["dst_stride"], j["src_stride"], j["w"], j["h"] = fr.pitch(sao[c]), fr.pitch(rec[c]), cw, ch
        j["offset_val"][:, 1:] = rng.integers(-(1 << (bd - 5)) + 1, 1 << (bd - 5), size=(len(x0), 4))
        edge = rng.random(len(x0)) < 0.75
        j["type"] = np.where(edge, 3, 1)
        j["eo"], j["band_position"] = rng.integers(0, 4, size=len(x0)), rng.integers(0, 32, size=len(x0))
        j["borders"][:, 0], j["borders"][:, 1] = x0 == 0, y0 == 0
        j["borders"][:, 2], j["borders"][:, 3] = x0 + cw == w, y0 + ch == h
        sj.append(j)
    sao_all = np.concatenate(sj)
    d_sao = fr.upload(sao_all.view(np.uint8))
    n_sao = len(sao_all)
    if not SAO_TABLES:
        chain.append(Stage("sao", f"sao_vec_kernel<{bd}>", lambda st: lib.vvc355_sao_ctb_batch(st, bd, ptr(d_sao), n_sao, CTB), frame_bytes * 2))
    else:
        # the stage driver: the same per-CTB parameters as a table (fc->tab.sao), one slice, no tiles
        tab = batch.job_array(abi.SaoCtb, fr.n_ctus)
        for c in range(3):
            tab["offset_val"][:, c, :] = sj[c]["offset_val"]
            tab["type_idx"][:, c] = np.where(sj[c]["type"] == 3, 2, 1)
            tab["band_position"][:, c], tab["eo_class"][:, c] = sj[c]["band_position"], sj[c]["eo"]
        d_tab = fr.upload(tab.view(np.uint8))
        d_slice = fr.upload(np.zeros(fr.n_ctus, np.int16))
        d_col, d_row = fr.upload(np.zeros(fr.ncx + 1, np.int16)), fr.upload(np.zeros(fr.ncy + 1, np.int16))
        sf = abi.SaoFrame()
        for c in range(3):
            sf.dst[c], sf.src[c], sf.dst_stride[c], sf.src_stride[c] = ptr(sao[c]), ptr(rec[c]), fr.pitch(sao[c]), fr.pitch(rec[c])
        sf.sao, sf.slice_idx, sf.ctb_to_col_bd, sf.ctb_to_row_bd = ptr(d_tab), ptr(d_slice), ptr(d_col), ptr(d_row)
        sf.width, sf.height, sf.ctb_width, sf.ctb_height = fr.width, fr.height, fr.ncx, fr.ncy
        sf.ctb_log2, sf.hs, sf.vs, sf.n_comp, sf.lfase, sf.no_tile_filter = 7, 1, 1, 3, 1, 0
        d_sf = fr.upload(np.frombuffer(bytes(sf), np.uint8))
        fr.keep.append(sf)
        def check_sao(fc, orc, env):
            work = [env.before[ptr(t)].copy() for t in sao]
            for t, w_ in zip(sao, work):
                env.mirror.add(ptr(t), w_)
            for t in rec:
                env.mirror.add(ptr(t), env.snap(t))
            f = fc.translate(sf, env.mirror, fc.SAO_PTRS)
            orc.orc_sao_frame_pass(bd, ctypes.byref(f))
            bad = sum(int(not np.array_equal(w_[:d_[1], :d_[0]], env.after[ptr(t)][:d_[1], :d_[0]])) for t, w_, d_ in zip(sao, work, fr.dims))
            return fr.n_ctus, bad

        chain.append(Stage("sao", f"sao_frame_kernel<{bd}>", lambda st: lib.vvc355_sao_frame_pass(st, bd, ptr(d_sf), ctypes.addressof(sf)),
                           frame_bytes * 2, writes=sao, check=check_sao))

    if ALF_TABLES:
        # ------------------------------------------------------------ ALF through the stage driver: per-CTB ALFParams + APS tables in,
        # job descriptors built on the device, luma (classify + gather + 7x7 diamond fused), chroma (5x5) and CC-ALF over the
        # WHOLE picture (every CTB has all three flags on; luma filter sets split between the fixed sets and two APSs)
        aps_l = alf_filter_sets(rng, 2)
        d_aps_l = [tuple(fr.upload(a) for a in s[:2]) for s in aps_l]
        d_cco = fr.upload(rng.integers(-64, 64, size=(8, 6)).astype(np.int16))
        d_ccl = fr.upload(rng.integers(0, 4, size=(8, 6)).astype(np.uint8))
        d_ccc = [fr.upload(rng.integers(-32, 32, size=(4, 7)).astype(np.int16)) for _ in range(2)]
        atab = batch.job_array(abi.AlfCtb, fr.n_ctus)
        atab["ctb_flag"][:] = 1
        atab["filt_set_idx_y"] = rng.integers(0, 18, size=fr.n_ctus)
        atab["alt_idx"] = rng.integers(0, 8, size=(fr.n_ctus, 2))
        atab["cc_idc"] = rng.integers(1, 5, size=(fr.n_ctus, 2))
        asl = abi.AlfSlice()
        for k in range(2):
            asl.luma_coeff[k], asl.luma_clip_idx[k] = ptr(d_aps_l[k][0]), ptr(d_aps_l[k][1])
        asl.chroma_coeff, asl.chroma_clip_idx, asl.cc_coeff[0], asl.cc_coeff[1] = ptr(d_cco), ptr(d_ccl), ptr(d_ccc[0]), ptr(d_ccc[1])
        d_atab, d_asl = fr.upload(atab.view(np.uint8)), fr.upload(np.frombuffer(bytes(asl), np.uint8))
        d_aslice = fr.upload(np.zeros(fr.n_ctus, np.int16))
        d_acol, d_arow = fr.upload(np.zeros(fr.ncx + 1, np.int16)), fr.upload(np.zeros(fr.ncy + 1, np.int16))
        af = abi.AlfFrame()
        for c in range(3):
            af.dst[c], af.src[c], af.dst_stride[c], af.src_stride[c] = ptr(out[c]), ptr(sao[c]), fr.pitch(out[c]), fr.pitch(sao[c])
        af.alf, af.slices, af.slice_idx, af.ctb_to_col_bd, af.ctb_to_row_bd = ptr(d_atab), ptr(d_asl), ptr(d_aslice), ptr(d_acol), ptr(d_arow)
        af.width, af.height, af.ctb_width, af.ctb_height = fr.width, fr.height, fr.ncx, fr.ncy
        af.ctb_log2, af.hs, af.vs, af.n_comp, af.lfase, af.lfate = 7, 1, 1, 3, 1, 1
        d_af = fr.upload(np.frombuffer(bytes(af), np.uint8))
        d_awork = fr.upload(np.zeros(lib.vvc355_alf_frame_work_bytes(fr.n_ctus), np.uint8), per_frame=False)       # device scratch: the ALF job arrays
        fr.keep.append(af)
        chroma_bytes = 2 * fr.dims[1][0] * fr.dims[1][1] * isz
        def check_alf(fc, orc, env):
            work = [env.before[ptr(t)].copy() for t in out]
            for t, w_ in zip(out, work):
                env.mirror.add(ptr(t), w_)
            for t in sao:
                env.mirror.add(ptr(t), env.snap(t))
            asl_h = fc.translate(asl, env.mirror, ("luma_coeff", "luma_clip_idx", "chroma_coeff", "chroma_clip_idx", "cc_coeff"))
            env.mirror.add(ptr(d_asl), np.frombuffer(bytes(asl_h), np.uint8).copy())
            f = fc.translate(af, env.mirror, fc.ALF_PTRS)
            orc.orc_alf_frame_pass(bd, ctypes.byref(f))
            bad = sum(int(not np.array_equal(w_[:d_[1], :d_[0]], env.after[ptr(t)][:d_[1], :d_[0]])) for t, w_, d_ in zip(out, work, fr.dims))
            return fr.n_ctus, bad

        # ALF (luma + chroma + CC) = every sample read once and written once, 4 B per sample at 10 bits.  The CTB kernel
        # does exactly that (the luma tile also serves CC-ALF); implementation_bytes adds its aprons (3 luma / 2 chroma samples per CTB side)
        alf_alg = (fr.width * fr.height + chroma_bytes // isz) * isz * 2
        # the per-CTB job builder is a stage of its own, like the inter and transform job builders (it reads only the ALF tables); run once
        # here as well, so that `--only alf` finds its jobs
        lib.vvc355_alf_frame_build(None, bd, ptr(d_af), ctypes.addressof(af), ptr(d_awork))
        chain.append(Stage("alf_job_build", f"alf_build_kernel<{bd}>", lambda st: lib.vvc355_alf_frame_build(st, bd, ptr(d_af), ctypes.addressof(af), ptr(d_awork)),
                           fr.n_ctus * (ctypes.sizeof(abi.AlfCtb) + 5 * ctypes.sizeof(abi.AlfJob) + 128), writes=[d_awork]))
        chain[-1].verify_note = "no check of its own: its descriptors are consumed by `alf`, which is checked over the whole picture"
        st_alf = Stage("alf", f"alf_ctb_kernel<{bd}, true>",
                       lambda st: lib.vvc355_alf_frame_filter(st, bd, ctypes.addressof(af), ptr(d_awork)),
                       alf_alg, writes=out, check=check_alf)
        st_alf.implementation_bytes = alf_alg + fr.n_ctus * ((134 * 144 - 128 * 128) + 2 * (68 * 80 - 64 * 64)) * isz
        chain.append(st_alf)
        return chain

    # ---------------------------------------------------------------- ALF luma: classify + coefficient gather + 7x7 diamond, fused
    sets = alf_filter_sets(rng, 8)
    d_sets = [tuple(fr.upload(a) for a in s) for s in sets]

    def per_ctb(rx, ry):
        s = d_sets[(rx * 3 + ry) % len(d_sets)]
        return ptr(s[0]), ptr(s[1]), ptr(s[2])

    aj = batch.alf_luma_jobs(ptr(out[0]), ptr(sao[0]), fr.pitch(sao[0]), isz, fr.width, fr.height, CTB, per_ctb)
    d_aj = fr.upload(np.frombuffer(bytes(aj), dtype=np.uint8))
    n_aj = len(aj)
    chain.append(Stage("alf_luma_fused", f"alf_luma_kernel<{bd}, 1>", lambda st: lib.vvc355_alf_luma_batch(st, bd, 1, ptr(d_aj), n_aj),
                       fr.width * fr.height * isz * 2))

    # ---------------------------------------------------------------- ALF chroma (5x5 diamond) and cross-component ALF
    cj, ccj = [], []
    clipv = np.array([1 << bd, 1 << (bd - 3), 1 << (bd - 5), 1 << (bd - 7)], np.int16)
    ch_coeff = fr.upload(rng.integers(-64, 64, size=(8, 6)).astype(np.int16))
    ch_clip = fr.upload(clipv[rng.integers(0, 4, size=(8, 6))])
    cc_coeff = fr.upload(rng.integers(-32, 32, size=(8, 8)).astype(np.int16))        # 7 taps, rows padded to 8
    for c in (1, 2):
        w, h = fr.dims[c]
        cs = CTB // 2
        x0, y0, cw, chh = batch.ctb_grid(w, h, cs)
        alt = rng.integers(0, 8, size=len(x0))
        j = batch.job_array(abi.AlfJob, len(x0))
        j["dst"] = ptr(out[c]) + y0 * fr.pitch(out[c]) + x0 * isz
        j["src"] = ptr(sao[c]) + y0 * fr.pitch(sao[c]) + x0 * isz
        j["dst_stride"], j["src_stride"] = fr.pitch(out[c]), fr.pitch(sao[c])
        j["coeff"], j["clip"] = ptr(ch_coeff) + alt * 12, ptr(ch_clip) + alt * 12
        j["w"], j["h"], j["vb_pos"] = cw, chh, cs - 2
        j["ext_l"], j["ext_t"] = np.minimum(2, x0), np.minimum(2, y0)
        j["ext_r"], j["ext_b"] = np.minimum(2, w - x0 - cw), np.minimum(2, h - y0 - chh)
        cj.append(j)
        k = batch.job_array(abi.AlfJob, len(x0))
        k["dst"], k["dst_stride"] = j["dst"], j["dst_stride"]
        k["src"] = ptr(sao[0]) + (2 * y0) * fr.pitch(sao[0]) + (2 * x0) * isz
        k["src_stride"] = fr.pitch(sao[0])
        k["coeff"] = ptr(cc_coeff) + rng.integers(0, 8, size=len(x0)) * 16
        k["w"], k["h"], k["vb_pos"], k["hs"], k["vs"] = cw, chh, CTB - 4, 1, 1
        k["ext_l"] = k["ext_t"] = k["ext_r"] = k["ext_b"] = 3
        # CC-ALF reads one luma sample around the co-located position: keep the outermost chroma ring of the picture out
        inner = (x0 > 0) & (y0 > 0) & (x0 + cw < w) & (y0 + chh < h)
        ccj.append(k[inner])
    c_all, cc_all = np.concatenate(cj), np.concatenate(ccj)
    d_c, d_cc = fr.upload(c_all.view(np.uint8)), fr.upload(cc_all.view(np.uint8))
    n_c, n_cc = len(c_all), len(cc_all)
    chroma_bytes = 2 * fr.dims[1][0] * fr.dims[1][1] * isz
    chain.append(Stage("alf_chroma", f"alf_chroma_kernel<{bd}>", lambda st: lib.vvc355_alf_chroma_batch(st, bd, ptr(d_c), n_c), chroma_bytes * 2))
    chain.append(Stage("alf_cc", f"alf_cc_kernel<{bd}>", lambda st: lib.vvc355_alf_cc_batch(st, bd, ptr(d_cc), n_cc),
                       chroma_bytes * 2 + fr.width * fr.height * isz))
    return chain


class VerifyEnv:
    """What a stage's check sees: host snapshots of the tensors the stage writes (before / after its launch), the host mirror of
    every uploaded table, the sampled CTUs, and a place to leave workload statistics."""

    def __init__(self, torch, mirror, picks):
        self.torch, self.mirror, self.picks = torch, mirror, picks
        self.before, self.after, self.stats = {}, {}, {}

    def snap(self, t):
        return t.cpu().numpy()


def verify_step(lib, torch, frame, chain, n_ctus):
    """One fresh, untimed step, stage by stage, every stage's device output compared with the CPU oracle (tests/frame_check.py):
    sampled CTUs for the prediction / transform stages, the whole picture for the table-driven loop-filter stages.  This is where
    the picture-size-dependent code (XCD renumbering over ~26k workgroups, 24-bit row offsets, 32-bit plane offsets) is checked at
    the bench's own size.  Returns {"stages": {name: {...}}, "stats": {...}}; raises SystemExit on any mismatch."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import frame_check as fc
    from conftest import load_oracle
    orc = load_oracle()
    fc.bind(orc)
    mirror = fc.Mirror()
    for dev_ptr, host in list(frame.host.items()) + list(frame.derived.items()):
        mirror.add(dev_ptr, host)
    for r in frame.refs:
        for t in r:
            mirror.add(t.data_ptr(), t.cpu().numpy())
    picks = fc.sample_ctus(np.random.default_rng(0xC7), frame.ncx, frame.ncy, n_ctus)
    env = VerifyEnv(torch, mirror, picks)
    stream = torch.cuda.current_stream().cuda_stream
    report, failed = {}, []
    for st in chain:
        env.before = {t.data_ptr(): env.snap(t) for t in st.writes} if st.check else {}
        st.launch(stream)
        torch.cuda.synchronize()
        if st.check is None:
            report[st.name] = {"checked": 0, "mismatching": 0, "note": getattr(st, "verify_note", "not checked")}
            continue
        env.after = {t.data_ptr(): env.snap(t) for t in st.writes}
        t0 = time.perf_counter()
        units, bad = st.check(fc, orc, env)
        report[st.name] = {"checked": int(units), "mismatching": int(bad), "oracle_s": round(time.perf_counter() - t0, 2)}
        if bad:
            failed.append(st.name)
    if failed:
        print(json.dumps({"verified": report}), file=sys.stderr, flush=True)
        raise SystemExit(f"bench.py: device output differs from the oracle in stage(s): {', '.join(failed)}")
    return {"stages": report, "stats": env.stats, "ctus_sampled": len(picks)}


def self_check(width, height, bd, n_ctus=12, noise=False):
    """Build the chain at the given size and run the oracle check of one step (tests/test_frame_check_gpu.py)."""
    import torch
    lib = abi.load()
    frame = Frame(torch, width, height, bd, seed=0x5EED0001)
    frame.noise = noise
    chain = build_chain(lib, torch, frame)
    stream = torch.cuda.current_stream().cuda_stream
    for st in chain:
        st.launch(stream)
    torch.cuda.synchronize()
    return verify_step(lib, torch, frame, chain, n_ctus)


def recorded_traffic(root, stage_name):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/pmc_traffic.json), or None."""
    path = os.path.join(root, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None
    return json.load(open(path)).get(stage_name, {}).get("hbm_bytes_per_launch")


def recorded_valu(root, stage_name):
    """VALU instruction counts / issue utilisation of the stage's kernel from the committed rocprofv3 --pmc SQ_* passes
    (profiles/mc_valu.json), or None."""
    path = os.path.join(root, "profiles", "mc_valu.json")
    if not os.path.exists(path):
        return None
    d = json.load(open(path)).get(stage_name)
    return None if d is None else {k: d[k] for k in ("valu_insts_per_wave", "valu_issue_utilisation", "by_tool_valu_per_wave")}


def cpu_chain(root, bd):
    """One inter CTU's worth of every stage through the CPU oracle (oracle/liborc.so, a scalar C restatement): the unit of work the
    CPU baseline repeats.  Returns a callable; nothing here touches the GPU or torch."""
    import subprocess
    so = os.path.join(root, "oracle", "liborc.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-s", "-C", os.path.join(root, "oracle")])
    orc = ctypes.CDLL(so)
    abi.bind(orc, "orc_", {k: v for k, v in abi.SLOT_SIGNATURES.items() if hasattr(orc, "orc_" + k)})
    rng = np.random.default_rng(7)
    dt = np.uint8 if bd == 8 else np.uint16
    isz = np.dtype(dt).itemsize
    addr = {}

    def A(a, o=0):                      # cached: numpy's .ctypes accessor costs about a microsecond, as much as a small oracle call
        k = id(a)
        if k not in addr:
            addr[k] = (a, a.ctypes.data)
        return addr[k][1] + o * a.itemsize
    luma = rng.integers(0, 1 << bd, size=(CTB + 16, CTB + 32)).astype(dt)
    ls = luma.shape[1]
    dst = np.zeros((CTB, CTB), dt)
    coeff_set, clip_idx, c2f = alf_filter_sets(rng, 1)[0]
    n = (CTB // 4) ** 2
    cls, tr = np.zeros(n, np.int32), np.zeros(n, np.int32)
    grad = np.zeros(((CTB + 4) // 2) ** 2 * 4, np.int32)
    coeff, clip = np.zeros((n, 12), np.int16), np.zeros((n, 12), np.int16)
    ch_c, ch_cl = rng.integers(-64, 64, size=6).astype(np.int16), np.full(6, 1 << bd, np.int16)
    cc_c = rng.integers(-32, 32, size=7).astype(np.int16)
    offs = np.array([0, 3, -2, 1, -4], np.int16)
    lut = np.sort(rng.integers(0, 1 << bd, size=1 << bd)).astype(dt)
    beta = np.array([40, 44, 40, 44], np.int32); tc = np.array([11, 0, 14, 9], np.int32)
    z4 = np.zeros(4, np.uint8); l3 = np.full(4, 3, np.uint8)
    borders = np.zeros(4, np.int32)
    res = {s: rng.integers(-(1 << 12), 1 << 12, size=(s, s)).astype(np.int32) for s in (4, 8, 16, 32, 64)}
    work = {k: np.empty_like(v) for k, v in res.items()}
    off = 8 * ls + 8
    saosrc = rng.integers(0, 1 << bd, size=(CTB + 2, 320 // isz)).astype(dt)

    # bi-prediction jobs of one CTU: two random reference pictures of 192x192 around it, motion within +-24 samples
    orc.orc_bipred_block.argtypes = [ctypes.c_int, ctypes.POINTER(abi.BipredJob)]
    orc.orc_bipred_block.restype = None
    bp_ref = [[rng.integers(0, 1 << bd, size=(192 >> (c > 0), 192 >> (c > 0))).astype(dt) for c in range(3)] for _ in range(2)]
    bp_rec = (abi.BipredResult * 64)()
    bp_jobs = []
    for b in range(64):
        mvb = [int(v) for v in rng.integers(-24 * 16, 24 * 16 + 1, size=4)]
        row = []
        for c in range(3):
            sh = 1 if c else 0
            j = abi.BipredJob()
            j.dst, j.dst_stride = A(dst), CTB * isz
            j.ref0, j.ref1 = A(bp_ref[0][c]), A(bp_ref[1][c])
            j.ref0_stride = j.ref1_stride = bp_ref[0][c].shape[1] * isz
            j.rec = ctypes.addressof(bp_rec[b])
            for k in range(4):
                j.mv[k] = mvb[k]
            j.x, j.y, j.w, j.h = (32 + (b % 8) * 16) >> sh, (32 + (b // 8) * 16) >> sh, 16 >> sh, 16 >> sh
            j.pic_w = j.pic_h = 192 >> sh
            j.chroma, j.hs, j.vs, j.dmvr, j.bdof = int(c > 0), 1, 1, 1, 1
            row.append(j)
        bp_jobs.append(row)

    # boundary strengths of one CTB's worth of side tables (same partition statistics as build_chain)
    sys.path.insert(0, os.path.join(root, "tests"))
    import bs_cases
    bs_t = bs_cases.BsTables(np.random.default_rng(0x5EED0B5), CTB, CTB, 7, split=(0.95, 0.45), cbf_p=0.4)
    bs_f = bs_t.frame(lambda name: getattr(bs_t, name).ctypes.data)
    orc.orc_deblock_bs_pass.argtypes = [ctypes.POINTER(abi.BsFrame)]
    orc.orc_deblock_bs_pass.restype = None

    def one_ctu():
        # regular bi-prediction with DMVR + BDOF: 64 luma 16x16 sub-blocks, then their 2 x 64 chroma 8x8 blocks
        for b in range(64):
            for c in range(3):
                orc.orc_bipred_block(bd, ctypes.byref(bp_jobs[b][c]))
        # inverse transform + residual add: the luma and chroma TB mix of build_chain
        for (s, cnt) in ((64, 1), (32, 4 + 2), (16, 16 + 8), (8, 64 + 32), (4, 128)):
            for _ in range(cnt):
                r = work[s]
                r[:] = res[s]
                lg = s.bit_length() - 1
                orc.orc_dequant(A(r), lg, lg, 0, 0, min(s, 12) - 1, min(s, 12) - 1, 30, 0, 1, bd, 15, None, 1, -1)
                orc.orc_itx(0, 0, lg, lg, A(r), min(s, 12), min(s, 12), 15, bd)
                orc.orc_add_residual(bd, A(dst), A(r), s, s, CTB * isz)
        orc.orc_lmcs_filter(bd, A(dst), CTB * isz, CTB, CTB, A(lut))
        orc.orc_deblock_bs_pass(ctypes.byref(bs_f))
        # deblock: 15 x 16 vertical and horizontal luma edge segments pairs + chroma
        for d in (1, 0):
            for e in range(15 * 16 + 2 * 7 * 8):
                o = (8 + (e % 14) * 8) * CTB + 8 + ((e // 14) % 14) * 8
                orc.orc_lf_filter_luma(bd, d, A(dst, o), CTB * isz, A(beta), A(tc), A(z4), A(z4), A(l3), A(l3), 0)
        # SAO edge on the three CTBs
        for (s, reps) in ((CTB, 1), (CTB // 2, 2)):
            for _ in range(reps):
                orc.orc_sao_edge_filter(bd, A(dst), A(saosrc, 320 // isz + 8), CTB * isz, A(offs), 2, s, s)
                orc.orc_sao_edge_restore(bd, 0, A(dst), A(saosrc, 320 // isz + 8), CTB * isz, 320, A(offs), 2, A(borders), s, s, A(z4), A(z4), A(z4))
        # ALF luma (classify + recon + filter), chroma x2, CC x2
        orc.orc_alf_classify(bd, A(cls), A(tr), A(luma, off), ls * isz, CTB, CTB, CTB - 4, A(grad))
        orc.orc_alf_recon_coeff_and_clip(bd, A(coeff), A(clip), A(cls), A(tr), n, A(coeff_set), A(clip_idx), A(c2f))
        orc.orc_alf_filter_luma(bd, A(dst), CTB * isz, A(luma, off), ls * isz, CTB, CTB, A(coeff), A(clip), CTB - 4)
        for _ in range(2):
            orc.orc_alf_filter_chroma(bd, A(dst), CTB * isz, A(luma, off), ls * isz, 64, 64, A(ch_c), A(ch_cl), 62)
            orc.orc_alf_filter_cc(bd, A(dst), CTB * isz, A(luma, off), ls * isz, 64, 64, 1, 1, A(cc_c), CTB - 4)

    one_ctu.keep = (bp_ref, bp_rec, bp_jobs, bs_t, bs_f, dst)      # the job structs hold raw addresses into these
    return one_ctu


def cpu_time_chain(one_ctu, seconds):
    """Repeat the chain for about `seconds`; returns (CTUs done, elapsed)."""
    one_ctu()
    n, t_0 = 0, time.perf_counter()
    while True:
        one_ctu()
        n += 1
        dt_s = time.perf_counter() - t_0
        if dt_s >= seconds:
            return n, dt_s


def cpu_worker_main(args):
    """``--cpu-worker SECONDS``: one host thread of the all-core CPU baseline.  A fresh process that never imports torch and never
    makes a HIP call; prints {"ctus": n, "s": elapsed}."""
    n, dt_s = cpu_time_chain(cpu_chain(ROOT, args.bd), args.cpu_worker)
    print(json.dumps({"ctus": n, "s": dt_s}), flush=True)
    return 0


def host_cpu():
    model = "?"
    try:
        with open("/proc/cpuinfo") as f:
            model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "?")
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    return model, os.cpu_count() or 1, usable


def cgroup_cpu_quota():
    """CPU time this process's cgroup may use, in cores (cgroup v2 cpu.max, else v1 cfs quota), or None when unlimited / unreadable:
    sched_getaffinity can list every hardware thread of the host while the container is granted a fraction of them."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        return None if quota == "max" else float(quota) / float(period)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            quota = float(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            period = float(f.read())
        return None if quota <= 0 else quota / period
    except (OSError, ValueError):
        return None


def cpu_baseline(root, fr, budget_s, max_workers=0):
    """The CPU oracle (kind "port") timed on the host: first on ONE core, then on every core this process may run on (one worker
    process per core, each repeating whole CTUs — one CTU per task, no shared state), over a bounded sample of the same per-CTU work:
    one inter CTU's worth of every stage.  `value` is the all-core figure in frames/s of the same chain; `one_core` sits beside it."""
    import subprocess
    model, nproc, usable = host_cpu()
    one = cpu_chain(root, fr.bd)
    n1, dt1 = cpu_time_chain(one, min(6.0, budget_s / 2))
    one_core = (n1 / fr.n_ctus) / dt1
    quota = cgroup_cpu_quota()
    granted = usable if quota is None else max(1, min(usable, int(quota + 0.5)))          # workers beyond the CPU quota only time-share
    cores = granted if max_workers <= 0 else min(granted, max_workers)
    t_each = max(2.0, budget_s - dt1 - 2.0)
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker", f"{t_each:.2f}", "--bd", str(fr.bd)]
    env = dict(os.environ, OMP_NUM_THREADS="1", HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    procs = [subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True) for _ in range(cores)]
    rate, n_all, t_max = 0.0, 0, 0.0
    for pr in procs:
        o, _ = pr.communicate(timeout=t_each * 4 + 120)
        if pr.returncode != 0:
            raise RuntimeError("CPU baseline worker failed")
        d = json.loads(o.strip().splitlines()[-1])
        rate += d["ctus"] / d["s"]
        n_all += d["ctus"]
        t_max = max(t_max, d["s"])
    # what the workers really got: the aggregate rate in units of the one-core rate (contention, SMT siblings and quotas show up here)
    effective = (rate / fr.n_ctus) / one_core if one_core > 0 else float(cores)
    return {
        "value": rate / fr.n_ctus,
        "unit": "frames/s",
        "cores": cores,
        "effective_cores": round(effective, 1),
        "kind": "port",
        "one_core": one_core,
        "host": {"cpu_model": model, "nproc": nproc, "usable_by_affinity": usable, "cgroup_cpu_quota_cores": quota},
        "sample": f"{n_all} CTUs in {t_max:.1f} s on {cores} worker processes ({n1} CTUs in {dt1:.1f} s on one core first); a frame is "
                  f"{fr.n_ctus} CTUs (128x128, {fr.bd}-bit 4:2:0); each CTU goes through the same stage chain as one inter CTU (bi-prediction "
                  f"with DMVR + BDOF, dequant + itx + residual, LMCS, boundary strengths, deblock, SAO, ALF + CC-ALF)",
        "reference_c": "not built here (needs configure-generated headers); BASELINE.md section 2's single-thread rates of the reference's "
                       "C path (Xeon 2.1 GHz, no assembler) sum to about 2.8 s per 8K inter frame, i.e. about 0.36 frames/s per core; the "
                       "oracle is a plain restatement and runs the same functions 1-8x slower (profiles/r02_oracle_rates_container.json)",
    }


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(n, argv):
    """``--gpus N`` with no rendezvous in the environment: start N fresh rank processes, one per GPU, and relay rank 0's JSON line.

    The parent never imports torch and never makes a HIP call, and the children are new processes started with subprocess (never
    an exec of a process that has touched the GPU).  Any failing rank fails the run: the others are stopped by PID and the exit code
    is non-zero.  Under ``python -m torch.distributed.run`` the environment already carries the rendezvous and this is not used."""
    import subprocess
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True))
    import threading
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)     # rank 0's stdout never blocks on a full pipe
    reader.start()
    failed = None
    while failed is None:
        rcs = [p.poll() for p in procs]
        bad = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad:
            failed = bad[0]
        elif all(rc == 0 for rc in rcs):
            break
        else:
            time.sleep(0.05)
    if failed is not None:
        for p in procs:
            if p.poll() is None:
                p.terminate()                       # exact PIDs of the children started above
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
        raise SystemExit(f"bench.py: rank {failed[0]} of {n} exited with code {failed[1]}")
    reader.join(timeout=10)
    out0 = out0[0] if out0 else ""
    lines = [ln for ln in out0.splitlines() if ln.startswith("{")]
    if not lines:
        raise SystemExit("bench.py: rank 0 printed no JSON line")
    line = json.loads(lines[-1])
    if line.get("n_gpus") != n:
        raise SystemExit(f"bench.py: rank 0 reports n_gpus = {line.get('n_gpus')}, expected {n}")
    print(lines[-1], flush=True)


def rendezvous(args):
    """(world, rank, local_rank, dist): one process per GPU; torch.distributed (gloo: the ranks exchange one barrier and one float,
    there is no data-path collective and no RCCL traffic) only when world > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE = {world}: launch with --gpus equal to the number of ranks")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    return world, rank, local_rank, dist


def stub_main(args, world, rank, dist):
    """The bench protocol around a host sleep instead of the GPU chain (TEST AID, --stub-step-ms): warm-up, barrier, exactly K timed
    steps, barrier, max over ranks, one JSON line from rank 0."""
    import torch
    step = lambda: time.sleep(args.stub_step_ms * 1e-3 * (1 + rank))      # noqa: E731  (rank r is slower: the max must pick the last rank)
    for _ in range(args.warmup):
        step()
    sharding.barrier(dist, world)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    elapsed = time.perf_counter() - t0
    sharding.barrier(dist, world)
    elapsed = sharding.max_over_ranks(dist, torch, world, elapsed, "cpu")
    if rank == 0:
        print(json.dumps({"metric": "STUB (host sleep, no GPU work)", "stub": True, "value": world * args.steps / elapsed, "unit": "frames/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "none", "data": "none",
                          "config": {"workload": "stub"}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def gop_order(g):
    """Decoding order of a hierarchical-B group of pictures of size g (a power of two) as (POC, lower reference POC, upper reference
    POC): POC g from POC 0 (the previous group's POC g) on both lists, then every interval's middle picture from the interval's two
    ends, coarsest level first — the random-access structure of the VVC common test conditions."""
    out, level = [(g, 0, 0)], [(0, g)]
    while level and level[0][1] - level[0][0] > 1:
        nxt = []
        for lo, hi in level:
            mid = (lo + hi) // 2
            out.append((mid, lo, hi))
            nxt += [(lo, mid), (mid, hi)]
        level = nxt
    return out


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--width", type=int, default=7680)
    ap.add_argument("--height", type=int, default=4320)
    ap.add_argument("--bd", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gop", type=int, default=16, help="pictures per hierarchical-B group (a power of two): a step decodes one group of one stream, every picture "
                                                        "behind its two reference pictures; 0 = independent frames (--frames-in-flight)")
    ap.add_argument("--frames-in-flight", type=int, default=8, help="independent frames processed concurrently per step (one HIP stream each): the whole step with --gop 0, "
                                                                    "the secondary `independent_frames` figure otherwise; 1 = latency of a single frame")
    ap.add_argument("--with-upload", action="store_true", help="--gop 0: additionally time the steps with every per-frame descriptor copied from pinned host memory first "
                                                               "(GOP mode always reports it: incl_descriptor_upload)")
    ap.add_argument("--no-upload", action="store_true", help="GOP mode: skip the incl_descriptor_upload leg")
    ap.add_argument("--gop-check", type=int, default=0, metavar="GROUPS",
                    help="GOP mode, no timing: decode GROUPS groups with the concurrent scheduler (one stream per picture, event waits on the reference pictures), "
                         "again one picture at a time in decoding order on one stream, and once more serially in REVERSED order; prints how many decoded pictures "
                         "differ between the first two (must be 0) and between the last two (must not be 0: the pictures really read their references)")
    ap.add_argument("--no-verify", action="store_true", help="skip the untimed oracle check of one step (the `verified` object)")
    ap.add_argument("--verify-ctus", type=int, default=32, help="CTUs sampled for the prediction / transform stages of the oracle check")
    ap.add_argument("--no-lmcs", action="store_true", help="profiling aid: a picture without LMCS (no forward map on the inter prediction, no chroma residual scaling)")
    ap.add_argument("--noise", action="store_true", help="profiling aid: uniformly random reference samples and dense residuals (checkasm-style) "
                                                         "instead of picture-like content: every DMVR search runs to the end, deblocking mostly early-outs")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="rough wall-clock budget of the CPU baseline leg (one core first, then all cores)")
    ap.add_argument("--cpu-workers", type=int, default=0, help="worker processes of the all-core CPU baseline (0 = one per usable core)")
    ap.add_argument("--cpu-worker", type=float, default=None, help="INTERNAL: run as one CPU-baseline worker for this many seconds (no GPU, no torch)")
    ap.add_argument("--mc-tools", type=int, default=3, help="profiling aid: 1 = DMVR, 2 = BDOF, 3 = both (the metric's workload)")
    ap.add_argument("--graph", action="store_true", help="replay the step as one captured hipGraph in the timed region (per-stage times then come from an untimed pass)")
    ap.add_argument("--alf-jobs", action="store_true", help="profiling aid: ALF from host-built per-CTB jobs (three batch launches, CC-ALF without the outermost CTB ring) instead of the stage driver")
    ap.add_argument("--sao-jobs", action="store_true", help="profiling aid: SAO from host-built per-CTB jobs (vvc355_sao_ctb_batch) instead of the stage driver")
    ap.add_argument("--deblock-jobs", action="store_true",
                    help="profiling aid: deblock from host-built edge jobs (vvc355_deblock_batch) instead of the stage driver")
    ap.add_argument("--affine-frac", type=float, default=0.06, help="fraction of the inter CTUs that are affine (+PROF); the metric uses 0.06")
    ap.add_argument("--inter-frac", type=float, default=0.8, help="fraction of inter CTUs; the metric uses 0.8; 0 = all-intra picture (BASELINE.json configs[1]: the in-order "
                                                                 "intra pass over every CTU, a dense wavefront)")
    ap.add_argument("--only", type=str, default="", help="comma-separated stage names (profiling aid; default = full chain)")
    ap.add_argument("--stub-step-ms", type=float, default=None,
                    help="TEST AID: replace the GPU chain by a host sleep of this many ms per step (exercises the launcher, the rendezvous, "
                         "the barrier / max-over-ranks protocol and the JSON line on a machine without a GPU); the line says \"stub\": true")
    return ap.parse_args(argv)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.cpu_worker is not None:
        return cpu_worker_main(args)
    # One HIP stream per frame in flight only helps if the streams do not share hardware queues: the runtime's default of 4 maps
    # several streams onto one in-order queue, and a frame's 2 ms intra pass then holds up the other frame behind it (measured, 8
    # frames in flight: 502 / 660 / 685 / 808 frames/s with 2 / 4 / 8 / 16 queues).  Set before the HIP runtime initialises; a
    # value already in the environment wins.  A host that uses the C ABI with several streams needs the same setting.  Every stream counts:
    # the GOP scheduler has 16 picture streams, two copy streams and a join stream — with 16 queues the copy streams shared queues with
    # picture streams and the upload leg fell from 498 to 390 frames/s; with 20: 524 (and the stream without upload 575 -> 595).
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(max(20, 2 * max(1, args.frames_in_flight) + 4)))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus, argv)         # before anything imports torch or touches HIP
    global MC_TOOLS, AFFINE_FRAC, DEBLOCK_JOBS, SAO_TABLES, ALF_TABLES, INTER_FRAC, LMCS
    LMCS = not args.no_lmcs
    INTER_FRAC = args.inter_frac
    SAO_TABLES = not args.sao_jobs
    ALF_TABLES = not args.alf_jobs
    DEBLOCK_JOBS = args.deblock_jobs
    MC_TOOLS = args.mc_tools & 3
    AFFINE_FRAC = args.affine_frac
    world, rank, local_rank, dist = rendezvous(args)
    if args.stub_step_ms is not None:
        return stub_main(args, world, rank, dist)
    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    # one rank per GPU; more ranks than devices (a rehearsal of the multi-rank path on a one-GPU box) share devices round-robin
    n_dev = torch.cuda.device_count()
    if local_rank >= n_dev:
        print(f"bench.py: rank {rank}: {n_dev} device(s) for local rank {local_rank} - sharing device {local_rank % n_dev} (rehearsal, not a measurement)", file=sys.stderr)
    local_dev = local_rank % n_dev
    torch.cuda.set_device(local_dev)
    lib = abi.load()
    lib.vvc355_set_device(local_dev)

    # ---- the frames of a step
    # GOP mode (default, --gop 16): a step decodes one hierarchical-B group of pictures of ONE stream — POC G from the previous group's
    # POC G, then G/2 from (0, G), G/4 and 3G/4, ... — every picture's inter prediction reading the decoded output (ALF planes) of its
    # two reference pictures, every picture on a HIP stream of its own behind event waits on its references (what the reference's
    # frame threads do with ff_vvc_report_progress / vvc_refs.h:37-56, at picture granularity).  Consecutive groups overlap: group
    # k + 1 only needs group k's POC G.  Three rotating sets of G frame objects keep a picture's buffers untouched until nothing reads
    # them any more (a decoder's DPB does the same with fresh frame buffers): group k uses set k % 3 and starts after group k - 2.
    # --gop 0: F independent frames per step (--frames-in-flight), no dependencies: the device's throughput ceiling.
    gop = max(0, args.gop)
    n_ff = max(1, args.frames_in_flight)
    n_sets = 3
    streams = [torch.cuda.Stream() for _ in range(16 if gop else n_ff)]
    t_build = time.perf_counter()
    if gop:
        order = gop_order(gop)
        objs = {(j, poc): Frame(torch, args.width, args.height, args.bd, seed=0x5EED0001 + rank + 977 * (j * gop + poc)) for j in range(n_sets) for poc in range(1, gop + 1)}
        for fr_i in objs.values():
            fr_i.pin = not args.no_upload            # the per-frame arena's host twin is pinned: the upload leg copies from it

        def ref_obj(j, poc):
            return objs[((j - 1) % n_sets, gop)] if poc == 0 else objs[(j, poc)]
        chains_of = {}
        for (j, poc), fr_i in objs.items():
            lo, hi = next((lo_, hi_) for (p_, lo_, hi_) in order if p_ == poc)
            fr_i.noise = args.noise
            fr_i.ref_frames = (ref_obj(j, lo), ref_obj(j, hi))
        for key, fr_i in objs.items():
            # every picture starts out as picture-like content, so that the first groups predict from something sensible
            for c in range(3):
                fr_i.out[c].copy_(fr_i.picture(c, 0)[:, :fr_i.out[c].shape[1]])
                fr_i.keep.pop()
            ch_i = build_chain(lib, torch, fr_i)
            chains_of[key] = [st for st in ch_i if st.name in args.only.split(",")] if args.only else ch_i
        frames = [objs[(0, poc)] for poc in range(1, gop, 2)][:n_ff] or [objs[(0, 1)]]        # the odd POCs of a group do not depend on each other
        chains = [chains_of[(0, poc)] for poc in range(1, gop, 2)][:n_ff] or [chains_of[(0, 1)]]
        n_ff = len(frames)
        done_ev = {key: torch.cuda.Event() for key in objs}
        copy_ev = {key: torch.cuda.Event() for key in objs}
        copy_streams = [torch.cuda.Stream() for _ in range(2)]
        recorded = set()
        gop_ev = [torch.cuda.Event() for _ in range(n_sets)]
        join_stream = torch.cuda.Stream()
    else:
        frames, chains = [], []
        for i in range(n_ff):
            fr_i = Frame(torch, args.width, args.height, args.bd, seed=0x5EED0001 + rank + 977 * i)
            fr_i.noise = args.noise
            ch_i = build_chain(lib, torch, fr_i)
            if args.only:
                ch_i = [st for st in ch_i if st.name in args.only.split(",")]
            frames.append(fr_i)
            chains.append(ch_i)
    frame, chain = frames[0], chains[0]
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t_build

    # events are created before the timed region; inside it they are only recorded
    pool = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in chain] for _ in range(args.steps)]

    def run_frame(f, events=None, step=0, only=None, stream=None):
        ts = stream if stream is not None else streams[f % len(streams)]
        sh = ts.cuda_stream
        for i, st in enumerate(chains[f]):
            if f == 0 and events is not None and (only is None or st.name == only):
                e0, e1 = pool[step][i]
                e0.record(ts)
                st.launch(sh)
                e1.record(ts)
                events.setdefault(st.name, []).append((e0, e1))
            else:
                st.launch(sh)

    def run_independent(events=None, step=0, only=None):
        for f in range(n_ff):
            run_frame(f, events, step, only)

    pinned_of = {}

    def run_gop(k, events=None, step=0, only=None, upload=False):
        j = k % n_sets
        for i, (poc, lo, hi) in enumerate(order):
            ts = streams[i % len(streams)]
            if upload:
                # what the decoder's host side sends for this picture: every per-frame table, record and descriptor array from pinned memory.
                # The copies do not depend on the reference pictures, so they go out on a copy stream as soon as the picture object is free
                # (its previous use, three groups back, is done) and the picture's own stream only waits for their event: a decoder that has
                # parsed ahead uploads ahead.
                cs = copy_streams[i % len(copy_streams)]
                if (j, poc) in recorded:
                    cs.wait_event(done_ev[(j, poc)])
                with torch.cuda.stream(cs):
                    for dst_t, src_t in pinned_of[(j, poc)]:
                        dst_t.copy_(src_t, non_blocking=True)
                copy_ev[(j, poc)].record(cs)
                ts.wait_event(copy_ev[(j, poc)])
            if k >= 2:
                ts.wait_event(gop_ev[(k - 2) % n_sets])            # nothing reads this set's pictures any more
            timed_picture = (j, poc) == (0, 1) and events is not None          # the frame whose dominant stage carries the events
            sh = ts.cuda_stream
            if not timed_picture:
                # the stages that read only what the parser produced (table fills, job builders, the intra blocks' transforms, boundary
                # strengths) do not depend on the reference pictures: they go out before the picture waits for them
                for st in chains_of[(j, poc)]:
                    if st.name in REF_FREE_STAGES:
                        st.launch(sh)
            for dep in {((j - 1) % n_sets, gop) if q == 0 else (j, q) for q in (lo, hi)}:
                if dep in recorded:
                    ts.wait_event(done_ev[dep])
            if timed_picture:
                run_frame(0, events, step, only, stream=ts)
            else:
                for st in chains_of[(j, poc)]:
                    if st.name not in REF_FREE_STAGES:
                        st.launch(sh)
            done_ev[(j, poc)].record(ts)
            recorded.add((j, poc))
        for (poc, _lo, _hi) in order:
            join_stream.wait_event(done_ev[(j, poc)])
        gop_ev[j].record(join_stream)

    gop_k = [0]

    def run_step(events=None, step=0, only=None):
        if gop:
            run_gop(gop_k[0], events, step, only)
            gop_k[0] += 1
        else:
            run_independent(events, step, only)
    frames_per_step = gop if gop else n_ff

    def barrier():
        sharding.barrier(dist, world, torch.cuda.synchronize)

    # vvc355_recon_frame.workgroups: 0 (the pass's default, fastest for a picture alone) for the one-picture legs, fewer persistent workgroups
    # while several pictures share the device (measured, eight in flight: 96 > 128 > 192 > 256 > 384 in frames/s; profiles/README.md)
    def recon_workgroups(n):
        for rf_ in RECON_FRAMES:
            rf_.workgroups = n
    concurrent_wgs = 96 if (gop or n_ff > 1) else 0
    recon_workgroups(concurrent_wgs)
    if args.gop_check:
        if not gop:
            raise SystemExit("--gop-check needs --gop")
        keys = sorted(objs)
        state = [t for key in keys for t in objs[key].out]            # the decoded pictures: what later pictures predict from
        init = [t.clone() for t in state]

        def restart():
            torch.cuda.synchronize()
            for t, t0 in zip(state, init):
                t.copy_(t0)
            recorded.clear()
            torch.cuda.synchronize()

        def serial(pictures):
            restart()
            for k in range(args.gop_check):
                for (poc, _lo, _hi) in pictures:
                    for st in chains_of[(k % n_sets, poc)]:
                        st.launch(streams[0].cuda_stream)
                    torch.cuda.synchronize()
            return [t.clone() for t in state]

        restart()
        for k in range(args.gop_check):
            run_gop(k)
        torch.cuda.synchronize()
        concurrent = [t.clone() for t in state]
        in_order, reversed_order = serial(order), serial(order[::-1])
        n_pic = len(keys)
        per_pic = lambda a, b: sum(any(not torch.equal(a[3 * i + c], b[3 * i + c]) for c in range(3)) for i in range(n_pic))      # noqa: E731
        rep = {"gop_check": {"groups": args.gop_check, "pictures": n_pic, "gop": gop, "streams": len(streams),
                             "concurrent_vs_in_order_mismatching_pictures": per_pic(concurrent, in_order),
                             "in_order_vs_reversed_order_differing_pictures": per_pic(in_order, reversed_order),
                             "pictures_changed_by_decoding": per_pic(in_order, init)}}
        print(json.dumps(rep))
        ok = rep["gop_check"]["concurrent_vs_in_order_mismatching_pictures"] == 0 and rep["gop_check"]["in_order_vs_reversed_order_differing_pictures"] > 0
        raise SystemExit(0 if ok else 1)
    for _ in range(args.warmup):
        run_step()
    barrier()
    recon_workgroups(0)
    # Untimed pass of ONE frame alone with HIP events around every stage: the per-stage breakdown (`stages`), the latency of a frame
    # and which stage dominates.  The timed region then carries events around that one stage only.
    events = {}
    t_lat = time.perf_counter()
    for k in range(args.steps):
        run_frame(0, events, k)
    torch.cuda.synchronize()
    frame_latency_ms = (time.perf_counter() - t_lat) / args.steps * 1e3
    breakdown = {name: float(np.mean([a.elapsed_time(b) for a, b in ev])) for name, ev in events.items()}
    # the in-order intra pass is a latency-bound dependent chain on a few hundred waves; with several frames in flight it runs
    # beside the other frames' kernels, so the kernel that bounds throughput is the largest of the batched stages
    throughput_stages = {k: v for k, v in breakdown.items() if k != "intra_recon_wavefront"} or breakdown
    dom_name = max(throughput_stages, key=throughput_stages.get)
    recon_workgroups(concurrent_wgs)
    barrier()
    # secondary figure: F mutually independent frames per step (in GOP mode: the odd pictures of one group) — the ceiling without
    # reference dependencies
    independent = None
    if gop and not args.graph:
        for _ in range(2):
            run_independent()
        torch.cuda.synchronize()
        t_i = time.perf_counter()
        n_i = max(4, args.steps // 2)
        for _ in range(n_i):
            run_independent()
        torch.cuda.synchronize()
        el_i = sharding.max_over_ranks(dist, torch, world, time.perf_counter() - t_i, "cpu")
        independent = {"value": world * n_ff * n_i / el_i, "unit": "frames/s", "frames_in_flight": n_ff, "steps": n_i,
                       "note": "mutually independent frames (the odd pictures of one group), one HIP stream each, no reference waits: not a decode rate"}
    barrier()
    events = {}
    if args.graph:
        # the timed region replays one captured hipGraph per frame in flight and step (one launch per frame instead of ~30), each on a
        # stream of its own; no events inside it (independent-frames mode only)
        gstreams = [lib.vvc355_stream_create() for _ in range(n_ff)]
        gexecs = []
        for f in range(n_ff):
            lib.vvc355_graph_begin(gstreams[f])
            for st in chains[f]:
                st.launch(gstreams[f])
            gexecs.append(lib.vvc355_graph_end(gstreams[f]))
        for f in range(n_ff):
            lib.vvc355_graph_launch(gexecs[f], gstreams[f])
        for f in range(n_ff):
            lib.vvc355_stream_sync(gstreams[f])
        barrier()
        t0 = time.perf_counter()
        for k in range(args.steps):
            for f in range(n_ff):
                lib.vvc355_graph_launch(gexecs[f], gstreams[f])
        for f in range(n_ff):
            lib.vvc355_stream_sync(gstreams[f])
        elapsed = time.perf_counter() - t0
        frames_per_step = n_ff
    else:
        t0 = time.perf_counter()
        for k in range(args.steps):
            run_step(events, k, only=dom_name)
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
    barrier()
    elapsed = sharding.max_over_ranks(dist, torch, world, elapsed, "cpu")

    # second figure: the same frames with every per-frame descriptor (job arrays, side tables, command lists) copied from pinned host
    # memory at the start of each frame — what a decoder that builds them on the host pays over PCIe (the coefficient levels, which
    # are generated on the device here, are not included: their size is reported).  Measured on the F independent frames.
    upload = None
    if gop and not args.graph and not args.no_upload:
        for key, fr_i in objs.items():
            lo_, hi_ = fr_i.arena.data_ptr(), fr_i.arena.data_ptr() + fr_i.arena.numel()
            pinned_of[key] = [(fr_i.arena[:fr_i.arena_used], fr_i.arena_host[:fr_i.arena_used])]          # one copy: the picture's arena
            pinned_of[key] += [(fr_i.torch_of[p_], torch.from_numpy(h_).pin_memory()) for p_, h_ in fr_i.host.items() if not lo_ <= p_ < hi_]
        up_bytes = sum(h_.nbytes for h_ in frame.host.values())
        k0 = gop_k[0] + (-gop_k[0]) % n_sets                      # continue the group sequence
        torch.cuda.synchronize()
        recorded.clear()
        for k in range(2):
            run_gop(k0 + k, upload=True)
        barrier()
        t1 = time.perf_counter()
        for k in range(2, 2 + args.steps):
            run_gop(k0 + k, upload=True)
        torch.cuda.synchronize()
        el_up = sharding.max_over_ranks(dist, torch, world, time.perf_counter() - t1, "cpu")
        # what the link gives for the same kind of copy alone (one picture's arena, pinned host -> device, back to back): the upload leg's ceiling
        a_dst, a_src = pinned_of[(0, 1)][0]
        for _ in range(2):
            a_dst.copy_(a_src, non_blocking=True)
        torch.cuda.synchronize()
        t_h = time.perf_counter()
        for _ in range(16):
            a_dst.copy_(a_src, non_blocking=True)
        torch.cuda.synchronize()
        h2d_gbps = 16 * a_src.numel() * a_src.element_size() / (time.perf_counter() - t_h) / 1e9
        upload = {"value": world * gop * args.steps / el_up, "unit": "frames/s", "descriptor_bytes_per_frame": int(up_bytes),
                  "copy_GBps_in_this_leg": world * gop * args.steps * up_bytes / el_up / 1e9 / world, "measured_pinned_h2d_GBps": h2d_gbps,
                  "ms_per_step": el_up / args.steps * 1e3, "frames_per_step": gop,
                  "what": "the same groups of pictures with every picture's per-frame tables, records and descriptor arrays copied from pinned host memory "
                          "first — on a copy stream, as soon as the picture object is free; the picture's own stream waits for the copy's event",
                  "not_included": "coefficient levels (int32 in the reference ABI, generated on the device here)"}
    elif args.with_upload and not args.graph:
        pinned = [[(frames[f].torch_of[p_], torch.from_numpy(h_).pin_memory()) for p_, h_ in frames[f].host.items()] for f in range(n_ff)]
        up_bytes = sum(h_.nbytes for h_ in frames[0].host.values())

        def run_step_upload():
            for f in range(n_ff):
                with torch.cuda.stream(streams[f % len(streams)]):
                    for dst_t, src_t in pinned[f]:
                        dst_t.copy_(src_t, non_blocking=True)
                run_frame(f)
        run_step_upload()
        barrier()
        t1 = time.perf_counter()
        for k in range(args.steps):
            run_step_upload()
        torch.cuda.synchronize()
        el_up = sharding.max_over_ranks(dist, torch, world, time.perf_counter() - t1, "cpu")
        upload = {"value": world * n_ff * args.steps / el_up, "unit": "frames/s", "descriptor_bytes_per_frame": int(up_bytes),
                  "ms_per_step": el_up / args.steps * 1e3, "frames_in_flight": n_ff,
                  "not_included": "coefficient levels (int32 in the reference ABI, generated on the device here)"}

    if rank == 0:
        stages = {}
        kernel_ms_total = sum(breakdown.values())
        for st in chain:
            gbs = st.algorithmic_bytes / (breakdown[st.name] * 1e-3) / 1e9
            stages[st.name] = {"kernel": st.kernel, "ms": breakdown[st.name], "algorithmic_bytes": st.algorithmic_bytes,
                               "GB/s": gbs, "frac_of_hbm_peak": gbs / HBM_PEAK_GBS, "share_of_frame_kernel_time": breakdown[st.name] / kernel_ms_total}
            if getattr(st, "extra", None):
                stages[st.name].update(st.extra)
                if "longest_dependency_chain_commands" in st.extra:
                    stages[st.name]["us_per_command_on_the_chain"] = breakdown[st.name] * 1e3 / max(1, st.extra["longest_dependency_chain_commands"])
        dom = next(st for st in chain if st.name == dom_name)         # the batched stage with the largest launch time
        in_region = float(np.mean([a.elapsed_time(b) for a, b in events[dom_name]])) if events.get(dom_name) else None
        recon_ms = breakdown.get("intra_recon_wavefront")
        if gop:
            work = (f"one hierarchical-B group of {gop} pictures of one stream per step (decode order {' '.join(str(p_) for p_, _l, _h in order)}; every picture's inter "
                    f"prediction reads the decoded output of its two reference pictures and starts behind their ALF stage; consecutive groups overlap; "
                    f"{n_sets} rotating sets of {gop} frame buffers)")
            par = f"one picture per HIP stream ({len(streams)} streams), event waits on the reference pictures"
        else:
            work = f"{n_ff} independent frame(s) in flight per step, one HIP stream each, no reference dependencies"
            par = f"{n_ff} frame(s) in flight per GPU, one HIP stream each"
        out = {
            "metric": "decoded frames/sec (4K/8K 10-bit VVC) per GPU",
            "parity": "device output == in-repo CPU oracle on this frame (`verified`: sampled CTUs per prediction / transform stage, whole picture "
                      "for the loop-filter stages); the oracle itself is unpinned (no FATE bitstreams or reference build in this environment)",
            "value": world * frames_per_step * args.steps / elapsed,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "frames_per_step": frames_per_step,
            "frame_latency_ms": frame_latency_ms,          # one frame alone, stage after stage (untimed pass): latency, not throughput
            "independent_frames": independent,             # the same frames without reference dependencies: a ceiling, not a decode rate
            "incl_descriptor_upload": upload,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8" if args.bd == 8 else "u16",
            "data": "synthetic",
            "config": {
                "workload": f"{args.width}x{args.height} {args.bd}-bit 4:2:0 random-access pictures = {frame.n_ctus} CTUs of 128x128 each; {work}; "
                            f"per picture {INTER_FRAC:.0%} inter CTUs: regular bi-prediction with DMVR + BDOF, {GPM_FRAC:.0%} of the blocks geometric partitions, {AFFINE_FRAC:.0%} of the CTUs affine + PROF, "
                            f"{CIIP_FRAC:.0%} of all CTUs combined inter / intra; {1 - INTER_FRAC:.0%} intra CTUs reconstructed in decoding order with LFNST / implicit MTS; "
                            f"{'uniform-noise' if args.noise else 'picture-like'} content; HBM-resident; "
                            f"stages per picture: {', '.join(st.name for st in chain)}",
                "not_yet_in_chain": MISSING + ([] if MC_TOOLS == 3 and not args.only and AFFINE_FRAC == 0.06 and INTER_FRAC == 0.8 and SAO_TABLES and ALF_TABLES and not DEBLOCK_JOBS and not args.graph and not args.noise else ["PROFILING RUN: --noise / --graph / --mc-tools / --only / --affine-frac / --inter-frac / --sao-jobs / --alf-jobs / --deblock-jobs change the workload; not the metric"]),
                "parallelism": f"{world} independent stream(s), one per GPU, no collective; {par}, GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')}; in-order pass: "
                               f"{concurrent_wgs or 192} persistent workgroups per picture in the timed region, 192 for the one-picture legs (`stages`, frame_latency_ms)",
                "frame_objects_built": len(objs) if gop else n_ff, "build_s": round(build_s, 1),
            },
            "roofline": {
                "stage": dom.name,
                "kernel": dom.kernel,
                "bound": "hbm",
                # duration: HIP events on the launch stream around the stage, one frame alone (the K-step pass right before the timed
                # region) = the kernels' own duration, which is what rocprofv3's per-kernel average of one frame in flight shows
                # (profiles/r03_bench_f1_kernel_stats.csv).  bytes: SURVEY.md 8(d)'s compulsory traffic (ALF: 4 B per sample).
                "achieved": dom.algorithmic_bytes / (breakdown[dom.name] * 1e-3) / 1e9,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": dom.algorithmic_bytes / (breakdown[dom.name] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "ms_per_launch": breakdown[dom.name],
                "algorithmic_bytes_per_launch": dom.algorithmic_bytes,
                "implementation_bytes_per_launch": getattr(dom, "implementation_bytes", None),
                "in_timed_region": None if in_region is None else {"ms_per_launch": in_region, "note": "event pair around the same stage while the other pictures share the device: "
                                                                                                    "includes the time its kernels wait for CUs"},
                # the longest kernel of a picture is not this stage but the in-order intra pass: a latency-bound dependency chain on a few
                # hundred waves (21 MB of traffic), which the other pictures' batched stages overlap
                "in_order_intra_pass": None if recon_ms is None else {"ms": recon_ms, "share_of_frame_kernel_time": recon_ms / kernel_ms_total,
                                                                      "note": "longest kernel of a picture; excluded from `stage` because it is bound by its dependency chain, not by bandwidth"},
                "note": "dominant = the batched stage with the largest launch time (stages / frame_latency_ms list every stage of one picture alone)",
                "traffic": recorded_traffic(ROOT, dom.name),      # L2 <-> fabric bytes per launch from separate rocprofv3 --pmc passes of the same command (profiles/pmc_traffic.json)
                "recorded": {"source": "profiles/pmc_traffic.json, profiles/mc_valu.json (rocprofv3 --pmc passes, see profiles/README.md)",
                             "valu": recorded_valu(ROOT, dom.name)},
            },
            "stages": stages,
        }
        # what a plain device-to-device copy reaches on this box (report the practical peak beside the vendor figure)
        n_copy = 1 << 28
        a_buf = torch.empty(n_copy, dtype=torch.uint8, device="cuda")
        b_buf = torch.empty_like(a_buf)
        b_buf.copy_(a_buf)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record()
        for _ in range(10):
            b_buf.copy_(a_buf)
        c1.record()
        torch.cuda.synchronize()
        out["roofline"]["measured_copy_GBps"] = 2 * n_copy * 10 / (c0.elapsed_time(c1) * 1e-3) / 1e9      # read + write
        del a_buf, b_buf
        if not args.no_verify and not args.only:
            out["verified"] = verify_step(lib, torch, frame, chain, args.verify_ctus)
        if not args.no_cpu_baseline and world == 1:       # a reported baseline, timed once: rank 0 of the single-GPU run
            out["cpu_baseline"] = cpu_baseline(ROOT, frame, args.cpu_seconds, args.cpu_workers)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
