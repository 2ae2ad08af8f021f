"""GPU parity of the fused prediction stage (one launch, many blocks, straight to pixels) vs the oracle run the way the
reference chains the slots per block: put[..] x2 + avg / w_avg (vvc_inter.c:253-296) or put_uni / put_uni_w (:222-251)."""
import numpy as np
import pytest

import inter_cases as ic
from conftest import P, rand_pixels
from ffvvc_amd import abi, batch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_pred_fused_batch(dev, orc, bd):
    rng = np.random.default_rng(0x5EED0700 + bd)
    luma_f, chroma_f = ic.tables(dev, "vvc355_")
    pw, ph, pad = 256, 160, 32
    refs = [rand_pixels(rng, (ph + 2 * pad, pw + 2 * pad), bd) for _ in range(2)]
    isz = refs[0].itemsize
    rs = refs[0].shape[1]
    want = np.full((ph, pw), 0x33, refs[0].dtype)
    n = 0
    jobs = []
    # tile the picture with blocks of assorted sizes, every (frac x frac) combination, all four modes
    y = 0
    while y + 16 <= ph:
        x = 0
        while x + 16 <= pw:
            w, h = int(rng.choice([2, 4, 8, 16])), int(rng.choice([2, 4, 6, 8, 12, 16]))
            chroma = int(rng.integers(0, 2))
            mode = int(rng.integers(0, 4))
            frac = int(rng.integers(0, 16))
            j = abi.PredJob()
            j.w, j.h, j.chroma, j.mode, j.frac = w, h, chroma, mode, frac
            tab, nph, ntap = (chroma_f, 32, 4) if chroma else (luma_f, 16, 8)
            filt = []
            for name in ("hf0", "vf0", "hf1", "vf1"):
                f = np.ascontiguousarray(tab[int(rng.integers(0, 3)), int(rng.integers(1, nph))])
                filt.append(f)
                for k in range(ntap):
                    getattr(j, name)[k] = int(f[k])
            j.denom = int(rng.integers(0, 8))
            j.w0, j.w1, j.o0, j.o1 = (int(v) for v in rng.integers(-128, 128, size=4))
            mv = rng.integers(-12, 13, size=(2, 2))
            offs = [int((y + mv[r][1] + pad) * rs + x + mv[r][0] + pad) for r in range(2)]
            # ---- oracle: the slot chain of the reference
            t = [np.zeros((16, 128), np.int16), np.zeros((16, 128), np.int16)]
            dst = np.zeros((h, w), refs[0].dtype)
            if mode < 2:
                for r in range(2):
                    orc.orc_put(bd, chroma, (frac >> (2 * r + 1)) & 1, (frac >> (2 * r)) & 1, P(t[r]), P(refs[r], offs[r]), rs * isz,
                                h, P(filt[2 * r]), P(filt[2 * r + 1]), w)
                if mode == 0:
                    orc.orc_avg(bd, P(dst), w * isz, P(t[0]), P(t[1]), w, h)
                else:
                    orc.orc_w_avg(bd, P(dst), w * isz, P(t[0]), P(t[1]), w, h, j.denom, j.w0, j.w1, j.o0, j.o1)
            elif mode == 2:
                orc.orc_put_uni(bd, chroma, (frac >> 1) & 1, frac & 1, P(dst), w * isz, P(refs[0], offs[0]), rs * isz, h, P(filt[0]), P(filt[1]), w)
            else:
                orc.orc_put_uni_w(bd, chroma, (frac >> 1) & 1, frac & 1, P(dst), w * isz, P(refs[0], offs[0]), rs * isz, h,
                                  j.denom, j.w0, j.o0, P(filt[0]), P(filt[1]), w)
            want[y:y + h, x:x + w] = dst
            jobs.append((j, x, y, offs))
            n += 1
            x += 16
        y += 16

    pitched = batch.to_pitched(np.full((ph, pw), 0x33, refs[0].dtype))
    pitch = pitched.shape[1] * isz
    d_dst = batch.DeviceBuffer.from_host(pitched)
    d_refs = [batch.DeviceBuffer.from_host(r) for r in refs]
    arr = (abi.PredJob * n)()
    for i, (j, x, y, offs) in enumerate(jobs):
        j.dst = d_dst.ptr + y * pitch + x * isz
        j.src0, j.src1 = d_refs[0].ptr + offs[0] * isz, d_refs[1].ptr + offs[1] * isz
        j.dst_stride, j.src0_stride, j.src1_stride = pitch, rs * isz, rs * isz
        arr[i] = j
    d_jobs = batch.jobs_to_device(arr)
    dev.vvc355_pred_fused_batch(None, bd, d_jobs.ptr, n)
    dev.vvc355_stream_sync(None)
    got = d_dst.to_host(pitched.dtype, pitched.shape)[:, :pw]
    bad = np.argwhere(got != want)
    assert len(bad) == 0, f"{len(bad)} samples differ, first at {bad[0].tolist()}: job {[(jj.w, jj.h, jj.chroma, jj.mode, jj.frac) for jj, x, y, o in jobs if x <= bad[0][1] < x + 16 and y <= bad[0][0] < y + 16]}"
    assert n > 100
