"""GPU parity: LMCS, SAO and deblocking slots through the C ABI vs the CPU oracle, bit-exact.
Input distributions follow tests/checkasm/vvc_sao.c:51-66,76,113 (offsets < 2^(bd-5), band class 0..31, eo 0..3) and, for
deblocking (which has no checkasm test in the reference), beta'/tc' of Table 43 for QP 22..50 on smooth-plus-step content
so that every decision branch (none / weak / strong / long-tap) is reached."""
import ctypes

import numpy as np
import pytest

from conftest import P, px_dtype, rand_pixels

pytestmark = pytest.mark.gpu

TC = [0] * 18 + [3, 4, 4, 4, 4, 5, 5, 5, 5, 7, 7, 8, 9, 10, 10, 11, 13, 14, 15, 17, 19, 21, 24, 25, 29, 33, 36, 41, 45, 51,
                 57, 64, 71, 80, 89, 100, 112, 125, 141, 157, 177, 198, 222, 250, 280, 314, 352, 395]
BETA = [0] * 16 + list(range(6, 19)) + list(range(20, 90, 2))


def both(orc, dev, fn):
    return fn(orc, "orc_"), fn(dev, "vvc355_")


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_lmcs(dev, orc, bd):
    rng = np.random.default_rng(0x5EED0300 + bd)
    for (w, h) in [(4, 4), (128, 128), (100, 37), (8, 64)]:
        img = rand_pixels(rng, (h + 2, w + 8), bd)
        lut = rand_pixels(rng, (1 << bd,), bd)

        def run(lib, pre):
            d = img.copy()
            getattr(lib, pre + "lmcs_filter")(bd, P(d, d.shape[1] + 4), d.shape[1] * d.itemsize, w, h, P(lut))
            return d

        a, b = both(orc, dev, run)
        assert np.array_equal(a, b)


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_lmcs_batch_rectangles(dev, orc, bd):
    """One launch, one LUT per job, rectangles at aligned and unaligned positions of a pitched plane (the vector path
    and the per-sample path of lmcs_kernel)."""
    from ffvvc_amd import abi, batch
    rng = np.random.default_rng(0x5EED0305 + bd)
    pic = rand_pixels(rng, (200, 400), bd)
    isz = pic.itemsize
    want = pic.copy()
    pitched = batch.to_pitched(pic)
    pitch = pitched.shape[1] * isz
    d_pic = batch.DeviceBuffer.from_host(pitched)
    rects = [(0, 0, 128, 128), (128, 0, 128, 64), (256, 0, 100, 37), (131, 70, 61, 50), (0, 130, 7, 70), (16, 130, 40, 1), (64, 136, 129, 64)]
    luts = [rand_pixels(rng, (1 << bd,), bd) for _ in rects]
    d_luts = [batch.DeviceBuffer.from_host(lut) for lut in luts]
    arr = (abi.BlendJob * len(rects))()
    for i, (x, y, w, h) in enumerate(rects):
        blk = np.ascontiguousarray(want[y:y + h, x:x + w])
        orc.orc_lmcs_filter(bd, P(blk), w * isz, w, h, P(luts[i]))
        want[y:y + h, x:x + w] = blk
        arr[i].dst, arr[i].dst_stride, arr[i].src0, arr[i].w, arr[i].h = d_pic.ptr + y * pitch + x * isz, pitch, d_luts[i].ptr, w, h
    d_jobs = batch.jobs_to_device(arr)
    dev.vvc355_lmcs_batch(None, bd, d_jobs.ptr, len(rects), 129, 128)
    dev.vvc355_stream_sync(None)
    got = d_pic.to_host(pitched.dtype, pitched.shape)[:, :pic.shape[1]]
    assert np.array_equal(got, want)


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_sao(dev, orc, bd):
    rng = np.random.default_rng(0x5EED0310 + bd)
    ss = (2 * 128 + 64) // (1 if bd == 8 else 2)          # implicit edge source stride in pixels
    for (w, h) in [(8, 8), (16, 12), (48, 48), (64, 64), (128, 128), (120, 70), (4, 128)]:
        offs = np.concatenate([[0], rng.integers(0, 1 << (bd - 5), size=4) * rng.choice([-1, 1], size=4)]).astype(np.int16)
        src = rand_pixels(rng, (h + 4, ss), bd)
        if rng.integers(0, 2):                              # plateaus so that "equal" comparisons occur
            src = (src >> 3 << 3).astype(src.dtype)
        left_class = int(rng.integers(0, 32))
        for eo in range(4):
            borders = rng.integers(0, 2, size=4).astype(np.int32)
            ve, he, de = (rng.integers(0, 2, size=n).astype(np.uint8) for n in (2, 2, 4))

            def run(lib, pre):
                ps = src.itemsize
                d0 = np.full((h + 2, w + 8), 0x55, src.dtype)
                getattr(lib, pre + "sao_band_filter")(bd, P(d0, d0.shape[1] + 4), P(src, 2 * ss + 8), d0.shape[1] * ps, ss * ps, P(offs), left_class, w, h)
                d1 = np.full((h + 2, w + 8), 0x55, src.dtype)
                getattr(lib, pre + "sao_edge_filter")(bd, P(d1, d1.shape[1] + 4), P(src, 2 * ss + 8), d1.shape[1] * ps, P(offs), eo, w, h)
                outs = [d0, d1]
                for variant in (0, 1):
                    d2 = d1.copy()
                    getattr(lib, pre + "sao_edge_restore")(bd, variant, P(d2, d2.shape[1] + 4), P(src, 2 * ss + 8), d2.shape[1] * ps, ss * ps,
                                                           P(offs), eo, P(borders), w, h, P(ve), P(he), P(de))
                    outs.append(d2)
                return outs

            a, b = both(orc, dev, run)
            for i, (x, y) in enumerate(zip(a, b)):
                assert np.array_equal(x, y), f"sao output {i} bd={bd} {w}x{h} eo={eo}"


def edge_picture(rng, bd, smooth):
    """24x24 picture around an edge at (12, 12): smooth ramps with a step across the edge + a little noise."""
    mx = (1 << bd) - 1
    y, x = np.mgrid[0:24, 0:24]
    base = int(rng.integers(mx // 4, 3 * mx // 4))
    step = int(rng.integers(0, 1 << (bd - 4)))
    img = base + (x >= 12) * step + (y >= 12) * step + (x * int(rng.integers(-2, 3))) + (y * int(rng.integers(-2, 3)))
    noise = 0 if smooth else rng.integers(-(1 << (bd - 5)), (1 << (bd - 5)) + 1, size=img.shape)
    return np.clip(img + noise, 0, mx).astype(px_dtype(bd))


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_deblock(dev, orc, bd):
    rng = np.random.default_rng(0x5EED0320 + bd)
    hits = 0
    for it in range(400):
        img = edge_picture(rng, bd, smooth=bool(it & 1))
        chroma = int(rng.integers(0, 2))
        dirn = int(rng.integers(0, 2))
        flag = int(rng.integers(0, 2))
        qp = rng.integers(22, 51, size=4)
        beta = np.array([BETA[q] for q in qp], np.int32)
        tc = np.array([TC[min(q + 2, 65)] for q in qp], np.int32)
        if rng.integers(0, 8) == 0:
            tc[int(rng.integers(0, 4))] = 0
        no_p = (rng.integers(0, 8, size=4) == 0).astype(np.uint8)
        no_q = (rng.integers(0, 8, size=4) == 0).astype(np.uint8)
        if chroma:
            lp = rng.choice([0, 1, 3], size=4).astype(np.uint8)
            lq = rng.choice([0, 1, 3], size=4).astype(np.uint8)
        else:
            lp = rng.choice([1, 2, 3, 5, 7], size=4).astype(np.uint8)
            lq = rng.choice([1, 2, 3, 5, 7], size=4).astype(np.uint8)
        off = 12 * 24 + 12 if dirn == 0 else 8 * 24 + 12
        if dirn == 0:
            off = 12 * 24 + 8

        def run(lib, pre):
            d = img.copy()
            fn = getattr(lib, pre + ("lf_filter_chroma" if chroma else "lf_filter_luma"))
            fn(bd, dirn, P(d, off), 24 * d.itemsize, P(beta), P(tc), P(no_p), P(no_q), P(lp), P(lq), flag)
            lvl = getattr(lib, pre + "lf_ladf_level")(bd, dirn, P(d, off), 24 * d.itemsize)
            return d, lvl

        (a, la), (b, lb) = both(orc, dev, run)
        assert np.array_equal(a, b), f"deblock it={it} bd={bd} chroma={chroma} dir={dirn} flag={flag} lp={lp} lq={lq}"
        assert la == lb
        hits += int(not np.array_equal(a, img))
    assert hits > 100          # the generator really exercises the filters


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_deblock_frame(dev, orc, bd):
    """All edges of one direction of a picture in ONE launch (vvc355_deblock_batch) vs the oracle filtering them one after the
    other: the kernel may only write the samples a filter changes, because neighbouring edges are filtered concurrently.
    Edge spacing follows what the standard allows: 8 samples apart with filter lengths <= 3, 32 apart for the 5 / 7 tap sides."""
    import bipred_cases as bc
    from ffvvc_amd import abi, batch
    rng = np.random.default_rng(0x5EED0330 + bd)
    pw, ph = 256, 128
    isz = 1 if bd == 8 else 2
    for (dirn, spacing, lens) in [(1, 8, [1, 2, 3]), (0, 8, [1, 2, 3]), (1, 32, [3, 5, 7]), (0, 32, [3, 5, 7])]:
        base = bc.smooth_picture(rng, ph, pw, bd, scale=32).astype(np.int64)
        # blocking artefacts: a small random offset per 8x8 block
        offs = rng.integers(-(1 << (bd - 6)), (1 << (bd - 6)) + 1, size=(ph // 8, pw // 8))
        pic = np.clip(base + np.kron(offs, np.ones((8, 8), np.int64)), 0, (1 << bd) - 1).astype(np.uint8 if bd == 8 else np.uint16)
        want = pic.copy()
        pitched = batch.to_pitched(pic)
        pitch = pitched.shape[1] * isz
        d_pic = batch.DeviceBuffer.from_host(pitched)
        jobs = []
        along, across = (ph, pw) if dirn == 1 else (pw, ph)          # dir 1: vertical edges (filter along rows)
        for e in range(spacing, across, spacing):
            for s in range(0, along, 8):
                qp = rng.integers(22, 51, size=4)
                j = abi.DeblockJob()
                for k in range(4):
                    j.beta[k], j.tc[k] = BETA[qp[k]], TC[min(qp[k] + 2, 65)]
                    j.no_p[k], j.no_q[k] = int(rng.integers(0, 8) == 0), int(rng.integers(0, 8) == 0)
                    j.max_len_p[k], j.max_len_q[k] = int(rng.choice(lens)), int(rng.choice(lens))
                j.dir, j.chroma, j.flag = dirn, 0, 0
                x, y = (e, s) if dirn == 1 else (s, e)
                jobs.append((j, x, y))
        for (j, x, y) in jobs:
            orc.orc_lf_filter_luma(bd, dirn, P(want, y * pw + x), pw * isz, ctypes.addressof(j) + abi.DeblockJob.beta.offset,
                                   ctypes.addressof(j) + abi.DeblockJob.tc.offset, ctypes.addressof(j) + abi.DeblockJob.no_p.offset,
                                   ctypes.addressof(j) + abi.DeblockJob.no_q.offset, ctypes.addressof(j) + abi.DeblockJob.max_len_p.offset,
                                   ctypes.addressof(j) + abi.DeblockJob.max_len_q.offset, 0)
        arr = (abi.DeblockJob * len(jobs))()
        for i, (j, x, y) in enumerate(jobs):
            j.pix, j.stride = d_pic.ptr + y * pitch + x * isz, pitch
            arr[i] = j
        d_jobs = batch.jobs_to_device(arr)
        dev.vvc355_deblock_batch(None, bd, d_jobs.ptr, len(jobs))
        dev.vvc355_stream_sync(None)
        got = d_pic.to_host(pitched.dtype, pitched.shape)[:, :pw]
        bad = np.argwhere(got != want)
        assert len(bad) == 0, f"dir={dirn} spacing={spacing} bd={bd}: {len(bad)} samples differ, first at {bad[0].tolist()}"
        assert np.count_nonzero(want != pic) > 500          # the filters really fired
