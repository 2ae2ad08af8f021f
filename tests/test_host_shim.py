"""The C host shim (ffvvc_amd/host/dsp_init_mi355.c): ff_vvc_dsp_init_mi355 fills the VVCDSPContext-shaped table."""
import ctypes
import os

import pytest

from conftest import ROOT

HOST = os.path.join(ROOT, "ffvvc_amd", "libvvc_mi355_host.so")
# pointers in the table: inter 3*56 + 11 + sad + 4 dmvr, intra 10, itx 3 + 441 + 1, lmcs 1, lf 6, sao 20, alf 5
TABLE_POINTERS = 3 * 56 + 16 + 10 + 445 + 1 + 6 + 20 + 5


def load():
    lib = ctypes.CDLL(HOST)
    lib.ff_vvc_dsp_init_mi355.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.vvc355_dsp_count_slots.argtypes = [ctypes.c_void_p]
    return lib


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_installer_fills_every_slot_it_owns(bd):
    lib = load()
    buf = (ctypes.c_void_p * TABLE_POINTERS)()
    lib.ff_vvc_dsp_init_mi355(buf, bd)
    filled = lib.vvc355_dsp_count_slots(buf)
    # not installed by the standalone shim: 3 context-taking intra slots, 2 edge_restore, and the itx combinations the
    # reference leaves NULL (vvcdsp_template.c:142-159 installs 264 of the 441 entries)
    n_itx = sum(1 for h in range(3) for v in range(3) for lw in range(7) for lh in range(7) if itx_exists(h, v, lw, lh))
    assert filled == TABLE_POINTERS - 3 - 2 - (441 - n_itx)


def itx_exists(trh, trv, lw, lh):
    if not lw and not lh:
        return False
    if not lh:
        return trv == 0 and (lw in (4, 5) or (lw == 6 and trh == 0))
    if not lw:
        return trh == 0 and (lh in (4, 5) or (lh == 6 and trv == 0))
    if trh and not 2 <= lw <= 5:
        return False
    if trv and not 2 <= lh <= 5:
        return False
    return True


@pytest.mark.gpu
@pytest.mark.parametrize("bd", [8, 10, 12])
def test_calls_through_the_table_match_direct_calls(bd):
    assert load().vvc355_dsp_table_selftest(bd) == 0


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_context_shim_completes_the_table(bd):
    """With ff_vvc_dsp_init_mi355_ctx (the in-tree half: slots that take VVCLocalContext / SAOParams) every slot the reference has
    is installed: only the itx combinations the reference itself leaves NULL stay empty."""
    lib = load()
    lib.ff_vvc_dsp_init_mi355_ctx.argtypes = [ctypes.c_void_p, ctypes.c_int]
    buf = (ctypes.c_void_p * TABLE_POINTERS)()
    lib.ff_vvc_dsp_init_mi355(buf, bd)
    lib.ff_vvc_dsp_init_mi355_ctx(buf, bd)
    n_itx = sum(1 for h in range(3) for v in range(3) for lw in range(7) for lh in range(7) if itx_exists(h, v, lw, lh))
    assert lib.vvc355_dsp_count_slots(buf) == TABLE_POINTERS - (441 - n_itx)


# index of the first pointer of each sub-table (member order of VVCDSPContext, vvcdsp.h:160-168)
INTRA0 = 3 * 56 + 16
SAO0 = INTRA0 + 10 + 445 + 1 + 6


@pytest.mark.gpu
@pytest.mark.parametrize("bd", [8, 10])
def test_context_taking_slots_through_the_table(orc, bd):
    """intra_pred, intra_cclm_pred, lmcs_scale_chroma and sao.edge_restore called through the table with a hand-built decoder
    context (include/vvc_mi355_ctx.h), host planes — against the oracle's flattened forms on the same block."""
    import numpy as np
    import ctx_mirror as cm
    from conftest import rand_pixels
    from ffvvc_amd import abi
    lib = load()
    lib.ff_vvc_dsp_init_mi355_ctx.argtypes = [ctypes.c_void_p, ctypes.c_int]
    tab = (ctypes.c_void_p * TABLE_POINTERS)()
    lib.ff_vvc_dsp_init_mi355(tab, bd)
    lib.ff_vvc_dsp_init_mi355_ctx(tab, bd)
    LC = ctypes.POINTER(cm.VVCLocalContext)
    cclm_fn = ctypes.CFUNCTYPE(None, LC, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int)(tab[INTRA0 + 0])
    lmcs_fn = ctypes.CFUNCTYPE(None, LC, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int)(tab[INTRA0 + 1])
    pred_fn = ctypes.CFUNCTYPE(None, LC, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int)(tab[INTRA0 + 2])
    restore_fn = [ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_ssize_t, ctypes.c_ssize_t, ctypes.POINTER(cm.SAOParams),
                                   ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p)(tab[SAO0 + 18 + v])
                  for v in range(2)]
    orc.orc_intra_pred_flat.argtypes = [ctypes.c_int, ctypes.c_void_p]
    orc.orc_intra_cclm_pred_flat.argtypes = [ctypes.c_int, ctypes.c_void_p]
    orc.orc_lmcs_scale_chroma_flat.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    host = cm.load_host()
    host.vvc355_ctx_flatten_intra_pred.argtypes = [LC] + [ctypes.c_int] * 5 + [ctypes.POINTER(abi.IntraJob)]
    host.vvc355_ctx_flatten_cclm.argtypes = [LC] + [ctypes.c_int] * 4 + [ctypes.POINTER(abi.CclmJob)]
    host.vvc355_ctx_flatten_lmcs_scale.argtypes = [LC, ctypes.c_int, ctypes.c_int, ctypes.POINTER(abi.LmcsScaleJob)]
    rng = np.random.default_rng(0x5EED1200 + bd)
    w, h, isz = 256, 128, (1 if bd == 8 else 2)
    planes = [rand_pixels(rng, (h, w), bd), rand_pixels(rng, (h // 2, w // 2), bd), rand_pixels(rng, (h // 2, w // 2), bd)]
    imf = np.zeros((h // 4) * (w // 4), np.uint8)
    fc = cm.VVCFrameContext()
    fc.width, fc.height, fc.bit_depth, fc.ctb_log2_size_y, fc.min_cb_log2_size_y, fc.min_cb_width = w, h, bd, 7, 2, w // 4
    for c in range(3):
        fc.hshift[c] = fc.vshift[c] = int(c > 0)
        fc.linesize[c] = planes[c].shape[1] * isz
    fc.imf = fc.imm = fc.imtf = imf.ctypes.data
    fc.lmcs.min_bin_idx, fc.lmcs.max_bin_idx = 1, 14
    for i in range(17):
        fc.lmcs.pivot[i] = i << (bd - 4)
    for i in range(16):
        fc.lmcs.chroma_scale_coeff[i] = 1800 + 37 * i
    lc = cm.VVCLocalContext()
    lc.fc = ctypes.pointer(fc)
    lc.ctb_left_flag, lc.ctb_up_flag = 1, 0            # the CTU at (128, 0): a left neighbour, nothing above (picture top)
    lc.end_of_tiles_x = w
    # the CTU at (128, 0) with its left half reconstructed: areas of 32x32 coding units in decoding order
    for ch in range(2):
        n = 0
        for (ax, ay) in ((128, 0), (160, 0), (192, 0), (128, 32), (160, 32), (128, 64), (160, 64)):
            a = lc.ras[ch][n]
            a.x, a.y, a.w, a.h = ax >> ch, ay >> ch, 32 >> ch, 32 >> ch
            n += 1
        lc.num_ras[ch] = n
    cu = cm.CodingUnit()
    cu.x0, cu.y0, cu.cb_width, cu.cb_height = 192, 32, 32, 16
    cu.intra_pred_mode_y, cu.intra_pred_mode_c = 5, 82          # 5 on a 32x16 block is remapped by the wide-angle rule
    lc.cu = ctypes.pointer(cu)
    lc.na.cand_up_left = 1

    def both(run_dev, run_orc):
        got, want = [p.copy() for p in planes], [p.copy() for p in planes]
        for c in range(3):
            fc.data[c] = got[c].ctypes.data
        run_dev()
        for c in range(3):
            fc.data[c] = want[c].ctypes.data
        run_orc()
        for c in range(3):
            assert np.array_equal(got[c], want[c]), c
        return any(not np.array_equal(want[c], planes[c]) for c in range(3))

    def orc_pred():
        j = abi.IntraJob()
        host.vvc355_ctx_flatten_intra_pred(ctypes.byref(lc), 192, 32, 32, 16, 0, ctypes.byref(j))
        # left: the areas at x = 160 from y = 32 down to the CTU's end (64 rows); top: the one area at (192, 0), 32 wide
        assert j.mode == 70 and j.left_avail == 64 and j.top_avail == 32
        orc.orc_intra_pred_flat(bd, ctypes.addressof(j))
    try:
        changed = both(lambda: pred_fn(ctypes.byref(lc), 192, 32, 32, 16, 0), orc_pred)
    except AssertionError:
        # the expectations in orc_pred are about this hand-built context; report them plainly
        j = abi.IntraJob()
        host.vvc355_ctx_flatten_intra_pred(ctypes.byref(lc), 192, 32, 32, 16, 0, ctypes.byref(j))
        raise AssertionError((j.mode, j.left_avail, j.top_avail))
    assert changed

    def orc_cclm():
        j = abi.CclmJob()
        host.vvc355_ctx_flatten_cclm(ctypes.byref(lc), 192, 32, 32, 16, ctypes.byref(j))
        orc.orc_intra_cclm_pred_flat(bd, ctypes.addressof(j))
    assert both(lambda: cclm_fn(ctypes.byref(lc), 192, 32, 32, 16), orc_cclm)

    coeff = rng.integers(-(1 << bd), 1 << bd, size=16 * 8).astype(np.int32)
    d_dev, d_orc = np.zeros_like(coeff), np.zeros_like(coeff)
    for c in range(3):
        fc.data[c] = planes[c].ctypes.data
    lmcs_fn(ctypes.byref(lc), d_dev.ctypes.data, coeff.ctypes.data, 16, 8, 192, 32)
    j = abi.LmcsScaleJob()
    host.vvc355_ctx_flatten_lmcs_scale(ctypes.byref(lc), 192, 32, ctypes.byref(j))
    orc.orc_lmcs_scale_chroma_flat(bd, ctypes.addressof(j), d_orc.ctypes.data, coeff.ctypes.data, 16, 8)
    assert np.array_equal(d_dev, d_orc) and np.any(d_dev != coeff)

    # sao.edge_restore[v] with SAOParams
    sao = cm.SAOParams()
    for c in range(3):
        sao.eo_class[c] = c
        for k in range(1, 5):
            sao.offset_val[c][k] = int(rng.integers(-7, 8))
    src = rand_pixels(rng, (40, 48), bd)
    borders = (ctypes.c_int * 4)(1, 0, 0, 1)
    ve, he, de = (ctypes.c_uint8 * 2)(0, 1), (ctypes.c_uint8 * 2)(1, 0), (ctypes.c_uint8 * 4)(0, 1, 1, 0)
    for v in range(2):
        a, b = rand_pixels(rng, (32, 32), bd), None
        b = a.copy()
        restore_fn[v](a.ctypes.data, src.ctypes.data + (4 * 48 + 8) * isz, 32 * isz, 48 * isz, ctypes.byref(sao), borders, 32, 32, 1, ve, he, de)
        orc.orc_sao_edge_restore(bd, v, b.ctypes.data, src.ctypes.data + (4 * 48 + 8) * isz, 32 * isz, 48 * isz, ctypes.addressof(sao.offset_val[1]), sao.eo_class[1],
                                 ctypes.addressof(borders), 32, 32, ctypes.addressof(ve), ctypes.addressof(he), ctypes.addressof(de))
        assert np.array_equal(a, b), v


@pytest.mark.gpu
def test_slots_called_from_four_threads_at_once(dev, orc):
    """The reference calls the table from thread_count executor workers at once (libavutil/executor.c:92-110, vvc_thread.c:647-654):
    every slot must be re-entrant.  Four host threads hammer different slots with their own data; every result is checked."""
    import threading
    import numpy as np
    from conftest import P, rand_pixels
    errors = []

    def worker(t):
        try:
            rng = np.random.default_rng(0x5EED1300 + t)
            bd = (8, 10, 12, 10)[t]
            isz = 1 if bd == 8 else 2
            for it in range(25):
                kind = (t + it) % 3
                if kind == 0:       # ALF luma
                    w = h = 32
                    src = rand_pixels(rng, (h + 16, w + 16), bd)
                    n = (w // 4) * (h // 4)
                    coeff = rng.integers(-128, 128, size=(n, 12)).astype(np.int16)
                    clip = np.full((n, 12), 1 << bd, np.int16)
                    out = []
                    for lib, pre in ((orc, "orc_"), (dev, "vvc355_")):
                        dst = np.zeros((h, w), src.dtype)
                        getattr(lib, pre + "alf_filter_luma")(bd, P(dst), w * isz, P(src, 8 * src.shape[1] + 8), src.shape[1] * isz, w, h, P(coeff), P(clip), h - 4)
                        out.append(dst)
                elif kind == 1:     # 8-tap hv put
                    src = rand_pixels(rng, (40, 64), bd)
                    hf = np.array([-1, 4, -11, 40, 40, -11, 4, -1], np.int8)
                    vf = np.array([0, 1, -3, 63, 4, -2, 1, 0], np.int8)
                    out = []
                    for lib, pre in ((orc, "orc_"), (dev, "vvc355_")):
                        dst = np.zeros((16, 128), np.int16)
                        getattr(lib, pre + "put")(bd, 0, 1, 1, P(dst), P(src, 8 * 64 + 8), 64 * isz, 16, P(hf), P(vf), 32)
                        out.append(dst)
                else:               # inverse transform DST7 x DCT2 16x8
                    co = np.zeros((8, 16), np.int32)
                    co[:3, :5] = rng.integers(-2000, 2000, size=(3, 5))
                    out = []
                    for lib, pre in ((orc, "orc_"), (dev, "vvc355_")):
                        c2 = co.copy()
                        getattr(lib, pre + "itx")(1, 0, 4, 3, P(c2), 5, 3, 15, bd)
                        out.append(c2)
                if not np.array_equal(out[0], out[1]):
                    errors.append((t, it, kind))
        except Exception as e:          # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:5]


@pytest.mark.gpu
def test_error_policy_records_instead_of_aborting(dev):
    """vvc355_set_error_policy(1): a failing HIP call inside an entry is recorded, not fatal (freeing an address that is no
    allocation: rejected by the runtime without touching the device); the default policy is restored afterwards."""
    import ctypes
    dev.vvc355_clear_error()
    assert dev.vvc355_last_error() == 0
    dev.vvc355_set_error_policy(1)
    try:
        dev.vvc355_free(ctypes.c_void_p(0x1234560))
        assert dev.vvc355_last_error() != 0
        assert b"failed" in ctypes.string_at(dev.vvc355_last_error_string())
        dev.vvc355_clear_error()
        assert dev.vvc355_last_error() == 0
    finally:
        dev.vvc355_set_error_policy(0)
