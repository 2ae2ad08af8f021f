"""vvc355_tab_fill_pass: the decoder's per-unit side tables (MvField, transform / coding block positions and sizes, coded / pcm / joint
flags, MergeSubblockFlag, InterAffineFlag) written on the device from per-unit records, against (i) the oracle's restatement of the
reference's table setters and (ii) the tables the test generator filled unit by unit the way the reference's parser does
(set_cb_pos / set_cb_tab, set_tb_pos / set_tb_tab, ff_vvc_set_mvf: vvc_ctu.c:41-140, :1144-1250)."""
import ctypes

import numpy as np
import pytest

import bs_cases
from ffvvc_amd import abi, batch


def _oracle_tables(orc, t, cu, tu, mv):
    orc.orc_tab_fill_pass.argtypes = [ctypes.POINTER(abi.TabFill)]
    orc.orc_tab_fill_pass.restype = None
    out = {name: np.full_like(getattr(t, name), 0x5A if getattr(t, name).dtype != np.int32 else 0x5A5A5A5A) for name in t.FILLED if name != "mvf"}
    out["mvf"] = np.zeros_like(t.mvf)
    out["mvf"].view(np.uint8)[:] = 0x5A
    f = t.fill_frame(cu.ctypes.data, tu.ctypes.data, mv.ctypes.data, (len(cu), len(tu), len(mv)), lambda n: out[n].ctypes.data)
    orc.orc_tab_fill_pass(ctypes.byref(f))
    return out


@pytest.mark.parametrize("fmt,ctb_log2", [((1, 1), 7), ((1, 0), 6), ((0, 0), 5)])
def test_records_reproduce_the_parsers_tables(orc, fmt, ctb_log2):
    """CPU: the records carry everything the tables hold (oracle restatement == unit-by-unit fill)."""
    t = bs_cases.BsTables(np.random.default_rng(0x5EED0F00 + ctb_log2), 328, 200, ctb_log2, n_slices=2, hs=fmt[0], vs=fmt[1])
    cu, tu, mv = t.records()
    got = _oracle_tables(orc, t, cu, tu, mv)
    for name in t.FILLED:
        assert np.array_equal(got[name].view(np.uint8), getattr(t, name).view(np.uint8)), name
    assert 24 * t.mvf.size + 60 * t.mvf.size > 6 * (cu.nbytes + tu.nbytes + mv.nbytes)        # the records are several times smaller than the tables


@pytest.mark.gpu
@pytest.mark.parametrize("fmt,ctb_log2,w,h", [((1, 1), 7, 1480, 840), ((1, 0), 6, 328, 200), ((0, 0), 5, 136, 104)])
def test_tab_fill_pass(dev, orc, fmt, ctb_log2, w, h):
    t = bs_cases.BsTables(np.random.default_rng(0x5EED0F10 + ctb_log2), w, h, ctb_log2, hs=fmt[0], vs=fmt[1])
    n_ctb = t.cw * t.ch
    grouped = [t.group_per_ctu(r, ctb_log2, t.cw, n_ctb) for r in t.records()]
    (cu, tu, mv), firsts = [g[0] for g in grouped], [g[1] for g in grouped]
    want = _oracle_tables(orc, t, cu, tu, mv)
    d_rec = [batch.DeviceBuffer.from_host(a.view(np.uint8)) for a in (cu, tu, mv)]
    d_first = [batch.DeviceBuffer.from_host(a) for a in firsts]
    d_tab = {}
    for name in t.FILLED:
        init = np.zeros_like(getattr(t, name))
        init.view(np.uint8)[:] = 0x5A
        d_tab[name] = batch.DeviceBuffer.from_host(init.view(np.uint8))
    f = t.fill_frame(d_rec[0].ptr, d_rec[1].ptr, d_rec[2].ptr, (len(cu), len(tu), len(mv)), lambda n: d_tab[n].ptr, tuple(d.ptr for d in d_first))
    d_f = batch.DeviceBuffer.from_host(np.frombuffer(bytes(f), np.uint8))
    dev.vvc355_tab_fill_pass(None, d_f.ptr, ctypes.addressof(f))
    dev.vvc355_stream_sync(None)
    for name in t.FILLED:
        ref = getattr(t, name)
        got = d_tab[name].to_host(np.uint8, (ref.nbytes,))
        assert np.array_equal(got, want[name].view(np.uint8).reshape(-1)), f"table {name} differs from the oracle"
        assert np.array_equal(got, ref.view(np.uint8).reshape(-1)), f"table {name} differs from the unit-by-unit fill"
