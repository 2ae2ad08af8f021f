#!/usr/bin/env python3
"""Single-thread rates of the CPU oracle's restatements on the inputs BASELINE.md section 2 quotes for the reference's C path
(measured there in this container): the cross-check BASELINE.md section 3 asks for.  Prints one JSON object."""
import ctypes
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
I, S, V = ctypes.c_int, ctypes.c_ssize_t, ctypes.c_void_p
SIG = {
    "orc_put": [I, I, I, I, V, V, S, I, V, V, I],
    "orc_put_uni": [I, I, I, I, V, S, V, S, I, V, V, I],
    "orc_avg": [I, V, S, V, V, I, I],
    "orc_w_avg": [I, V, S, V, V, I, I, I, I, I, I, I],
    "orc_dmvr": [I, I, I, V, V, S, I, S, S, I],
    "orc_sad": [V, V, I, I, I, I],
    "orc_apply_bdof": [I, V, S, V, V, I, I],
    "orc_itx": [I, I, I, I, V, S, S, S, S],
    "orc_add_residual": [I, V, V, I, I, S],
    "orc_pred_planar": [I, V, V, V, I, I, S],
    "orc_pred_dc": [I, V, V, V, I, I, S],
    "orc_pred_angular_v": [I, V, V, V, I, I, S, I, I, I, I, I],
    "orc_pred_angular_h": [I, V, V, V, I, I, S, I, I, I, I, I],
    "orc_alf_classify": [I, V, V, V, S, I, I, I, V],
    "orc_alf_filter_luma": [I, V, S, V, S, I, I, V, V, I],
    "orc_alf_filter_chroma": [I, V, S, V, S, I, I, V, V, I],
    "orc_alf_filter_cc": [I, V, S, V, S, I, I, I, I, V, I],
    "orc_sao_band_filter": [I, V, V, S, S, V, I, I, I],
    "orc_sao_edge_filter": [I, V, V, S, V, I, I, I],
    "orc_lf_filter_luma": [I, I, V, S, V, V, V, V, V, V, I],
    "orc_lmcs_filter": [I, V, S, I, I, V],
}


_ADDR = {}


def P(a, off=0):
    """Address of element `off`; cached, because numpy's .ctypes accessor costs about a microsecond per use."""
    k = id(a)
    if k not in _ADDR:
        assert a.flags["C_CONTIGUOUS"]
        _ADDR[k] = (a, a.ctypes.data)
    return _ADDR[k][1] + off * a.itemsize


def rate(fn, px, min_s=0.25):
    fn()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < min_s:
        fn()
        n += 1
    dt = (time.perf_counter() - t0) / n
    return {"Mpx/s": round(px / dt / 1e6, 1), "us/call": round(dt * 1e6, 2)}


def main():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    orc = ctypes.CDLL(os.path.join(ROOT, "oracle", "liborc.so"))
    for k, a in SIG.items():
        getattr(orc, k).argtypes = a
    bd, rng = 10, np.random.default_rng(1)
    src = rng.integers(0, 1024, size=(160, 160)).astype(np.uint16)          # stride 320 bytes
    s = P(src, 8 * 160 + 8)
    hf = np.array([-1, 4, -11, 40, 40, -11, 4, -1], np.int8)
    vf = np.array([0, 1, -3, 63, 4, -2, 1, 0], np.int8)
    cf = np.array([-4, 36, 36, -4], np.int8)
    t0, t1 = (rng.integers(0, 1 << 14, size=(140, 128)).astype(np.int16) for _ in range(2))
    dst = np.zeros((136, 128), np.uint16)                                    # stride 256 bytes
    out = {}
    out["MC luma 8-tap hv put 64x64"] = rate(lambda: orc.orc_put(bd, 0, 1, 1, P(t0), s, 320, 64, P(hf), P(vf), 64), 4096)
    out["MC luma 8-tap hv put 16x16"] = rate(lambda: orc.orc_put(bd, 0, 1, 1, P(t0), s, 320, 16, P(hf), P(vf), 16), 256)
    out["MC luma 8-tap h put 64x64"] = rate(lambda: orc.orc_put(bd, 0, 0, 1, P(t0), s, 320, 64, P(hf), P(vf), 64), 4096)
    out["MC luma hv put_uni 64x64"] = rate(lambda: orc.orc_put_uni(bd, 0, 1, 1, P(dst), 256, s, 320, 64, P(hf), P(vf), 64), 4096)
    out["MC chroma 4-tap hv put 32x32"] = rate(lambda: orc.orc_put(bd, 1, 1, 1, P(t0), s, 320, 32, P(cf), P(cf), 32), 1024)
    out["avg 64x64"] = rate(lambda: orc.orc_avg(bd, P(dst), 256, P(t0), P(t1), 64, 64), 4096)
    out["w_avg 64x64"] = rate(lambda: orc.orc_w_avg(bd, P(dst), 256, P(t0), P(t1), 64, 64, 3, 5, 3, 1, -1), 4096)
    out["DMVR bilinear hv 20x20"] = rate(lambda: orc.orc_dmvr(bd, 1, 1, P(t0), s, 320, 20, 5, 9, 20), 400)
    out["DMVR sad 16x16"] = rate(lambda: orc.orc_sad(P(t0, 4 * 128 + 4), P(t1, 4 * 128 + 4), 1, 2, 16, 16), 256)
    out["BDOF apply_bdof 16x16"] = rate(lambda: orc.orc_apply_bdof(bd, P(dst), 256, P(t0, 2 * 128 + 2), P(t1, 2 * 128 + 2), 16, 16), 256)
    for (n, trh, trv, nz, name) in ((8, 0, 0, 8, "itx DCT2^2 8x8"), (32, 0, 0, 32, "itx DCT2^2 32x32"), (64, 0, 0, 32, "itx DCT2^2 64x64 (nz 32)"),
                                    (16, 1, 1, 16, "itx DST7^2 16x16"), (32, 1, 1, 16, "itx DST7^2 32x32 (nz 16)")):
        co = rng.integers(-512, 512, size=(n, n)).astype(np.int32)
        lg = int(np.log2(n))
        work = np.empty_like(co)

        def one(co=co, work=work, lg=lg, trh=trh, trv=trv, nz=nz):
            work[:] = co
            orc.orc_itx(trh, trv, lg, lg, P(work), nz, nz, 15, bd)
        out[name] = rate(one, n * n)
    res = rng.integers(-64, 64, size=1024).astype(np.int32)
    out["add_residual 32x32"] = rate(lambda: orc.orc_add_residual(bd, P(dst), P(res), 32, 32, 256), 1024)
    edge = rng.integers(0, 1024, size=512).astype(np.uint16)
    out["intra planar 32x32"] = rate(lambda: orc.orc_pred_planar(bd, P(dst), P(edge, 100), P(edge, 300), 32, 32, 128), 1024)
    out["intra DC 32x32"] = rate(lambda: orc.orc_pred_dc(bd, P(dst), P(edge, 100), P(edge, 300), 32, 32, 128), 1024)
    out["intra angular-V mode 58 32x32"] = rate(lambda: orc.orc_pred_angular_v(bd, P(dst), P(edge, 100), P(edge, 300), 32, 32, 128, 0, 58, 0, 1, 1), 1024)
    out["intra angular-H mode 10 32x32"] = rate(lambda: orc.orc_pred_angular_h(bd, P(dst), P(edge, 100), P(edge, 300), 32, 32, 128, 0, 10, 0, 1, 1), 1024)
    n4 = 32 * 32
    cls, tr = np.zeros(n4, np.int32), np.zeros(n4, np.int32)
    grad = np.zeros(80 * 80 * 4, np.int32)
    coeff, clip = rng.integers(-128, 128, size=(n4, 12)).astype(np.int16), np.full((n4, 12), 1024, np.int16)
    out["ALF classify 128x128"] = rate(lambda: orc.orc_alf_classify(bd, P(cls), P(tr), s, 320, 128, 128, 124, P(grad)), 16384)
    out["ALF luma filter 128x128"] = rate(lambda: orc.orc_alf_filter_luma(bd, P(dst), 256, s, 320, 128, 128, P(coeff), P(clip), 124), 16384)
    cc, ccl = rng.integers(-64, 64, size=6).astype(np.int16), np.full(6, 1024, np.int16)
    out["ALF chroma 64x64"] = rate(lambda: orc.orc_alf_filter_chroma(bd, P(dst), 256, s, 320, 64, 64, P(cc), P(ccl), 62), 4096)
    c7 = rng.integers(-32, 32, size=7).astype(np.int16)
    out["CC-ALF 64x64"] = rate(lambda: orc.orc_alf_filter_cc(bd, P(dst), 256, s, 320, 64, 64, 1, 1, P(c7), 124), 4096)
    offs = np.array([0, 3, -2, 1, -4], np.int16)
    sao_src = rng.integers(0, 1024, size=(136, 160)).astype(np.uint16)       # the edge filter's implicit source stride (320 bytes)
    out["SAO band 128x128"] = rate(lambda: orc.orc_sao_band_filter(bd, P(dst), P(sao_src, 160 + 8), 256, 320, P(offs), 7, 128, 128), 16384)
    out["SAO edge 128x128"] = rate(lambda: orc.orc_sao_edge_filter(bd, P(dst), P(sao_src, 160 + 8), 256, P(offs), 1, 128, 128), 16384)
    beta, tc = np.array([40, 44], np.int32), np.array([11, 14], np.int32)
    z2, l3 = np.zeros(2, np.uint8), np.full(2, 3, np.uint8)
    out["deblock luma one 8-line vertical edge"] = rate(lambda: orc.orc_lf_filter_luma(bd, 1, P(dst, 16 * 128 + 16), 256, P(beta), P(tc), P(z2), P(z2), P(l3), P(l3), 0), 64)
    lut = np.sort(rng.integers(0, 1024, size=1024)).astype(np.uint16)
    out["LMCS LUT 128x128"] = rate(lambda: orc.orc_lmcs_filter(bd, P(dst), 256, 128, 128, P(lut)), 16384)
    cpu = next((ln.split(":")[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")), "?")
    print(json.dumps({"cpu": cpu, "threads": 1,
                      "note": "oracle/*.c (gcc -O3) restatements, one thread, called through ctypes (about 1 us of Python call overhead per call is included, which dominates the sub-10-us rows)",
                      "rates": out}, indent=1))


if __name__ == "__main__":
    main()
