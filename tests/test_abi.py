"""CPU: the product library loads and exports every entry point that include/*.h declares (no compute calls)."""
import ctypes
import glob
import os
import re

from conftest import ROOT
from ffvvc_amd import abi


def declared_symbols():
    names = set()
    for path in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text = open(path).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(vvc355_\w+|ff_vvc_dsp_init_mi355\w*)\s*\(", text))
    return names


# exported by the C host shim (libvvc_mi355_host.so): the table installers and the context flattening of include/vvc_mi355_ctx.h
HOST_SYMBOLS = {"ff_vvc_dsp_init_mi355", "ff_vvc_dsp_init_mi355_ctx", "vvc355_dsp_count_slots", "vvc355_dsp_table_selftest",
                "vvc355_ctx_flatten_cclm", "vvc355_ctx_flatten_intra_pred", "vvc355_ctx_flatten_lmcs_scale", "vvc355_ctx_left_available",
                "vvc355_ctx_top_available", "vvc355_ctx_set_availability"}


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(abi.LIB_PATH)
    host = ctypes.CDLL(os.path.join(ROOT, "ffvvc_amd", "libvvc_mi355_host.so"))
    names = declared_symbols()
    assert len(names) > 60
    missing = [n for n in sorted(names) if not hasattr(host if n in HOST_SYMBOLS else lib, n)]
    assert not missing, f"declared in include/ but not exported: {missing}"


def test_python_binding_covers_every_declared_symbol():
    bound = {"vvc355_" + k for t in (abi.SLOT_SIGNATURES, abi.RUNTIME_SIGNATURES, abi.BATCH_SIGNATURES, abi.FLAT_SIGNATURES) for k in t}
    names = {n for n in declared_symbols() if n.startswith("vvc355_")} - HOST_SYMBOLS
    assert names == bound, f"only in header: {sorted(names - bound)}; only in binding: {sorted(bound - names)}"


def test_job_struct_sizes_match_header():
    assert ctypes.sizeof(abi.AlfJob) == 64
    assert ctypes.sizeof(abi.DequantJob) == 32
    assert ctypes.sizeof(abi.BipredJob) == 104 and ctypes.sizeof(abi.BipredResult) == 32
    assert ctypes.sizeof(abi.AffineJob) == 96
    assert ctypes.sizeof(abi.MvField) == 24 and ctypes.sizeof(abi.BsFrame) == 312
    assert ctypes.sizeof(abi.ReconCmd) == 40 and ctypes.sizeof(abi.ReconCtu) == 12 and ctypes.sizeof(abi.ReconFrame) == 128 and ctypes.sizeof(abi.LmcsModel) == 72 and ctypes.sizeof(abi.LmcsResidJob) == 56
    assert ctypes.sizeof(abi.LfnstJob) == 32 and ctypes.sizeof(abi.GpmJob) == 120 and ctypes.sizeof(abi.ItxJob) == 48
    assert ctypes.sizeof(abi.InterPu) == 20 and ctypes.sizeof(abi.InterSlice) == 390 and ctypes.sizeof(abi.InterFrame) == 136 and ctypes.sizeof(abi.RefPic) == 40
    assert ctypes.sizeof(abi.AlfCtb) == 8 and ctypes.sizeof(abi.AlfSlice) == 160 and ctypes.sizeof(abi.AlfFrame) == 136


def test_flat_slots_reject_neighbours_outside_the_picture():
    """A context that claims an upper neighbour for a block in the picture's first row (round 2's unexplained abort: lmcs_scale_chroma with
    lc->ctb_up_flag = 1 at y = 0 read in front of the staged window — a GPU memory fault, a bare abort) is outside the slots' domain:
    the library says so and aborts before touching the device.  Runs without a GPU (the check precedes every HIP call)."""
    import subprocess
    import sys
    code = f"""
import ctypes, sys
sys.path.insert(0, {ROOT!r})
from ffvvc_amd import abi
lib = ctypes.CDLL(abi.LIB_PATH)
j = abi.LmcsScaleJob()
j.x_vpdu, j.y_vpdu, j.pic_w, j.pic_h, j.size_y, j.avail_t, j.avail_l = 192, 0, 256, 128, 64, 1, 1
buf = (ctypes.c_int * 16)()
lib.vvc355_lmcs_scale_chroma_flat.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
lib.vvc355_lmcs_scale_chroma_flat(10, ctypes.byref(j), buf, buf, 4, 4)
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, timeout=120)
    assert r.returncode == -6, (r.returncode, r.stderr[-300:])
    assert b"lmcs_scale_chroma at (192, 0) claims neighbours outside the picture" in r.stderr
