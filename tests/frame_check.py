"""Checker for a whole synthetic frame as bench.py runs it: the oracle (oracle/liborc.so, TEST INFRASTRUCTURE) recomputes what one
step of every stage must produce from host snapshots of that stage's inputs, and the result is compared bit for bit with the
device's output.  The prediction / transform stages are checked on a sample of CTUs (all blocks of the sampled CTUs, picture-edge
and partial CTUs always included), the table-driven loop-filter stages over the whole picture.

Used by bench.py's untimed `verified` leg at the bench's own size (7680x4320) and by tests/test_frame_check_gpu.py at small sizes.
The product path never imports this module.
"""
from __future__ import annotations

import ctypes

import numpy as np

from ffvvc_amd import abi


class Mirror:
    """Device allocation -> host copy.  `host_addr(dev_ptr)` translates an address inside any registered allocation."""

    def __init__(self):
        self.ent = {}

    def add(self, dev_ptr: int, host: np.ndarray):
        assert host.flags["C_CONTIGUOUS"]
        self.ent[int(dev_ptr)] = host

    def host_addr(self, dev_ptr: int) -> int:
        dev_ptr = int(dev_ptr)
        if dev_ptr == 0:
            return 0
        h = self.ent.get(dev_ptr)
        if h is not None:
            return h.ctypes.data
        for base, h in self.ent.items():
            if base <= dev_ptr < base + h.nbytes:
                return h.ctypes.data + (dev_ptr - base)
        raise KeyError(f"device address {dev_ptr:#x} is not mirrored on the host")

    def array(self, dev_ptr: int) -> np.ndarray:
        return self.ent[int(dev_ptr)]


def sample_ctus(rng, ncx, ncy, n):
    """CTU raster indices: the four corners, one more on every picture edge, the rest random (the last row / column are partial CTUs
    whenever the picture size is not a multiple of the CTU size)."""
    pick = {0, ncx - 1, (ncy - 1) * ncx, ncy * ncx - 1, ncx // 2, (ncy - 1) * ncx + ncx // 2, (ncy // 2) * ncx, (ncy // 2) * ncx + ncx - 1}
    while len(pick) < min(n, ncx * ncy):
        pick.add(int(rng.integers(0, ncx * ncy)))
    return sorted(pick)


def bind(orc):
    for name, ty in (("orc_deblock_frame_pass", abi.DeblockFrame), ("orc_sao_frame_pass", abi.SaoFrame), ("orc_alf_frame_pass", abi.AlfFrame)):
        getattr(orc, name).argtypes = [ctypes.c_int, ctypes.POINTER(ty)]
        getattr(orc, name).restype = None
    orc.orc_deblock_bs_pass.argtypes = [ctypes.POINTER(abi.BsFrame)]
    orc.orc_deblock_bs_pass.restype = None
    orc.orc_bipred_block.argtypes = [ctypes.c_int, ctypes.POINTER(abi.BipredJob)]
    orc.orc_bipred_block.restype = None
    orc.orc_affine_block.argtypes = [ctypes.c_int, ctypes.POINTER(abi.AffineJob)]
    orc.orc_affine_block.restype = None


def _plane_xy(job_dst, plane_ptr, pitch, isz):
    off = int(job_dst) - int(plane_ptr)
    return (off % pitch) // isz, off // pitch


def check_bipred(orc, bd, jobs, idx, mirror, planes_after, plane_ptrs, pitches, rec_after=None, rec_ptr=0):
    """jobs: numpy array with the layout of vvc355_bipred_job (device addresses).  Every job in `idx` is recomputed by
    orc_bipred_block into a scratch block and compared with the device's block (and, for luma, its refinement record)."""
    isz = 1 if bd == 8 else 2
    dt = np.uint8 if bd == 8 else np.uint16
    bad = 0
    for i in idx:
        j = abi.BipredJob.from_buffer_copy(jobs[i].tobytes())
        c = next(k for k, p in enumerate(plane_ptrs) if planes_after[k] is not None and p <= j.dst < p + planes_after[k].nbytes)
        x, y = _plane_xy(j.dst, plane_ptrs[c], pitches[c], isz)
        blk = np.zeros((j.h, j.w), dt)
        rec = abi.BipredResult()
        dev_rec = None
        if j.rec:
            k = (int(j.rec) - int(rec_ptr)) // 32
            dev_rec = rec_after[k]
            if j.chroma:                          # chroma follows the luma launch's record
                for m in range(4):
                    rec.mv[m] = int(dev_rec[m])
        j.dst, j.dst_stride = blk.ctypes.data, j.w * isz
        j.ref0, j.ref1 = mirror.host_addr(j.ref0), mirror.host_addr(j.ref1)
        j.lmcs_lut = mirror.host_addr(j.lmcs_lut) if j.lmcs_lut else 0
        j.rec = ctypes.addressof(rec) if dev_rec is not None else 0
        orc.orc_bipred_block(bd, ctypes.byref(j))
        got = planes_after[c][y:y + j.h, x:x + j.w]
        if not np.array_equal(got, blk):
            bad += 1
        if dev_rec is not None and not j.chroma:
            want = [rec.mv[0], rec.mv[1], rec.mv[2], rec.mv[3], rec.bdof, rec.min_sad, rec.searched]
            if want != [int(v) for v in dev_rec[:7]]:
                bad += 1
    return bad


def check_itx(orc, bd, jobs, idx, mirror, before, after, plane_ptrs, pitches, coeffs_after=None):
    """dequant (fused scaling process) + inverse transform + residual add of the transform blocks in `idx`; blocks without a
    destination (store_coeffs: the residual stays in the buffer) are compared in coeffs_after = (device address, array)."""
    isz = 1 if bd == 8 else 2
    bad = 0
    for i in idx:
        j = jobs[i]
        lw, lh = int(j["log2_w"]), int(j["log2_h"])
        w, h = 1 << lw, 1 << lh
        dst = int(j["dst"])
        if dst:
            c = next(k for k, p in enumerate(plane_ptrs) if p <= dst < p + before[k].nbytes)
            x, y = _plane_xy(dst, plane_ptrs[c], pitches[c], isz)
        src = np.ctypeslib.as_array((ctypes.c_int32 * (w * h)).from_address(mirror.host_addr(int(j["coeffs"]))))
        co = np.ascontiguousarray(src).copy()
        nzw, nzh, rng_, qp = int(j["nzw"]), int(j["nzh"]), int(j["range"]), int(j["dq_qp"])
        if int(j["dq_flags"]) & 1:
            sm = mirror.host_addr(int(j["scale_matrix"])) if int(j["scale_matrix"]) else None
            orc.orc_dequant(co.ctypes.data, lw, lh, 0, 0, nzw - 1, nzh - 1, qp, 0, (int(j["dq_flags"]) >> 1) & 1, bd, rng_, sm,
                            int(j["log2_matrix_size"]), int(j["dc"]))
        orc.orc_itx(int(j["trh"]), int(j["trv"]), lw, lh, co.ctypes.data, nzw, nzh, rng_, int(j["bd"]))
        if not dst:
            first = (int(j["coeffs"]) - coeffs_after[0]) // 4
            bad += not np.array_equal(coeffs_after[1][first:first + w * h], co)
            continue
        blk = np.ascontiguousarray(before[c][y:y + h, x:x + w])
        orc.orc_add_residual(bd, blk.ctypes.data, co.ctypes.data, w, h, w * isz)
        if not np.array_equal(after[c][y:y + h, x:x + w], blk):
            bad += 1
    return bad


def check_lmcs(orc, bd, rects, lut, before, after):
    isz = 1 if bd == 8 else 2
    bad = 0
    for (x, y, w, h) in rects:
        blk = np.ascontiguousarray(before[y:y + h, x:x + w])
        orc.orc_lmcs_filter(bd, blk.ctypes.data, w * isz, w, h, lut.ctypes.data)
        bad += not np.array_equal(after[y:y + h, x:x + w], blk)
    return bad


def translate(struct, mirror, fields):
    """A copy of a frame descriptor with the named address fields (scalars or arrays of addresses) moved to the host mirror."""
    cp = type(struct).from_buffer_copy(bytes(struct))
    for name in fields:
        v = getattr(cp, name)
        if isinstance(v, int):
            setattr(cp, name, mirror.host_addr(v))
        elif hasattr(v[0], "__len__"):
            for a in range(len(v)):
                for b in range(len(v[a])):
                    v[a][b] = mirror.host_addr(v[a][b])
        else:
            for a in range(len(v)):
                v[a] = mirror.host_addr(v[a])
    return cp


BS_IN = ("mvf", "ref_poc", "slice_idx", "ctb_to_col_bd", "ctb_to_row_bd", "tu_coded_flag", "tu_joint_cbcr", "pcmf", "tb_pos_x0", "tb_pos_y0",
         "tb_width", "tb_height", "cb_pos_x", "cb_pos_y", "cb_width", "cb_height", "msf", "iaf")
BS_OUT = ("bs", "max_len_p", "max_len_q")
DEBLOCK_PTRS = ("plane", "bs", "max_len_p", "max_len_q", "tb_size_c", "qp_y", "qp_c", "db_params")
SAO_PTRS = ("dst", "src", "sao", "slice_idx", "ctb_to_col_bd", "ctb_to_row_bd")
ALF_PTRS = ("dst", "src", "alf", "slices", "slice_idx", "ctb_to_col_bd", "ctb_to_row_bd")
