"""Helpers for the batched (device-resident) entries of the C ABI: device buffers, planes and job arrays.

Used by the GPU tests and by bench.py to lay synthetic frames out in HBM the way a frame-resident
decoder integration would (one planar uint8/uint16 array per component, row pitch a multiple of 256 bytes so
every CTB row starts on a 128-byte line).  No compute happens here.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import abi


class DeviceBuffer:
    """A plain hipMalloc allocation owned through the C ABI."""

    def __init__(self, nbytes: int):
        self.lib = abi.load()
        self.nbytes = int(nbytes)
        self.ptr = self.lib.vvc355_malloc(self.nbytes)

    @classmethod
    def from_host(cls, arr: np.ndarray) -> "DeviceBuffer":
        arr = np.ascontiguousarray(arr)
        buf = cls(arr.nbytes)
        buf.lib.vvc355_upload(buf.ptr, arr.ctypes.data, arr.nbytes)
        return buf

    def to_host(self, dtype, shape) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        self.lib.vvc355_download(out.ctypes.data, self.ptr, out.nbytes)
        return out

    def free(self):
        if self.ptr:
            self.lib.vvc355_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def plane_pitch(width: int, itemsize: int) -> int:
    """Row pitch in bytes: the picture width rounded up to a multiple of 256 bytes."""
    return (width * itemsize + 255) // 256 * 256


def to_pitched(arr: np.ndarray) -> np.ndarray:
    """Copy a (H, W) plane into a (H, pitch/itemsize) array."""
    h, w = arr.shape
    pitch = plane_pitch(w, arr.itemsize) // arr.itemsize
    out = np.zeros((h, pitch), dtype=arr.dtype)
    out[:, :w] = arr
    return out


def jobs_to_device(jobs) -> DeviceBuffer:
    """Upload a ctypes array of job structs."""
    raw = np.frombuffer(bytes(jobs), dtype=np.uint8)
    return DeviceBuffer.from_host(raw)


def alf_luma_jobs(dst_ptr, src_ptr, pitch, itemsize, width, height, ctb, per_ctb):
    """One fused-ALF job per CTB of a width x height luma plane.

    per_ctb(rx, ry) -> (coeff_set_dev, clip_idx_dev, class_to_filt_dev) device addresses for that CTB.
    Mirrors ff_vvc_alf_filter (libavcodec/vvc/vvc_filter.c:1254-1318): rectangle = CTB clipped to the picture,
    vb_pos = ctb - 4, picture edges replicate (edges[] -> ext_* = 0).
    """
    ncx, ncy = (width + ctb - 1) // ctb, (height + ctb - 1) // ctb
    arr = (abi.AlfJob * (ncx * ncy))()
    for ry in range(ncy):
        for rx in range(ncx):
            x0, y0 = rx * ctb, ry * ctb
            w, h = min(ctb, width - x0), min(ctb, height - y0)
            j = arr[ry * ncx + rx]
            off = y0 * pitch + x0 * itemsize
            j.dst, j.src = dst_ptr + off, src_ptr + off
            j.dst_stride = j.src_stride = pitch
            j.w, j.h, j.vb_pos = w, h, ctb - 4
            j.ext_l, j.ext_t = min(3, x0), min(3, y0)
            j.ext_r, j.ext_b = min(3, width - x0 - w), min(3, height - y0 - h)
            j.coeff, j.clip, j.class_to_filt = per_ctb(rx, ry)
    return arr


# ---------------------------------------------------------------------------------------------------------------------
# Vectorised job builders (numpy structured arrays laid out like the C job structs) for whole-frame batches.

def job_array(struct_cls, n: int) -> np.ndarray:
    """Zeroed array of n job descriptors with the exact C layout of `struct_cls` (a ctypes.Structure)."""
    return np.zeros(n, dtype=np.dtype(struct_cls, align=True))


def ctb_grid(width: int, height: int, ctb: int):
    """(x0, y0, w, h) of every CTB of a width x height plane, raster order (partial CTBs at the right / bottom edge)."""
    xs, ys = np.arange(0, width, ctb), np.arange(0, height, ctb)
    x0, y0 = np.meshgrid(xs, ys)
    x0, y0 = x0.ravel(), y0.ravel()
    return x0, y0, np.minimum(ctb, width - x0), np.minimum(ctb, height - y0)


def block_grid(width: int, height: int, bw: int, bh: int):
    """Origins of every full bw x bh block of the plane (the plane is assumed to be a multiple of the block size)."""
    xs, ys = np.arange(0, width - bw + 1, bw), np.arange(0, height - bh + 1, bh)
    x0, y0 = np.meshgrid(xs, ys)
    return x0.ravel(), y0.ravel()
