"""Shared test plumbing.

* ``orc``  — the CPU oracle (oracle/liborc.so): the CHECKER.  Only tests, smoke() and the bench's
  cpu_baseline leg may touch it.
* ``dev``  — the product library (ffvvc_amd/libvvc_mi355.so), called through its C ABI.
Tests marked ``gpu`` need an MI355X; everything else runs on CPU.
"""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from ffvvc_amd import abi  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_oracle() -> ctypes.CDLL:
    so = os.path.join(ROOT, "oracle", "liborc.so")
    srcs = [os.path.join(ROOT, "oracle", f) for f in os.listdir(os.path.join(ROOT, "oracle"))
            if f.endswith((".c", ".h"))]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    lib = ctypes.CDLL(so)
    table = {k: v for k, v in abi.SLOT_SIGNATURES.items() if hasattr(lib, "orc_" + k)}
    abi.bind(lib, "orc_", table)
    return lib


@pytest.fixture(scope="session")
def orc():
    return load_oracle()


@pytest.fixture(scope="session")
def dev():
    lib = abi.load()
    if lib.vvc355_device_count() < 1:
        pytest.fail("gpu test selected but no HIP device is visible")
    return lib


def P(arr: np.ndarray, offset_elems: int = 0) -> int:
    """Address of element `offset_elems` of a C-contiguous numpy array."""
    assert arr.flags["C_CONTIGUOUS"]
    return arr.ctypes.data + offset_elems * arr.itemsize


def px_dtype(bd: int):
    return np.uint8 if bd == 8 else np.uint16


def rand_pixels(rng, shape, bd):
    return rng.integers(0, 1 << bd, size=shape, dtype=np.int64).astype(px_dtype(bd))
