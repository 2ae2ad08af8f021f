"""ffvvc_amd — MI355X-native (gfx950) pixel-kernel backend for the ffvvc VVC decoder.

The product is the C-ABI shared library ``libvvc_mi355.so`` (hand-written HIP kernels, see
``include/vvc_mi355.h``) plus the C host shim that installs it into a VVCDSPContext-shaped
function-pointer table (``ffvvc_amd/host/dsp_init_mi355.c``).  This Python package only binds the
C ABI for tests and for ``bench.py``; it contains no compute path of its own.
"""
from . import abi  # noqa: F401
from .abi import load  # noqa: F401
