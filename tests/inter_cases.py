"""Input generators for the inter-prediction slots, mirroring tests/checkasm/vvc_mc.c:34-66,87-90,185-188,302-306:
uniform pixels, 14-bit intermediates, uniformly random MV fractions / filter sets, weights and offsets in int8 range."""
import ctypes

import numpy as np

from conftest import P, px_dtype, rand_pixels  # noqa: F401

PB = 128


def tables(lib, prefix):
    luma = np.ctypeslib.as_array((ctypes.c_int8 * 384).in_dll(lib, prefix + "tab_inter_luma_filters")).reshape(3, 16, 8).copy()
    chroma = np.ctypeslib.as_array((ctypes.c_int8 * 384).in_dll(lib, prefix + "tab_inter_chroma_filters")).reshape(3, 32, 4).copy()
    return luma, chroma


def src_plane(rng, bd):
    """(128+16) x (128+16+16) pixel plane; block origin at (8, 8)."""
    plane = rand_pixels(rng, (PB + 16, PB + 32), bd)
    return plane, 8 * plane.shape[1] + 8


def i16_plane(rng, rows=PB + 8, lo_bits=14):
    return rng.integers(0, 1 << lo_bits, size=(rows, PB), dtype=np.int64).astype(np.int16)


def signed_i16_plane(rng, rows=PB + 8):
    """14-bit-scaled prediction samples the way put() produces them (can be slightly negative / above 2^14)."""
    return rng.integers(-2000, (1 << 14) + 2000, size=(rows, PB), dtype=np.int64).astype(np.int16)
