"""Input generators for the ALF slots, mirroring the distributions of the reference's own unit test
(tests/checkasm/vvc_alf.c:52-66,77-79,96,108): int8-range coefficients, clip values from
{2^bd, 2^(bd-3), 2^(bd-5), 2^(bd-7)}, virtual boundary 4 (luma) / 2 (chroma) rows above the CTB bottom."""
import numpy as np

from conftest import P, px_dtype, rand_pixels  # noqa: F401

PAD = 8          # ALF_PADDING_SIZE: apron of the padded source plane
SRC_W = 128 + 2 * PAD + 16


def clip_values(bd):
    return np.array([1 << bd, 1 << (bd - 3), 1 << (bd - 5), 1 << (bd - 7)], dtype=np.int16)


def make_src(rng, bd, smooth=False):
    """Padded source plane (160 rows x SRC_W) and the element offset of the CTB origin."""
    src = rand_pixels(rng, (128 + 2 * PAD + 16, SRC_W), bd)
    if smooth:   # low-contrast content exercises the directionality classes more evenly
        base = rng.integers(0, 1 << bd, dtype=np.int64)
        ramp = (np.arange(SRC_W)[None, :] * rng.integers(0, 5) + np.arange(src.shape[0])[:, None] * rng.integers(0, 5))
        src = np.clip(base + ramp + rng.integers(-6, 7, size=src.shape), 0, (1 << bd) - 1).astype(px_dtype(bd))
    return src, PAD * SRC_W + PAD


def luma_params(rng, bd, w, h):
    n = (w // 4) * (h // 4)
    coeff = rng.integers(-128, 128, size=(n, 12), dtype=np.int64).astype(np.int16)
    clip = clip_values(bd)[rng.integers(0, 4, size=(n, 12))]
    return np.ascontiguousarray(coeff), np.ascontiguousarray(clip)


def run_filter(lib, prefix, kind, bd, src, off, w, h, coeff, clip, vb_pos):
    ps = src.itemsize
    dst = np.full((h + 4, w + 16), 0x55, dtype=src.dtype)      # canary-padded output
    fn = getattr(lib, f"{prefix}alf_filter_{kind}")
    fn(bd, P(dst, 2 * dst.shape[1] + 8), dst.shape[1] * ps, P(src, off), src.shape[1] * ps, w, h, P(coeff), P(clip), vb_pos)
    return dst


def run_classify(lib, prefix, bd, src, off, w, h, vb_pos):
    n = (w // 4) * (h // 4)
    cls = np.full(n, -1, dtype=np.int32)
    tr = np.full(n, -1, dtype=np.int32)
    grad = np.zeros(((h + 4) // 2) * ((w + 4) // 2) * 4, dtype=np.int32)
    getattr(lib, f"{prefix}alf_classify")(bd, P(cls), P(tr), P(src, off), src.shape[1] * src.itemsize, w, h, vb_pos, P(grad))
    return cls, tr
