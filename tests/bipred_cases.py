"""Shared builders for the regular bi-prediction stage tests (oracle property tests on CPU, parity tests on the GPU)."""
import ctypes

import numpy as np

from ffvvc_amd import abi


def bind_oracle(orc):
    orc.orc_bipred_block.argtypes = [ctypes.c_int, ctypes.POINTER(abi.BipredJob)]
    orc.orc_bipred_block.restype = None


def smooth_picture(rng, h, w, bd, scale=8):
    """Band-limited content (bilinear up-sampling of a coarse random grid plus a little noise): the DMVR cost surface of such
    pictures has a real minimum, so both the search and its early terminations are exercised."""
    gh, gw = h // scale + 2, w // scale + 2
    g = rng.integers(0, 1 << bd, size=(gh, gw)).astype(np.float64)
    ys, xs = np.arange(h) / scale, np.arange(w) / scale
    y0, x0 = ys.astype(int), xs.astype(int)
    fy, fx = (ys - y0)[:, None], (xs - x0)[None, :]
    p = (g[y0][:, x0] * (1 - fy) * (1 - fx) + g[y0][:, x0 + 1] * (1 - fy) * fx +
         g[y0 + 1][:, x0] * fy * (1 - fx) + g[y0 + 1][:, x0 + 1] * fy * fx)
    p += rng.normal(0, (1 << bd) / 256, size=p.shape)
    return np.ascontiguousarray(np.clip(np.rint(p), 0, (1 << bd) - 1).astype(np.uint8 if bd == 8 else np.uint16))


def shifted(pic, dx, dy):
    """pic displaced by whole samples with edge replication: out[y, x] = pic[y + dy, x + dx]."""
    h, w = pic.shape
    yy = np.clip(np.arange(h) + dy, 0, h - 1)
    xx = np.clip(np.arange(w) + dx, 0, w - 1)
    return np.ascontiguousarray(pic[yy][:, xx])


def random_blocks(rng, pic_w, pic_h, n_max=10 ** 9):
    """Tile the luma picture with sub-blocks of the sizes DMVR / BDOF allow (8 or 16 on a side) -> (x, y, w, h)."""
    out = []
    y = 0
    while y + 16 <= pic_h:
        x = 0
        while x + 16 <= pic_w:
            if rng.random() < 0.25:
                # small blocks (no DMVR / BDOF for those in VVC): 4 wide / 4 or 12 high -> chroma 2 wide
                w, h = int(rng.choice([4, 8, 16])), int(rng.choice([4, 12]))
            else:
                w, h = int(rng.choice([8, 16])), int(rng.choice([8, 16]))
            out.append((x, y, w, h))
            x += 16
        y += 16
    return out[:n_max]
