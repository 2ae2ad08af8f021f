"""Synthetic RECON workloads (vvc355_recon_frame_pass / orc_recon_frame_pass): a random partition of every CTU into coding units
(tests/bs_cases._split), intra or not, flattened into the per-CTU command lists the RECON stage driver consumes — one command per
reference call, in the order ff_vvc_reconstruct / reconstruct / predict_intra / itransform make them (vvc_intra.c:245-274,
:431-527).  Used by the GPU parity test and by bench.py's intra stage."""
import numpy as np

from ffvvc_amd import abi
from bs_cases import _split

CMD = np.dtype(abi.ReconCmd, align=True)
CTU = np.dtype(abi.ReconCtu, align=True)


class ReconWork:
    """cmds / ctus / order as numpy arrays with the C layouts; `resid_len` int32 entries of residual storage are addressed by
    cmds["resid"] as ELEMENT offsets until `bind(base_address)` turns them into addresses."""

    def __init__(self, rng, width, height, ctb_log2=7, hs=1, vs=1, intra_frac=1.0, intra_ctu=None, cclm_frac=0.2, coded_p=0.7,
                 n_slices=1, tiles=False, min_cu=8, split=(0.85, 0.45), tools=True, ciip_frac=0.0, ciip_ctu=None, isp_p=0.15, lmcs=False, resid_ctu=None):
        self.width, self.height, self.ctb_log2, self.hs, self.vs = width, height, ctb_log2, hs, vs
        ctb = 1 << ctb_log2
        self.ncx, self.ncy = (width + ctb - 1) // ctb, (height + ctb - 1) // ctb
        n_ctb = self.ncx * self.ncy
        cuts = np.sort(rng.integers(1, max(2, n_ctb), size=n_slices - 1)) if n_slices > 1 else []
        self.slice_idx = np.searchsorted(cuts, np.arange(n_ctb), side="right").astype(np.int16)
        if tiles and self.ncx > 2 and self.ncy > 1:
            cx, cy = int(rng.integers(1, self.ncx)), int(rng.integers(1, self.ncy))
            self.col_bd = np.array([0 if x < cx else cx for x in range(self.ncx)] + [self.ncx], np.int16)
            self.row_bd = np.array([0 if y < cy else cy for y in range(self.ncy)] + [self.ncy], np.int16)
        else:
            self.col_bd = np.array([0] * self.ncx + [self.ncx], np.int16)
            self.row_bd = np.array([0] * self.ncy + [self.ncy], np.int16)
        cmds, ctus = [], np.zeros(n_ctb, CTU)
        kind_of = np.zeros(n_ctb, np.int8)          # 2: the walk writes this CTU's luma (intra / CIIP units); 1: only residuals of inter units (a LIGHT CTU)
        self.resid_len = 0
        self.tbs = []                     # (c_idx, x0, y0 luma, w, h component samples, element offset): what the transform stage must fill
        self.ciip = []                    # (c_idx, x0, y0, w, h luma units, pixel offset into the inter-prediction storage, command index)
        self.ciip_len = 0
        self.ciip_frac = ciip_frac
        self.isp_p = isp_p
        self.lmcs = lmcs                  # chroma residual scaling on (sh_lmcs_used_flag && ph_chroma_residual_scale_flag): RESID bit 3 on chroma blocks of more than 4 samples
        for rs in range(n_ctb):
            rx, ry = rs % self.ncx, rs // self.ncx
            first = len(cmds)
            ctu_intra = intra_frac if intra_ctu is None else (1.0 if intra_ctu[rs] else 0.0)
            leaves = []
            whole_ciip = ciip_ctu is not None and bool(ciip_ctu[rs])
            if whole_ciip:          # a CTU of 32x32 combined inter / intra coding units
                leaves = [(x, y, 32, 32) for y in range(ry * ctb, min(ry * ctb + ctb, height - 31), 32) for x in range(rx * ctb, min(rx * ctb + ctb, width - 31), 32)]
            else:
                _split(rng, rx * ctb, ry * ctb, ctb, ctb, width, height, min_cu, leaves, *split)
            any_intra = False
            inter_resid = False
            cu_cmds = []
            for (x, y, w, h) in leaves:
                if whole_ciip or rng.random() >= ctu_intra and self.ciip_frac and w * h >= 64 and w < 128 and h < 128 and rng.random() < self.ciip_frac:
                    # combined inter / intra (ff_vvc_predict_ciip): planar intra prediction of each component, blended with the inter
                    # prediction the batched stage left aside; the area is recorded like any reconstructed coding unit
                    any_intra = True
                    cu = (x, y, w, h)
                    iw = int(rng.integers(1, 4))
                    for c in range(3):
                        if c and (w >> self.hs) <= 2:
                            continue
                        cu_cmds.append(self._cmd(abi.RECON_PRED, c, x, y, w, h, *cu, mode=0))
                        off = self.ciip_len
                        self.ciip_len += (w >> (self.hs if c else 0)) * (h >> (self.vs if c else 0))
                        self.ciip.append((c, x, y, w, h, off, len(cmds) + len(cu_cmds)))
                        cu_cmds.append(self._cmd(abi.RECON_CIIP, c, x, y, w, h, *cu, resid=off, joint=iw))
                    cu_cmds.append(self._cmd(abi.RECON_MARK, 0, x, y, w, h, *cu))
                    cu_cmds.append(self._cmd(abi.RECON_MARK, 1, x, y, w, h, *cu))
                    continue
                if rng.random() >= ctu_intra:
                    # not intra-coded: prediction and residual come from the batched stages, the walk only records the area — unless the
                    # chroma residuals of this CTU's inter units are left to the walk (resid_ctu): with chroma residual scaling they depend
                    # on the reconstructed luma around their 64x64 unit, which may be an intra CTU's (itransform runs for every coding unit
                    # in the reference's RECON stage, vvc_intra.c:480-496)
                    cu_cmds.append(self._cmd(abi.RECON_MARK, 0, x, y, w, h, x, y, w, h))
                    cu_cmds.append(self._cmd(abi.RECON_MARK, 1, x, y, w, h, x, y, w, h))
                    if resid_ctu is not None and resid_ctu[rs] and (w >> self.hs) >= 4 and rng.random() < coded_p:
                        cwc, chc = w >> self.hs, h >> self.vs
                        for c in (1, 2):
                            cu_cmds.append(self._cmd(abi.RECON_RESID, c, x, y, cwc, chc, x, y, w, h, resid=self._resid(c, x, y, cwc, chc), joint=8 if (lmcs and cwc * chc > 4) else 0))
                        inter_resid = True
                    continue
                any_intra = True
                self._intra_cu(rng, cu_cmds, x, y, w, h, ctb, cclm_frac, coded_p, tools)
            if any_intra or inter_resid:
                cmds += cu_cmds
                ctus[rs]["first_cmd"], ctus[rs]["n_cmd"] = first, len(cu_cmds)
                kind_of[rs] = 2 if any_intra else 1
        # scheduling hints of the device pass (vvc355_recon_ctu.flags): LIGHT CTUs and whose luma they wait for
        for rs in np.nonzero(kind_of == 1)[0]:
            rx, ry = rs % self.ncx, rs // self.ncx
            ctus[rs]["flags"] = (abi.RECON_CTU_LIGHT | (abi.RECON_CTU_LUMA_LEFT if rx and kind_of[rs - 1] == 2 else 0) |
                                 (abi.RECON_CTU_LUMA_UP if ry and kind_of[rs - self.ncx] == 2 else 0))
        self.cmds = np.array(cmds, CMD) if cmds else np.zeros(0, CMD)
        self.ctus = ctus
        self.order = np.nonzero(ctus["n_cmd"])[0].astype(np.int32)

    @staticmethod
    def _cmd(kind, c_idx, x0, y0, w, h, cu_x, cu_y, cb_w, cb_h, resid=0, mode=0, ref_idx=0, is_mip=0, mip_mode=0, mip_transposed=0,
             isp_split=0, bdpcm_flag=0, joint=0):
        # a tuple in the field order of vvc355_recon_cmd (plain tuples: a frame has hundreds of thousands of commands)
        return (resid, x0, y0, w, h, cu_x, cu_y, cb_w, cb_h, mode, kind, c_idx, ref_idx, is_mip, mip_mode, mip_transposed, isp_split,
                bdpcm_flag, joint, (0,) * 6)

    def _resid(self, c_idx, x0, y0, w, h):
        off = self.resid_len
        self.resid_len += w * h
        self.tbs.append((c_idx, x0, y0, w, h, off))
        return off

    def _intra_cu(self, rng, out, x, y, w, h, ctb, cclm_frac, coded_p, tools):
        hs, vs = self.hs, self.vs
        cu = (x, y, w, h)
        mode = int(rng.choice([0, 1, 18, 50] + list(range(2, 67))))
        y0b = y & (ctb - 1)
        isp = tools and h <= 64 and ((w >= 8 and h >= 16) or (w in (4, 8, 16) and h >= 8 and w * h >= 64 and rng.random() < 0.5)) and rng.random() < self.isp_p
        is_mip = tools and not isp and w <= 64 and h <= 64 and rng.random() < 0.1
        ref_idx = int(rng.choice([1, 2])) if (tools and y0b and not is_mip and not isp and mode != 0 and rng.random() < 0.15) else 0
        bdpcm = int(tools and mode in (18, 50) and not isp and not is_mip and w <= 32 and h <= 32 and rng.random() < 0.3)
        kw = dict(mode=0 if is_mip else mode, ref_idx=ref_idx, is_mip=int(is_mip), isp_split=int(isp), bdpcm_flag=bdpcm)
        if is_mip:
            size_id = 0 if (w == 4 and h == 4) else 1 if (w == 4 or h == 4 or (w == 8 and h == 8)) else 2
            kw["mip_mode"], kw["mip_transposed"] = int(rng.integers(0, (16, 8, 6)[size_id])), int(rng.integers(0, 2))
        # luma (ch_type 0): one transform unit, or ISP's horizontal sub-partitions — predict, record, add the residual
        parts = [(x, y, w, h)]
        vertical = isp and w <= 16 and (h < 16 or rng.random() < 0.5)
        if vertical:
            # ISP_VER_SPLIT: four sub-partitions side by side, 1 / 2 / 4 samples wide for coding units 4 / 8 / 16 wide.  Below 4 samples the
            # prediction covers 4 columns at once, on every (4 / width)-th sub-partition only (get_luma_predict_unit, vvc_intra.c:216-226)
            parts = [(x + i * (w // 4), y, w // 4, h) for i in range(4)]
        elif isp:
            k = 4 if h >= 32 else 2
            parts = [(x, y + i * h // k, w, h // k) for i in range(k)]
        for idx, (px, py, pw, ph) in enumerate(parts):
            if pw >= 4 or idx % (4 // pw) == 0:
                out.append(self._cmd(abi.RECON_PRED, 0, px, py, max(pw, 4), ph, *cu, **kw))
                out.append(self._cmd(abi.RECON_MARK, 0, px, py, max(pw, 4), ph, *cu))
            if rng.random() < coded_p:
                out.append(self._cmd(abi.RECON_RESID, 0, px, py, pw, ph, *cu, resid=self._resid(0, px, py, pw, ph)))
        # chroma (ch_type 1): both components predicted over the coding unit, then their residuals (joint Cb-Cr now and then)
        cw, chh = w >> hs, h >> vs
        if tools and rng.random() < cclm_frac:
            out.append(self._cmd(abi.RECON_CCLM, 1, x, y, w, h, *cu, mode=int(rng.choice([81, 82, 83]))))
        else:
            cmode = int(rng.choice([0, 1, 18, 50, mode]))
            for c in (1, 2):
                out.append(self._cmd(abi.RECON_PRED, c, x, y, w, h, *cu, mode=cmode, bdpcm_flag=0))
        out.append(self._cmd(abi.RECON_MARK, 1, x, y, w, h, *cu))
        joint = tools and rng.random() < 0.15
        sc = 8 if (self.lmcs and cw * chh > 4) else 0        # itransform's chroma_scale (vvc_intra.c:449)
        for c in (1, 2):
            if rng.random() < coded_p:
                out.append(self._cmd(abi.RECON_RESID, c, x, y, cw, chh, *cu, resid=self._resid(c, x, y, cw, chh), joint=sc))
                if joint and c == 1:
                    # add_residual_for_joint_coding_chroma (:166-186): the same residual goes to the other component, signed / halved
                    out.append(self._cmd(abi.RECON_RESID, 2, x, y, cw, chh, *cu, resid=out[-1][0], joint=sc | 1 | (2 * int(rng.integers(0, 2))) | (4 * int(rng.integers(0, 2)))))
                    break

    def bind(self, base, ciip_base=0, isz=2):
        """Command array with residual element offsets turned into addresses at `base` (int32 storage), CIIP pixel offsets into
        addresses at `ciip_base` (pixels of `isz` bytes)."""
        c = self.cmds.copy()
        is_res = c["kind"] == abi.RECON_RESID
        c["resid"][is_res] = base + c["resid"][is_res] * 4
        is_ci = c["kind"] == abi.RECON_CIIP
        c["resid"][is_ci] = ciip_base + c["resid"][is_ci] * isz
        return c

    @staticmethod
    def lmcs_model(rng, bd):
        """An LMCS model as lmcs_derive_chroma_scale reads it: 16 bins with increasing pivots, a chroma scale per bin (11-bit fixed point around 1.0)."""
        m = abi.LmcsModel()
        cuts = np.sort(rng.choice(np.arange(1, 1 << bd), size=15, replace=False))
        for i, v in enumerate([0] + [int(v) for v in cuts] + [1 << bd]):
            m.pivot[i] = min(v, 65535)
        for i in range(16):
            m.chroma_scale_coeff[i] = int(rng.integers(1024, 4096))
        m.min_bin_idx, m.max_bin_idx = int(rng.integers(0, 3)), int(rng.integers(12, 16))
        return m

    def frame(self, planes, strides, cmds_ptr, ctus_ptr, order_ptr, state_ptr, slice_ptr, col_ptr, row_ptr, wpp=0, collocated=0, lmcs_ptr=0):
        f = abi.ReconFrame()
        for c in range(3):
            f.plane[c], f.stride[c] = planes[c], strides[c]
        f.cmds, f.ctus, f.order, f.state = cmds_ptr, ctus_ptr, order_ptr, state_ptr
        f.slice_idx, f.ctb_to_col_bd, f.ctb_to_row_bd = slice_ptr, col_ptr, row_ptr
        f.width, f.height, f.ctb_width, f.ctb_height, f.n_work = self.width, self.height, self.ncx, self.ncy, len(self.order)
        f.ctb_log2, f.hs, f.vs, f.wpp, f.collocated = self.ctb_log2, self.hs, self.vs, wpp, collocated
        f.lmcs_model = lmcs_ptr
        return f


def critical_order(lib, ctus, ncx, ncy):
    """vvc355_recon_order (host helper of the C ABI): the ticket order that keeps the longest dependency chains moving."""
    out = np.zeros(ncx * ncy, np.int32)
    table = np.ascontiguousarray(ctus)
    n = lib.vvc355_recon_order(table.ctypes.data, ncx, ncy, out.ctypes.data)
    return out[:n].copy()
