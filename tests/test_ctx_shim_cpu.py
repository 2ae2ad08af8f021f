"""CPU: the host shim's flattening of the decoder context into job descriptors (ffvvc_amd/host/dsp_ctx_shim.c, the in-tree half of
the drop-in boundary for intra_pred / intra_cclm_pred / lmcs_scale_chroma): the availability process over the running list of
reconstructed areas, the wide-angle mapping and cand_up_left must match what the oracle's RECON walk derives for the same block of
the same partition (oracle/orc_recon.c, restating vvc_intra.c:574-714 independently).  No GPU work: only descriptors are compared."""
import ctypes

import numpy as np

import ctx_mirror as cm
import recon_cases
from ffvvc_amd import abi


def test_flattened_intra_jobs_match_the_oracle_walk(orc):
    host = cm.load_host()
    host.vvc355_ctx_flatten_intra_pred.argtypes = [ctypes.POINTER(cm.VVCLocalContext)] + [ctypes.c_int] * 5 + [ctypes.POINTER(abi.IntraJob)]
    host.vvc355_ctx_flatten_intra_pred.restype = None
    orc.orc_recon_debug_job.argtypes = [ctypes.POINTER(abi.ReconFrame), ctypes.c_int, ctypes.c_int, ctypes.POINTER(abi.IntraJob)]
    orc.orc_recon_debug_job.restype = None
    checked = 0
    for case, (hs, vs, ctb_log2, n_slices, tiles, wpp) in enumerate([(1, 1, 6, 3, True, 0), (0, 0, 5, 1, False, 1), (1, 0, 7, 2, False, 0)]):
        rng = np.random.default_rng(0x5EED1100 + case)
        w, h = 328, 200
        work = recon_cases.ReconWork(rng, w, h, ctb_log2, hs, vs, intra_frac=0.8, n_slices=n_slices, tiles=tiles)
        cmds = work.bind(0)
        f = work.frame([0, 0, 0], [w, w >> hs, w >> hs], cmds.ctypes.data, work.ctus.ctypes.data, work.order.ctypes.data, 0,
                       work.slice_idx.ctypes.data, work.col_bd.ctypes.data, work.row_bd.ctypes.data, wpp=wpp)
        ctb = 1 << ctb_log2
        min_cb_w = w // 4
        imf, imm, imtf = (np.zeros((h // 4) * min_cb_w, np.uint8) for _ in range(3))
        fc = cm.VVCFrameContext()
        fc.width, fc.height, fc.bit_depth, fc.ctb_log2_size_y, fc.min_cb_log2_size_y, fc.min_cb_width = w, h, 10, ctb_log2, 2, min_cb_w
        for c in range(3):
            fc.hshift[c], fc.vshift[c] = (hs, vs) if c else (0, 0)
            fc.linesize[c] = (w >> (hs if c else 0)) * 2
        fc.sps_entropy_coding_sync_enabled_flag = wpp
        fc.imf, fc.imm, fc.imtf = imf.ctypes.data, imm.ctypes.data, imtf.ctypes.data
        lc = cm.VVCLocalContext()
        lc.fc = ctypes.pointer(fc)
        for rs in work.order[::3]:
            rx, ry = int(rs) % work.ncx, int(rs) // work.ncx
            # ff_vvc_decode_neighbour (vvc_ctu.c:2468-2495), the host side's job
            left_tile = rx > 0 and work.col_bd[rx] != work.col_bd[rx - 1]
            upper_tile = ry > 0 and work.row_bd[ry] != work.row_bd[ry - 1]
            upper_slice = ry > 0 and work.slice_idx[rs] != work.slice_idx[rs - work.ncx]
            lc.end_of_tiles_x = min(rx * ctb + ctb, w) if work.col_bd[rx] != work.col_bd[rx + 1] else w
            lc.ctb_left_flag, lc.ctb_up_flag = int(rx > 0 and not left_tile), int(ry > 0 and not upper_tile and not upper_slice)
            lc.num_ras[0] = lc.num_ras[1] = 0
            first, n = int(work.ctus[rs]["first_cmd"]), int(work.ctus[rs]["n_cmd"])
            for k in range(n):
                c = cmds[first + k]
                kind, c_idx = int(c["kind"]), int(c["c_idx"])
                if kind == abi.RECON_MARK:
                    ch = int(c_idx > 0)
                    a = lc.ras[ch][lc.num_ras[ch]]
                    sh_x, sh_y = (hs, vs) if ch else (0, 0)
                    a.x, a.y, a.w, a.h = int(c["x0"]) >> sh_x, int(c["y0"]) >> sh_y, int(c["w"]) >> sh_x, int(c["h"]) >> sh_y
                    lc.num_ras[ch] += 1
                elif kind == abi.RECON_PRED and k % 2 == 0:
                    cu = cm.CodingUnit()
                    cu.x0, cu.y0, cu.cb_width, cu.cb_height = int(c["cu_x0"]), int(c["cu_y0"]), int(c["cb_width"]), int(c["cb_height"])
                    cu.intra_pred_mode_y = cu.intra_pred_mode_c = int(c["mode"])
                    cu.intra_luma_ref_idx, cu.isp_split_type = int(c["ref_idx"]), int(c["isp_split"])
                    cu.mip_chroma_direct_flag = 1
                    cu.bdpcm_flag[c_idx] = int(c["bdpcm_flag"])
                    lc.cu = ctypes.pointer(cu)
                    at = (int(c["y0"]) >> 2) * min_cb_w + (int(c["x0"]) >> 2)
                    imf[at], imm[at], imtf[at] = int(c["is_mip"]), int(c["mip_mode"]), int(c["mip_transposed"])
                    # ff_vvc_set_neighbour_available (vvc_ctu.c:2497-2510)
                    x0b, y0b = int(c["x0"]) & (ctb - 1), int(c["y0"]) & (ctb - 1)
                    cand_up, cand_left = bool(lc.ctb_up_flag or y0b), bool(lc.ctb_left_flag or x0b)
                    lc.na.cand_up_left = int((cand_left and cand_up) if (x0b or y0b) else (lc.ctb_left_flag and lc.ctb_up_flag))
                    got, want = abi.IntraJob(), abi.IntraJob()
                    host.vvc355_ctx_flatten_intra_pred(ctypes.byref(lc), int(c["x0"]), int(c["y0"]), int(c["w"]), int(c["h"]), c_idx, ctypes.byref(got))
                    orc.orc_recon_debug_job(ctypes.byref(f), int(rs), k, ctypes.byref(want))
                    for name in ("x", "y", "w", "h", "mode", "cb_width", "cb_height", "left_avail", "top_avail", "c_idx", "ref_idx", "is_mip",
                                 "mip_mode", "mip_transposed", "isp_split", "bdpcm_flag", "cand_up_left"):
                        assert getattr(got, name) == getattr(want, name), (case, int(rs), k, name, getattr(got, name), getattr(want, name))
                    checked += 1
    assert checked > 300
