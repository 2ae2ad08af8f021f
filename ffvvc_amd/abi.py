"""ctypes binding of libvvc_mi355.so (the C ABI declared in include/vvc_mi355.h).

The library is the product: there is no CPU fallback.  Loading fails loudly when the in-tree
shared object has not been built (``python -c 'import __graft_entry__ as g; g.build()'``).
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvvc_mi355.so")

_C = {
    "i": ctypes.c_int,
    "p": ctypes.c_void_p,
    "q": ctypes.c_ssize_t,     # ptrdiff_t / intptr_t
    "z": ctypes.c_size_t,
    "v": None,
}

# name (without the vvc355_/orc_ prefix) -> (return code, argument codes).
# The CPU oracle used by the tests exports the same slot signatures under the orc_ prefix, so the
# table is shared between the two libraries (tests/conftest.py).
SLOT_SIGNATURES = {
    # ---- ALF
    "alf_filter_luma":          ("v", "ipqpqiippi"),
    "alf_filter_chroma":        ("v", "ipqpqiippi"),
    "alf_filter_cc":            ("v", "ipqpqiiiipi"),
    "alf_classify":             ("v", "ipppqiiip"),
    "alf_recon_coeff_and_clip": ("v", "ippppippp"),
    # ---- inter
    "put":                ("v", "iiiippqippi"),
    "put_uni":            ("v", "iiiipqpqippi"),
    "put_uni_w":          ("v", "iiiipqpqiiiippi"),
    "avg":                ("v", "ipqppii"),
    "w_avg":              ("v", "ipqppiiiiiii"),
    "put_ciip":           ("v", "ipqiipqi"),
    "put_gpm":            ("v", "ipqiipppii"),
    "fetch_samples":      ("v", "ippqii"),
    "bdof_fetch_samples": ("v", "ippqiiii"),
    "prof_grad_filter":   ("v", "ippqpqiii"),
    "apply_prof":         ("v", "ipppp"),
    "apply_prof_uni":     ("v", "ipqppp"),
    "apply_prof_uni_w":   ("v", "ipqpppiii"),
    "apply_bdof":         ("v", "ipqppii"),
    "sad":                ("i", "ppiiii"),
    "dmvr":               ("v", "iiippqiqqi"),
    # ---- LMCS / SAO / deblock
    "lmcs_filter":        ("v", "ipqiip"),
    "sao_band_filter":    ("v", "ippqqpiii"),
    "sao_edge_filter":    ("v", "ippqpiii"),
    "sao_edge_restore":   ("v", "iippqqpipiippp"),
    "lf_filter_luma":     ("v", "iipqppppppi"),
    "lf_filter_chroma":   ("v", "iipqppppppi"),
    "lf_ladf_level":      ("i", "iipq"),
    # ---- inverse transform / residual (no leading bd where the reference slot is bit-depth independent)
    "itx":                  ("i", "iiiipzzqq"),
    "inv_lfnst_1d":         ("v", "ppiiiii"),
    "dequant":              ("v", "piiiiiiiiiiipii"),
    "ilfnst_transform":     ("i", "piiiii"),
    "derive_transform_type": ("i", "iiiiii"),
    "add_residual":         ("v", "ippiiq"),
    "add_residual_joint":   ("v", "ippiiqii"),
    "pred_residual_joint":  ("v", "piiii"),
    "transform_bdpcm":      ("v", "piiii"),
    # ---- intra (leaf predictors: stride in PIXELS)
    "pred_planar":        ("v", "ipppiiq"),
    "pred_dc":            ("v", "ipppiiq"),
    "pred_v":             ("v", "ippiiq"),
    "pred_h":             ("v", "ippiiq"),
    "pred_angular_v":     ("v", "ipppiiqiiiii"),
    "pred_angular_h":     ("v", "ipppiiqiiiii"),
    "pred_mip":           ("v", "ipppiiqii"),
    "intra_pred_flat":    ("v", "ip"),
}

# flattened "fat" slots whose signatures differ between the product (needs staging bounds) and the oracle
FLAT_SIGNATURES = {
    "intra_cclm_pred_flat":    ("v", "ipii"),
    "lmcs_scale_chroma_flat":  ("v", "ipppii"),
}

RUNTIME_SIGNATURES = {
    "device_count":   ("i", ""),
    "set_device":     ("v", "i"),
    "malloc":         ("p", "z"),
    "free":           ("v", "p"),
    "upload":         ("v", "ppz"),
    "download":       ("v", "ppz"),
    "copy_async":     ("v", "pppz"),
    "stream_create":  ("p", ""),
    "stream_destroy": ("v", "p"),
    "stream_sync":    ("v", "p"),
    "graph_begin":    ("v", "p"),
    "graph_end":      ("p", "p"),
    "graph_launch":   ("v", "pp"),
    "graph_destroy":  ("v", "p"),
    "version":        ("p", ""),
    "set_error_policy": ("v", "i"),
    "last_error":     ("i", ""),
    "last_error_string": ("p", ""),
    "clear_error":    ("v", ""),
}

BATCH_SIGNATURES = {
    "alf_luma_batch":   ("v", "piipi"),
    "alf_chroma_batch": ("v", "pipi"),
    "alf_cc_batch":     ("v", "pipi"),
    "mc_batch":         ("v", "pipiii"),
    "blend_batch":      ("v", "pipiii"),
    "bdof_batch":       ("v", "pipi"),
    "sao_batch":        ("v", "pipiii"),
    "sao_ctb_batch":    ("v", "pipii"),
    "deblock_batch":    ("v", "pipi"),
    "lmcs_batch":       ("v", "pipiii"),
    "itx_batch":        ("v", "pipii"),
    "itx_shape_batch":  ("v", "pipiii"),
    "dequant_batch":    ("v", "ppi"),
    "intra_pred_batch": ("v", "pipii"),
    "cclm_batch":       ("v", "pipi"),
    "lmcs_chroma_resid_batch": ("v", "pipip"),
    "lmcs_vpdu_scale_pass": ("v", "pipp"),
    "pred_fused_batch": ("v", "pipi"),
    "bipred_batch":     ("v", "pipi"),
    "bipred_chroma_batch": ("v", "pipi"),
    "affine_batch":     ("v", "pipi"),
    "gpm_batch":        ("v", "pipi"),
    "inter_frame_build": ("v", "ppp"),
    "inter_frame_pass": ("v", "pipp"),
    "deblock_frame_pass": ("v", "pipp"),
    "sao_frame_pass":   ("v", "pipp"),
    "alf_frame_pass":   ("v", "pippp"),
    "alf_frame_build":  ("v", "pippp"),
    "alf_frame_filter": ("v", "pipp"),
    "deblock_bs_pass":  ("v", "ppp"),
    "alf_frame_work_bytes": ("z", "i"),
    "lfnst_batch":      ("v", "ppi"),
    "recon_frame_pass": ("v", "pipp"),
    "recon_state_bytes": ("z", "i"),
    "recon_order":      ("i", "piip"),
    "tab_fill_pass":    ("v", "ppp"),
    "itx_frame_build":  ("v", "ppp"),
}


def bind(lib: ctypes.CDLL, prefix: str, table: dict) -> None:
    """Attach restype/argtypes for every entry of `table` found under `prefix` in `lib`."""
    for name, (ret, args) in table.items():
        fn = getattr(lib, prefix + name)        # AttributeError = missing symbol: loud on purpose
        fn.restype = _C[ret]
        fn.argtypes = [_C[a] for a in args]


_lib = None


def load() -> ctypes.CDLL:
    """Load libvvc_mi355.so (once) and bind every declared entry point."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build the HIP extension first (__graft_entry__.build()); "
                "this package has no CPU fallback")
        lib = ctypes.CDLL(LIB_PATH)
        bind(lib, "vvc355_", SLOT_SIGNATURES)
        bind(lib, "vvc355_", RUNTIME_SIGNATURES)
        bind(lib, "vvc355_", BATCH_SIGNATURES)
        bind(lib, "vvc355_", FLAT_SIGNATURES)
        _lib = lib
    return _lib


class AlfJob(ctypes.Structure):
    """Mirror of vvc355_alf_job (include/vvc_mi355.h)."""
    _fields_ = [
        ("dst", ctypes.c_uint64), ("src", ctypes.c_uint64), ("coeff", ctypes.c_uint64),
        ("clip", ctypes.c_uint64), ("class_to_filt", ctypes.c_uint64),
        ("dst_stride", ctypes.c_int32), ("src_stride", ctypes.c_int32),
        ("w", ctypes.c_int16), ("h", ctypes.c_int16), ("vb_pos", ctypes.c_int16),
        ("ext_l", ctypes.c_int8), ("ext_r", ctypes.c_int8), ("ext_t", ctypes.c_int8), ("ext_b", ctypes.c_int8),
        ("hs", ctypes.c_int8), ("vs", ctypes.c_int8), ("pad_", ctypes.c_int8 * 4),
    ]


class McJob(ctypes.Structure):
    """Mirror of vvc355_mc_job."""
    _fields_ = [
        ("dst", ctypes.c_uint64), ("src", ctypes.c_uint64),
        ("dst_stride", ctypes.c_int32), ("src_stride", ctypes.c_int32),
        ("w", ctypes.c_int16), ("h", ctypes.c_int16),
        ("hf", ctypes.c_int8 * 8), ("vf", ctypes.c_int8 * 8),
        ("kind", ctypes.c_uint8), ("chroma", ctypes.c_uint8), ("hfrac", ctypes.c_uint8), ("vfrac", ctypes.c_uint8),
        ("denom", ctypes.c_int16), ("wx", ctypes.c_int16), ("ox", ctypes.c_int16), ("pad_", ctypes.c_int16),
    ]


class BlendJob(ctypes.Structure):
    """Mirror of vvc355_blend_job."""
    _fields_ = [
        ("dst", ctypes.c_uint64), ("src0", ctypes.c_uint64), ("src1", ctypes.c_uint64), ("aux", ctypes.c_uint64),
        ("dst_stride", ctypes.c_int32), ("src0_stride", ctypes.c_int32), ("src1_stride", ctypes.c_int32),
        ("step_x", ctypes.c_int32), ("step_y", ctypes.c_int32),
        ("w", ctypes.c_int16), ("h", ctypes.c_int16), ("mode", ctypes.c_int16), ("denom", ctypes.c_int16),
        ("w0", ctypes.c_int16), ("w1", ctypes.c_int16), ("o0", ctypes.c_int16), ("o1", ctypes.c_int16),
        ("pad_", ctypes.c_int32),
    ]


class SaoJob(ctypes.Structure):
    """Mirror of vvc355_sao_job."""
    _fields_ = [
        ("dst", ctypes.c_uint64), ("src", ctypes.c_uint64),
        ("dst_stride", ctypes.c_int32), ("src_stride", ctypes.c_int32),
        ("w", ctypes.c_int16), ("h", ctypes.c_int16), ("offset_val", ctypes.c_int16 * 5),
        ("type", ctypes.c_uint8), ("eo", ctypes.c_uint8), ("band_position", ctypes.c_uint8), ("restore", ctypes.c_uint8),
        ("borders", ctypes.c_uint8 * 4), ("vert_edge", ctypes.c_uint8 * 2), ("horiz_edge", ctypes.c_uint8 * 2),
        ("diag_edge", ctypes.c_uint8 * 4), ("pad_", ctypes.c_uint8 * 2),
    ]


class DeblockJob(ctypes.Structure):
    """Mirror of vvc355_deblock_job."""
    _fields_ = [
        ("pix", ctypes.c_uint64), ("stride", ctypes.c_int32),
        ("beta", ctypes.c_int32 * 4), ("tc", ctypes.c_int32 * 4),
        ("no_p", ctypes.c_uint8 * 4), ("no_q", ctypes.c_uint8 * 4),
        ("max_len_p", ctypes.c_uint8 * 4), ("max_len_q", ctypes.c_uint8 * 4),
        ("dir", ctypes.c_uint8), ("chroma", ctypes.c_uint8), ("flag", ctypes.c_uint8), ("pad_", ctypes.c_uint8),
    ]


class ItxJob(ctypes.Structure):
    """Mirror of vvc355_itx_job."""
    _fields_ = [
        ("coeffs", ctypes.c_uint64), ("dst", ctypes.c_uint64), ("dst_stride", ctypes.c_int32),
        ("trh", ctypes.c_uint8), ("trv", ctypes.c_uint8), ("log2_w", ctypes.c_uint8), ("log2_h", ctypes.c_uint8),
        ("nzw", ctypes.c_uint8), ("nzh", ctypes.c_uint8), ("range", ctypes.c_uint8), ("bd", ctypes.c_uint8),
        ("store_coeffs", ctypes.c_uint8), ("dq_flags", ctypes.c_uint8), ("dq_qp", ctypes.c_uint8), ("log2_matrix_size", ctypes.c_uint8),
        ("scale_matrix", ctypes.c_uint64), ("dc", ctypes.c_int16),
        ("mts_flags", ctypes.c_uint8), ("tu_flags", ctypes.c_uint8), ("mts_idx", ctypes.c_uint8), ("lfnst_idx", ctypes.c_uint8),
        ("c_idx", ctypes.c_uint8), ("pad_", ctypes.c_uint8),
    ]


class LfnstJob(ctypes.Structure):
    """Mirror of vvc355_lfnst_job."""
    _fields_ = [
        ("coeffs", ctypes.c_uint64), ("scale_matrix", ctypes.c_uint64),
        ("log2_w", ctypes.c_uint8), ("log2_h", ctypes.c_uint8), ("max_x", ctypes.c_uint8), ("max_y", ctypes.c_uint8),
        ("qp", ctypes.c_uint8), ("dequant", ctypes.c_uint8), ("dep_quant", ctypes.c_uint8), ("bit_depth", ctypes.c_uint8),
        ("range", ctypes.c_uint8), ("log2_matrix_size", ctypes.c_uint8), ("dc", ctypes.c_int16),
        ("pred_mode_intra", ctypes.c_int8), ("lfnst_idx", ctypes.c_uint8), ("pad_", ctypes.c_uint8 * 2),
    ]


class ReconCmd(ctypes.Structure):
    """Mirror of vvc355_recon_cmd (and of the oracle's orc_recon_cmd)."""
    _fields_ = [
        ("resid", ctypes.c_uint64),
        ("x0", ctypes.c_int16), ("y0", ctypes.c_int16), ("w", ctypes.c_int16), ("h", ctypes.c_int16),
        ("cu_x0", ctypes.c_int16), ("cu_y0", ctypes.c_int16), ("cb_width", ctypes.c_int16), ("cb_height", ctypes.c_int16),
        ("mode", ctypes.c_int8), ("kind", ctypes.c_uint8), ("c_idx", ctypes.c_uint8), ("ref_idx", ctypes.c_uint8),
        ("is_mip", ctypes.c_uint8), ("mip_mode", ctypes.c_uint8), ("mip_transposed", ctypes.c_uint8), ("isp_split", ctypes.c_uint8),
        ("bdpcm_flag", ctypes.c_uint8), ("joint", ctypes.c_uint8), ("pad_", ctypes.c_uint8 * 6),
    ]


class ReconCtu(ctypes.Structure):
    """Mirror of vvc355_recon_ctu."""
    _fields_ = [("first_cmd", ctypes.c_uint32), ("n_cmd", ctypes.c_uint32), ("flags", ctypes.c_uint32)]


class ReconFrame(ctypes.Structure):
    """Mirror of vvc355_recon_frame (and of the oracle's orc_recon_frame)."""
    _fields_ = [
        ("plane", ctypes.c_uint64 * 3), ("cmds", ctypes.c_uint64), ("ctus", ctypes.c_uint64), ("order", ctypes.c_uint64), ("state", ctypes.c_uint64),
        ("slice_idx", ctypes.c_uint64), ("ctb_to_col_bd", ctypes.c_uint64), ("ctb_to_row_bd", ctypes.c_uint64),
        ("stride", ctypes.c_int32 * 3), ("width", ctypes.c_int32), ("height", ctypes.c_int32), ("ctb_width", ctypes.c_int32),
        ("ctb_height", ctypes.c_int32), ("n_work", ctypes.c_int32),
        ("ctb_log2", ctypes.c_uint8), ("hs", ctypes.c_uint8), ("vs", ctypes.c_uint8), ("wpp", ctypes.c_uint8), ("collocated", ctypes.c_uint8),
        ("pad_", ctypes.c_uint8), ("workgroups", ctypes.c_uint16), ("lmcs_model", ctypes.c_uint64),
    ]


class LmcsModel(ctypes.Structure):
    """Mirror of vvc355_lmcs_model / orc_lmcs_model."""
    _fields_ = [("pivot", ctypes.c_uint16 * 17), ("chroma_scale_coeff", ctypes.c_uint16 * 16), ("min_bin_idx", ctypes.c_uint8), ("max_bin_idx", ctypes.c_uint8),
                ("pad_", ctypes.c_uint8 * 4)]


RECON_MARK, RECON_PRED, RECON_CCLM, RECON_RESID, RECON_CIIP = 0, 1, 2, 3, 4
RECON_CTU_LIGHT, RECON_CTU_LUMA_LEFT, RECON_CTU_LUMA_UP = 1, 2, 4
TU_MTS_ENABLED, TU_EXPLICIT_MTS_INTRA, TU_ISP, TU_SBT, TU_SBT_HORIZONTAL, TU_SBT_POS, TU_INTRA, TU_MIP = 1, 2, 4, 8, 16, 32, 64, 128
ITX_DERIVE_TYPE = 1


class BipredJob(ctypes.Structure):
    """Mirror of vvc355_bipred_job (and of the oracle's orc_bipred_job)."""
    _fields_ = [
        ("dst", ctypes.c_uint64), ("ref0", ctypes.c_uint64), ("ref1", ctypes.c_uint64), ("rec", ctypes.c_uint64),
        ("dst_stride", ctypes.c_int32), ("ref0_stride", ctypes.c_int32), ("ref1_stride", ctypes.c_int32),
        ("mv", ctypes.c_int32 * 4),
        ("x", ctypes.c_int16), ("y", ctypes.c_int16), ("w", ctypes.c_int16), ("h", ctypes.c_int16),
        ("pic_w", ctypes.c_int16), ("pic_h", ctypes.c_int16),
        ("denom", ctypes.c_int16), ("w0", ctypes.c_int16), ("w1", ctypes.c_int16), ("o0", ctypes.c_int16), ("o1", ctypes.c_int16),
        ("chroma", ctypes.c_uint8), ("hs", ctypes.c_uint8), ("vs", ctypes.c_uint8), ("dmvr", ctypes.c_uint8),
        ("bdof", ctypes.c_uint8), ("hf_idx", ctypes.c_uint8), ("vf_idx", ctypes.c_uint8), ("weight_flag", ctypes.c_uint8),
        ("pred_flag", ctypes.c_uint8), ("pad_", ctypes.c_uint8 * 5),
        ("lmcs_lut", ctypes.c_uint64),
    ]


class RefPic(ctypes.Structure):
    """Mirror of vvc355_ref_pic."""
    _fields_ = [("plane", ctypes.c_uint64 * 3), ("stride", ctypes.c_int32 * 3), ("pad_", ctypes.c_int32)]


class InterPu(ctypes.Structure):
    """Mirror of vvc355_inter_pu."""
    _fields_ = [("x0", ctypes.c_int16), ("y0", ctypes.c_int16), ("cb_width", ctypes.c_int16), ("cb_height", ctypes.c_int16),
                ("num_sb_x", ctypes.c_uint8), ("num_sb_y", ctypes.c_uint8), ("dmvr_flag", ctypes.c_uint8), ("bdof_flag", ctypes.c_uint8),
                ("ciip_flag", ctypes.c_uint8), ("hpel_if_idx", ctypes.c_uint8), ("slice", ctypes.c_uint8), ("pad_", ctypes.c_uint8),
                ("first_job", ctypes.c_uint32)]


class InterSlice(ctypes.Structure):
    """Mirror of vvc355_inter_slice."""
    _fields_ = [("weighted_pred", ctypes.c_uint8), ("weighted_bipred", ctypes.c_uint8), ("log2_denom", ctypes.c_uint8 * 2),
                ("lmcs_used", ctypes.c_uint8), ("pad_", ctypes.c_uint8),
                ("weight", ctypes.c_int16 * 16 * 3 * 2), ("offset", ctypes.c_int16 * 16 * 3 * 2)]


class InterFrame(ctypes.Structure):
    """Mirror of vvc355_inter_frame."""
    _fields_ = [("dst", ctypes.c_uint64 * 3), ("mvf", ctypes.c_uint64), ("refs", ctypes.c_uint64), ("pus", ctypes.c_uint64), ("slices", ctypes.c_uint64),
                ("jobs_luma", ctypes.c_uint64), ("jobs_chroma", ctypes.c_uint64), ("records", ctypes.c_uint64), ("dmvr_mvf", ctypes.c_uint64),
                ("dst_stride", ctypes.c_int32 * 3), ("mvf_stride", ctypes.c_int32), ("n_pus", ctypes.c_int32), ("n_jobs", ctypes.c_int32),
                ("width", ctypes.c_int32), ("height", ctypes.c_int32),
                ("hs", ctypes.c_uint8), ("vs", ctypes.c_uint8), ("chroma_format_idc", ctypes.c_uint8), ("pixel_shift", ctypes.c_uint8),
                ("pad_", ctypes.c_uint8 * 4), ("lmcs_fwd_lut", ctypes.c_uint64)]


class GpmJob(ctypes.Structure):
    """Mirror of vvc355_gpm_job (and of the oracle's orc_gpm_job)."""
    _fields_ = [("base", BipredJob), ("weights", ctypes.c_uint64), ("step_x", ctypes.c_int32), ("step_y", ctypes.c_int32)]


class BipredResult(ctypes.Structure):
    """Mirror of vvc355_bipred_result."""
    _fields_ = [("mv", ctypes.c_int32 * 4), ("bdof", ctypes.c_int32), ("min_sad", ctypes.c_int32),
                ("searched", ctypes.c_int32), ("pad_", ctypes.c_int32)]


class SaoCtb(ctypes.Structure):
    """Mirror of vvc355_sao_ctb."""
    _fields_ = [("offset_val", (ctypes.c_int16 * 5) * 3), ("type_idx", ctypes.c_uint8 * 3), ("band_position", ctypes.c_uint8 * 3),
                ("eo_class", ctypes.c_uint8 * 3), ("pad_", ctypes.c_uint8)]


class SaoFrame(ctypes.Structure):
    """Mirror of vvc355_sao_frame (and of the oracle's orc_sao_frame)."""
    _fields_ = [
        ("dst", ctypes.c_uint64 * 3), ("src", ctypes.c_uint64 * 3), ("sao", ctypes.c_uint64), ("slice_idx", ctypes.c_uint64),
        ("ctb_to_col_bd", ctypes.c_uint64), ("ctb_to_row_bd", ctypes.c_uint64),
        ("dst_stride", ctypes.c_int32 * 3), ("src_stride", ctypes.c_int32 * 3),
        ("width", ctypes.c_int32), ("height", ctypes.c_int32), ("ctb_width", ctypes.c_int32), ("ctb_height", ctypes.c_int32),
        ("ctb_log2", ctypes.c_uint8), ("hs", ctypes.c_uint8), ("vs", ctypes.c_uint8), ("n_comp", ctypes.c_uint8),
        ("lfase", ctypes.c_uint8), ("no_tile_filter", ctypes.c_uint8), ("pad_", ctypes.c_uint8 * 2),
    ]


class AlfCtb(ctypes.Structure):
    """Mirror of vvc355_alf_ctb."""
    _fields_ = [("ctb_flag", ctypes.c_uint8 * 3), ("filt_set_idx_y", ctypes.c_uint8), ("alt_idx", ctypes.c_uint8 * 2),
                ("cc_idc", ctypes.c_uint8 * 2)]


class AlfSlice(ctypes.Structure):
    """Mirror of vvc355_alf_slice."""
    _fields_ = [("luma_coeff", ctypes.c_uint64 * 8), ("luma_clip_idx", ctypes.c_uint64 * 8), ("chroma_coeff", ctypes.c_uint64),
                ("chroma_clip_idx", ctypes.c_uint64), ("cc_coeff", ctypes.c_uint64 * 2)]


class AlfFrame(ctypes.Structure):
    """Mirror of vvc355_alf_frame (and of the oracle's orc_alf_frame)."""
    _fields_ = [
        ("dst", ctypes.c_uint64 * 3), ("src", ctypes.c_uint64 * 3), ("alf", ctypes.c_uint64), ("slices", ctypes.c_uint64),
        ("slice_idx", ctypes.c_uint64), ("ctb_to_col_bd", ctypes.c_uint64), ("ctb_to_row_bd", ctypes.c_uint64),
        ("dst_stride", ctypes.c_int32 * 3), ("src_stride", ctypes.c_int32 * 3),
        ("width", ctypes.c_int32), ("height", ctypes.c_int32), ("ctb_width", ctypes.c_int32), ("ctb_height", ctypes.c_int32),
        ("ctb_log2", ctypes.c_uint8), ("hs", ctypes.c_uint8), ("vs", ctypes.c_uint8), ("n_comp", ctypes.c_uint8),
        ("lfase", ctypes.c_uint8), ("lfate", ctypes.c_uint8), ("pad_", ctypes.c_uint8 * 2),
    ]


class MvField(ctypes.Structure):
    """Mirror of vvc355_mvfield (= the reference's MvField, vvc_ctu.h:195-202)."""
    _fields_ = [("mv", (ctypes.c_int32 * 2) * 2), ("ref_idx", ctypes.c_int8 * 2), ("hpel_if_idx", ctypes.c_uint8),
                ("bcw_idx", ctypes.c_uint8), ("pred_flag", ctypes.c_uint8), ("ciip_flag", ctypes.c_uint8), ("pad_", ctypes.c_uint8 * 2)]


class BsFrame(ctypes.Structure):
    """Mirror of vvc355_bs_frame (and of the oracle's orc_bs_frame)."""
    _fields_ = [
        ("mvf", ctypes.c_uint64), ("ref_poc", ctypes.c_uint64), ("slice_idx", ctypes.c_uint64),
        ("ctb_to_col_bd", ctypes.c_uint64), ("ctb_to_row_bd", ctypes.c_uint64),
        ("tu_coded_flag", ctypes.c_uint64 * 3), ("tu_joint_cbcr", ctypes.c_uint64), ("pcmf", ctypes.c_uint64 * 2),
        ("tb_pos_x0", ctypes.c_uint64 * 2), ("tb_pos_y0", ctypes.c_uint64 * 2), ("tb_width", ctypes.c_uint64 * 2), ("tb_height", ctypes.c_uint64 * 2),
        ("cb_pos_x", ctypes.c_uint64), ("cb_pos_y", ctypes.c_uint64), ("cb_width", ctypes.c_uint64), ("cb_height", ctypes.c_uint64),
        ("msf", ctypes.c_uint64), ("iaf", ctypes.c_uint64),
        ("bs", (ctypes.c_uint64 * 3) * 2), ("max_len_p", ctypes.c_uint64 * 2), ("max_len_q", ctypes.c_uint64 * 2),
        ("width", ctypes.c_int32), ("height", ctypes.c_int32),
        ("min_tu_width", ctypes.c_int32), ("min_pu_width", ctypes.c_int32), ("min_cb_width", ctypes.c_int32), ("ctb_width", ctypes.c_int32),
        ("ctb_log2", ctypes.c_uint8), ("min_cb_log2", ctypes.c_uint8), ("hs", ctypes.c_uint8), ("vs", ctypes.c_uint8), ("n_comp", ctypes.c_uint8),
        ("lfase", ctypes.c_uint8), ("lfate", ctypes.c_uint8), ("pad_", ctypes.c_uint8),
    ]


class DeblockFrame(ctypes.Structure):
    """Mirror of vvc355_deblock_frame (and of the oracle's orc_deblock_frame)."""
    _fields_ = [
        ("plane", ctypes.c_uint64 * 3), ("bs", ctypes.c_uint64 * 3),
        ("max_len_p", ctypes.c_uint64), ("max_len_q", ctypes.c_uint64), ("tb_size_c", ctypes.c_uint64),
        ("qp_y", ctypes.c_uint64), ("qp_c", ctypes.c_uint64 * 2), ("db_params", ctypes.c_uint64),
        ("stride", ctypes.c_int32 * 3), ("width", ctypes.c_int32), ("height", ctypes.c_int32),
        ("min_tu_width", ctypes.c_int32), ("min_cb_width", ctypes.c_int32), ("ctb_width", ctypes.c_int32),
        ("ladf_lower_bound", ctypes.c_int32 * 5),
        ("min_cb_log2", ctypes.c_uint8), ("ctb_log2", ctypes.c_uint8), ("hs", ctypes.c_uint8), ("vs", ctypes.c_uint8),
        ("n_comp", ctypes.c_uint8), ("vertical", ctypes.c_uint8), ("qp_bd_offset", ctypes.c_uint8), ("ladf_enabled", ctypes.c_uint8),
        ("num_ladf_intervals", ctypes.c_uint8), ("ladf_lowest_qp_offset", ctypes.c_int8), ("ladf_qp_offset", ctypes.c_int8 * 4),
        ("pad_", ctypes.c_uint8 * 6),
    ]


class AffineJob(ctypes.Structure):
    """Mirror of vvc355_affine_job (and of the oracle's orc_affine_job)."""
    _fields_ = [
        ("dst", ctypes.c_uint64), ("ref0", ctypes.c_uint64), ("ref1", ctypes.c_uint64), ("diff_mv", ctypes.c_uint64),
        ("dst_stride", ctypes.c_int32), ("ref0_stride", ctypes.c_int32), ("ref1_stride", ctypes.c_int32),
        ("mv", ctypes.c_int32 * 4),
        ("x", ctypes.c_int16), ("y", ctypes.c_int16), ("pic_w", ctypes.c_int16), ("pic_h", ctypes.c_int16),
        ("denom", ctypes.c_int16), ("w0", ctypes.c_int16), ("w1", ctypes.c_int16), ("o0", ctypes.c_int16), ("o1", ctypes.c_int16),
        ("pred_flag", ctypes.c_uint8), ("prof0", ctypes.c_uint8), ("prof1", ctypes.c_uint8), ("weight_flag", ctypes.c_uint8),
        ("pad_", ctypes.c_uint8 * 6),
        ("lmcs_lut", ctypes.c_uint64),
    ]


class DequantJob(ctypes.Structure):
    """Mirror of vvc355_dequant_job."""
    _fields_ = [
        ("coeffs", ctypes.c_uint64), ("scale_matrix", ctypes.c_uint64),
        ("log2_w", ctypes.c_uint8), ("log2_h", ctypes.c_uint8), ("min_x", ctypes.c_uint8), ("min_y", ctypes.c_uint8),
        ("max_x", ctypes.c_uint8), ("max_y", ctypes.c_uint8),
        ("qp", ctypes.c_uint8), ("ts", ctypes.c_uint8), ("dep_quant", ctypes.c_uint8), ("bit_depth", ctypes.c_uint8),
        ("range", ctypes.c_uint8), ("log2_matrix_size", ctypes.c_uint8),
        ("dc", ctypes.c_int16), ("pad_", ctypes.c_uint8 * 2),
    ]


class IntraJob(ctypes.Structure):
    """Mirror of vvc355_intra_job (and of the oracle's orc_intra_job)."""
    _fields_ = [
        ("plane", ctypes.c_uint64), ("stride", ctypes.c_int32),
        ("x", ctypes.c_int16), ("y", ctypes.c_int16), ("w", ctypes.c_int16), ("h", ctypes.c_int16),
        ("mode", ctypes.c_int16), ("cb_width", ctypes.c_int16), ("cb_height", ctypes.c_int16),
        ("left_avail", ctypes.c_int16), ("top_avail", ctypes.c_int16),
        ("plane_w", ctypes.c_int16), ("plane_h", ctypes.c_int16),
        ("c_idx", ctypes.c_uint8), ("ref_idx", ctypes.c_uint8), ("is_mip", ctypes.c_uint8), ("mip_mode", ctypes.c_uint8),
        ("mip_transposed", ctypes.c_uint8), ("isp_split", ctypes.c_uint8), ("bdpcm_flag", ctypes.c_uint8),
        ("cand_up_left", ctypes.c_uint8), ("pad_", ctypes.c_uint8 * 6),
    ]


class CclmJob(ctypes.Structure):
    """Mirror of vvc355_cclm_job / orc_cclm_job."""
    _fields_ = [
        ("luma", ctypes.c_uint64), ("cb", ctypes.c_uint64), ("cr", ctypes.c_uint64),
        ("luma_stride", ctypes.c_int32), ("cb_stride", ctypes.c_int32), ("cr_stride", ctypes.c_int32),
        ("x0", ctypes.c_int16), ("y0", ctypes.c_int16), ("width", ctypes.c_int16), ("height", ctypes.c_int16),
        ("top_avail_c", ctypes.c_int16), ("left_avail_c", ctypes.c_int16),
        ("mode", ctypes.c_uint8), ("hs", ctypes.c_uint8), ("vs", ctypes.c_uint8), ("avail_t", ctypes.c_uint8),
        ("avail_l", ctypes.c_uint8), ("collocated", ctypes.c_uint8), ("ctu_boundary", ctypes.c_uint8), ("pad_", ctypes.c_uint8),
    ]


class LmcsScaleJob(ctypes.Structure):
    """Mirror of vvc355_lmcs_scale_job / orc_lmcs_scale_job."""
    _fields_ = [
        ("luma", ctypes.c_uint64), ("luma_stride", ctypes.c_int32),
        ("x_vpdu", ctypes.c_int16), ("y_vpdu", ctypes.c_int16), ("pic_w", ctypes.c_int16), ("pic_h", ctypes.c_int16),
        ("size_y", ctypes.c_int16),
        ("avail_t", ctypes.c_uint8), ("avail_l", ctypes.c_uint8), ("min_bin_idx", ctypes.c_uint8), ("max_bin_idx", ctypes.c_uint8),
        ("pivot", ctypes.c_uint16 * 17), ("chroma_scale_coeff", ctypes.c_uint16 * 16), ("pad_", ctypes.c_uint16 * 6),
    ]


class ItxTu(ctypes.Structure):
    """Mirror of vvc355_itx_tu."""
    _fields_ = [("coeff_off", ctypes.c_uint32), ("x0", ctypes.c_int16), ("y0", ctypes.c_int16),
                ("log2_w", ctypes.c_uint8), ("log2_h", ctypes.c_uint8), ("nzw", ctypes.c_uint8), ("nzh", ctypes.c_uint8),
                ("c_idx", ctypes.c_uint8), ("qp", ctypes.c_uint8), ("flags", ctypes.c_uint8), ("tr", ctypes.c_uint8)]


class ItxFrame(ctypes.Structure):
    """Mirror of vvc355_itx_frame."""
    _fields_ = [("tus", ctypes.c_uint64), ("jobs", ctypes.c_uint64), ("coeffs", ctypes.c_uint64), ("plane", ctypes.c_uint64 * 3),
                ("stride", ctypes.c_int32 * 3), ("n_tus", ctypes.c_int32),
                ("range", ctypes.c_uint8), ("bd", ctypes.c_uint8), ("pixel_shift", ctypes.c_uint8), ("tu_flags", ctypes.c_uint8),
                ("resid_jobs", ctypes.c_uint64), ("width", ctypes.c_int32), ("height", ctypes.c_int32),
                ("hs", ctypes.c_uint8), ("vs", ctypes.c_uint8), ("size_y", ctypes.c_uint8), ("pad_", ctypes.c_uint8),
                ("scale_table", ctypes.c_uint64)]


class LmcsScaleFrame(ctypes.Structure):
    """Mirror of vvc355_lmcs_scale_frame / orc_lmcs_scale_frame."""
    _fields_ = [("luma", ctypes.c_uint64), ("scale", ctypes.c_uint64), ("model", ctypes.c_uint64), ("slice_idx", ctypes.c_uint64),
                ("ctb_to_col_bd", ctypes.c_uint64), ("ctb_to_row_bd", ctypes.c_uint64),
                ("luma_stride", ctypes.c_int32), ("width", ctypes.c_int32), ("height", ctypes.c_int32), ("ctb_width", ctypes.c_int32),
                ("ctb_log2", ctypes.c_uint8), ("size_y", ctypes.c_uint8), ("pad_", ctypes.c_uint8 * 6)]


class CuRec(ctypes.Structure):
    """Mirror of vvc355_cu_rec / vvc355_tu_rec (same layout)."""
    _fields_ = [("x0", ctypes.c_int16), ("y0", ctypes.c_int16), ("w", ctypes.c_uint8), ("h", ctypes.c_uint8), ("flags", ctypes.c_uint8), ("pad_", ctypes.c_uint8)]


class MvRec(ctypes.Structure):
    """Mirror of vvc355_mv_rec."""
    _fields_ = [("x0", ctypes.c_int16), ("y0", ctypes.c_int16), ("w", ctypes.c_uint8), ("h", ctypes.c_uint8), ("pad_", ctypes.c_uint8 * 2), ("mvf", ctypes.c_int32 * 6)]


class TabFill(ctypes.Structure):
    """Mirror of vvc355_tab_fill / orc_tab_fill."""
    _fields_ = [("cu", ctypes.c_uint64), ("tu", ctypes.c_uint64), ("mv", ctypes.c_uint64),
                ("n_cu", ctypes.c_int32), ("n_tu", ctypes.c_int32), ("n_mv", ctypes.c_int32), ("unit_pitch", ctypes.c_int32), ("mvf_pitch", ctypes.c_int32),
                ("hs", ctypes.c_uint8), ("vs", ctypes.c_uint8), ("ctb_log2", ctypes.c_uint8), ("pad_", ctypes.c_uint8),
                ("ctu_first_cu", ctypes.c_uint64), ("ctu_first_tu", ctypes.c_uint64), ("ctu_first_mv", ctypes.c_uint64),
                ("width", ctypes.c_int32), ("height", ctypes.c_int32), ("ctb_width", ctypes.c_int32), ("ctb_height", ctypes.c_int32),
                ("mvf", ctypes.c_uint64),
                ("tu_coded_flag", ctypes.c_uint64 * 3), ("tu_joint_cbcr", ctypes.c_uint64), ("pcmf", ctypes.c_uint64 * 2),
                ("tb_pos_x0", ctypes.c_uint64 * 2), ("tb_pos_y0", ctypes.c_uint64 * 2), ("tb_width", ctypes.c_uint64 * 2), ("tb_height", ctypes.c_uint64 * 2),
                ("cb_pos_x", ctypes.c_uint64), ("cb_pos_y", ctypes.c_uint64), ("cb_width", ctypes.c_uint64), ("cb_height", ctypes.c_uint64),
                ("msf", ctypes.c_uint64), ("iaf", ctypes.c_uint64)]


class LmcsResidJob(ctypes.Structure):
    """Mirror of vvc355_lmcs_resid_job / orc_lmcs_resid_job."""
    _fields_ = [
        ("dst", ctypes.c_uint64), ("resid", ctypes.c_uint64), ("luma", ctypes.c_uint64),
        ("dst_stride", ctypes.c_int32), ("luma_stride", ctypes.c_int32),
        ("w", ctypes.c_int16), ("h", ctypes.c_int16), ("x_vpdu", ctypes.c_int16), ("y_vpdu", ctypes.c_int16),
        ("pic_w", ctypes.c_int16), ("pic_h", ctypes.c_int16), ("size_y", ctypes.c_int16),
        ("avail_l", ctypes.c_uint8), ("avail_t", ctypes.c_uint8), ("joint", ctypes.c_uint8), ("pad_", ctypes.c_uint8 * 7),
    ]


class PredJob(ctypes.Structure):
    """Mirror of vvc355_pred_job."""
    _fields_ = [
        ("dst", ctypes.c_uint64), ("src0", ctypes.c_uint64), ("src1", ctypes.c_uint64),
        ("dst_stride", ctypes.c_int32), ("src0_stride", ctypes.c_int32), ("src1_stride", ctypes.c_int32),
        ("hf0", ctypes.c_int8 * 8), ("vf0", ctypes.c_int8 * 8), ("hf1", ctypes.c_int8 * 8), ("vf1", ctypes.c_int8 * 8),
        ("w", ctypes.c_uint8), ("h", ctypes.c_uint8), ("chroma", ctypes.c_uint8), ("frac", ctypes.c_uint8),
        ("mode", ctypes.c_uint8), ("pad0_", ctypes.c_uint8),
        ("denom", ctypes.c_int16), ("w0", ctypes.c_int16), ("w1", ctypes.c_int16), ("o0", ctypes.c_int16), ("o1", ctypes.c_int16),
        ("pad1_", ctypes.c_int16 * 2),
    ]
