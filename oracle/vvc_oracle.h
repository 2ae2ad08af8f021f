/*
 * oracle/vvc_oracle.h — C API of the CPU oracle (TEST INFRASTRUCTURE ONLY; PARITY UNPINNED,
 * see orc_common.h).  One function per slot of the reference's VVCDSPContext
 * (libavcodec/vvc/vvcdsp.h:48-168): same argument order and meaning, with the bit depth the
 * reference selects at ff_vvc_dsp_init() time (vvcdsp.c:228) passed as a leading `bd` argument,
 * and table indices (luma/chroma, frac/int, h/v) passed as leading ints.
 */
#ifndef VVC_ORACLE_H
#define VVC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* implicit source stride of sao.edge_filter: 2*MAX_PB_SIZE + AV_INPUT_BUFFER_PADDING_SIZE bytes (vvcdsp.h:140) */
#define ORC_SAO_EDGE_SRC_STRIDE (2 * 128 + 64)

/* ---- in-loop filters (orc_filter.c) ---- */
void orc_lmcs_filter(int bd, uint8_t *dst, ptrdiff_t dst_stride, int width, int height, const uint8_t *lut);

void orc_alf_filter_luma(int bd, uint8_t *dst, ptrdiff_t dst_stride, const uint8_t *src, ptrdiff_t src_stride,
    int width, int height, const int16_t *filter, const int16_t *clip, int vb_pos);
void orc_alf_filter_chroma(int bd, uint8_t *dst, ptrdiff_t dst_stride, const uint8_t *src, ptrdiff_t src_stride,
    int width, int height, const int16_t *filter, const int16_t *clip, int vb_pos);
void orc_alf_filter_cc(int bd, uint8_t *dst, ptrdiff_t dst_stride, const uint8_t *luma, ptrdiff_t luma_stride,
    int width, int height, int hs, int vs, const int16_t *filter, int vb_pos);
void orc_alf_classify(int bd, int *class_idx, int *transpose_idx, const uint8_t *src, ptrdiff_t src_stride,
    int width, int height, int vb_pos, int *gradient_tmp);
void orc_alf_recon_coeff_and_clip(int bd, int16_t *coeff, int16_t *clip, const int *class_idx, const int *transpose_idx,
    int size, const int16_t *coeff_set, const uint8_t *clip_idx_set, const uint8_t *class_to_filt);
extern const uint8_t orc_alf_transpose_perm[4][12];

void orc_sao_band_filter(int bd, uint8_t *dst, const uint8_t *src, ptrdiff_t dst_stride, ptrdiff_t src_stride,
    const int16_t *sao_offset_val, int sao_left_class, int width, int height);
void orc_sao_edge_filter(int bd, uint8_t *dst, const uint8_t *src, ptrdiff_t dst_stride,
    const int16_t *sao_offset_val, int eo, int width, int height);
/* SAOParams flattened: offset_val = sao->offset_val[c_idx], eo_class = sao->eo_class[c_idx] */
void orc_sao_edge_restore(int bd, int variant, uint8_t *dst, const uint8_t *src, ptrdiff_t dst_stride, ptrdiff_t src_stride,
    const int16_t *offset_val, int eo_class, const int *borders, int width, int height,
    const uint8_t *vert_edge, const uint8_t *horiz_edge, const uint8_t *diag_edge);

/* dir: 0 = the [h] slot (horizontal edge), 1 = the [v] slot (vertical edge) */
void orc_lf_filter_luma(int bd, int dir, uint8_t *pix, ptrdiff_t stride, const int32_t *beta, const int32_t *tc,
    const uint8_t *no_p, const uint8_t *no_q, const uint8_t *max_len_p, const uint8_t *max_len_q, int hor_ctu_edge);
void orc_lf_filter_chroma(int bd, int dir, uint8_t *pix, ptrdiff_t stride, const int32_t *beta, const int32_t *tc,
    const uint8_t *no_p, const uint8_t *no_q, const uint8_t *max_len_p, const uint8_t *max_len_q, int shift);
int orc_lf_ladf_level(int bd, int dir, const uint8_t *pix, ptrdiff_t stride);

/* ---- inter prediction (orc_inter.c) ---- */
/* kind: 0 = put (int16 dst, stride 128), 1 = put_uni, 2 = put_uni_w.  vfrac/hfrac = the [!!my][!!mx] indices. */
void orc_put(int bd, int chroma, int vfrac, int hfrac, int16_t *dst, const uint8_t *src, ptrdiff_t src_stride,
    int height, const int8_t *hf, const int8_t *vf, int width);
void orc_put_uni(int bd, int chroma, int vfrac, int hfrac, uint8_t *dst, ptrdiff_t dst_stride,
    const uint8_t *src, ptrdiff_t src_stride, int height, const int8_t *hf, const int8_t *vf, int width);
void orc_put_uni_w(int bd, int chroma, int vfrac, int hfrac, uint8_t *dst, ptrdiff_t dst_stride,
    const uint8_t *src, ptrdiff_t src_stride, int height, int denom, int wx, int ox,
    const int8_t *hf, const int8_t *vf, int width);
void orc_avg(int bd, uint8_t *dst, ptrdiff_t dst_stride, const int16_t *src0, const int16_t *src1, int width, int height);
void orc_w_avg(int bd, uint8_t *dst, ptrdiff_t dst_stride, const int16_t *src0, const int16_t *src1, int width, int height,
    int denom, int w0, int w1, int o0, int o1);
void orc_put_ciip(int bd, uint8_t *dst, ptrdiff_t dst_stride, int width, int height,
    const uint8_t *inter, ptrdiff_t inter_stride, int intra_weight);
void orc_put_gpm(int bd, uint8_t *dst, ptrdiff_t dst_stride, int width, int height,
    const int16_t *src0, const int16_t *src1, const uint8_t *weights, int step_x, int step_y);
void orc_bdof_fetch_samples(int bd, int16_t *dst, const uint8_t *src, ptrdiff_t src_stride, int x_frac, int y_frac,
    int width, int height);
void orc_fetch_samples(int bd, int16_t *dst, const uint8_t *src, ptrdiff_t src_stride, int x_frac, int y_frac);
void orc_prof_grad_filter(int bd, int16_t *gradient_h, int16_t *gradient_v, ptrdiff_t gradient_stride,
    const int16_t *src, ptrdiff_t src_stride, int width, int height, int pad);
void orc_apply_prof(int bd, int16_t *dst, const int16_t *src, const int16_t *diff_mv_x, const int16_t *diff_mv_y);
void orc_apply_prof_uni(int bd, uint8_t *dst, ptrdiff_t dst_stride, const int16_t *src,
    const int16_t *diff_mv_x, const int16_t *diff_mv_y);
void orc_apply_prof_uni_w(int bd, uint8_t *dst, ptrdiff_t dst_stride, const int16_t *src,
    const int16_t *diff_mv_x, const int16_t *diff_mv_y, int denom, int wx, int ox);
void orc_apply_bdof(int bd, uint8_t *dst, ptrdiff_t dst_stride, int16_t *src0, int16_t *src1, int block_w, int block_h);
int  orc_sad(const int16_t *src0, const int16_t *src1, int dx, int dy, int block_w, int block_h);
void orc_dmvr(int bd, int vfrac, int hfrac, int16_t *dst, const uint8_t *src, ptrdiff_t src_stride, int height,
    intptr_t mx, intptr_t my, int width);

/* ---- inverse transform + residual (orc_itx.c) ---- */
enum { ORC_DCT2 = 0, ORC_DST7 = 1, ORC_DCT8 = 2 };
/* 1-D kernels on a strided int vector; n = 1,2,4,..,64 (DCT2) or 1,4,8,16,32 (DST7/DCT8) */
void orc_inv_tx_1d(int type, int n, int *coeffs, ptrdiff_t stride, size_t nz);
/* itx.itx[trh][trv][log2 w][log2 h]; returns 0, or -1 when the reference table has no entry for that combination */
void orc_dequant(int *coeffs, int log2_w, int log2_h, int min_x, int min_y, int max_x, int max_y, int qp, int ts,
                 int dep_quant, int bit_depth, int log2_transform_range, const uint8_t *scale_matrix,
                 int log2_matrix_size, int dc);
int  orc_itx(int trh, int trv, int log2_w, int log2_h, int *coeffs, size_t nzw, size_t nzh,
    intptr_t log2_transform_range, intptr_t bd);
void orc_inv_lfnst_1d(int *v, const int *u, int no_zero_size, int n_tr_s, int pred_mode_intra, int lfnst_idx,
    int log2_transform_range);
void orc_add_residual(int bd, uint8_t *dst, const int *res, int width, int height, ptrdiff_t stride);
void orc_add_residual_joint(int bd, uint8_t *dst, const int *res, int width, int height, ptrdiff_t stride, int c_sign, int shift);
void orc_pred_residual_joint(int *buf, int width, int height, int c_sign, int shift);
void orc_transform_bdpcm(int *coeffs, int width, int height, int vertical, int log2_transform_range);

/* ---- intra (orc_intra.c).  Leaf predictors: `stride` counts PIXELS, like the reference's POS() macro. ---- */
/* intra_pred with VVCLocalContext flattened; same layout as vvc355_intra_job (include/vvc_mi355.h), plane = HOST address */
/* ---- one regular bi-predicted sub-block incl. its callers' work (orc_inter.c, "pred_regular_blk") ----
 * Same layout as vvc355_bipred_job / vvc355_bipred_result of include/vvc_mi355.h, with host addresses. */
typedef struct orc_bipred_job {
    uint64_t dst, ref0, ref1, rec;
    int32_t  dst_stride, ref0_stride, ref1_stride;
    int32_t  mv[4];
    int16_t  x, y, w, h, pic_w, pic_h;
    int16_t  denom, w0, w1, o0, o1;
    uint8_t  chroma, hs, vs, dmvr, bdof, hf_idx, vf_idx, weight_flag;
    uint8_t  pred_flag;          /* 0 or 3: bi-prediction; 1: list 0 only; 2: list 1 only (mvf->pred_flag) */
    uint8_t  pad_[5];
    uint64_t lmcs_lut;           /* 0, or the forward luma map (lmcs.filter after predict_inter, vvc_inter.c:888-891 / on the CIIP inter part :573) */
} orc_bipred_job;
typedef struct orc_bipred_result {
    int32_t mv[4];
    int32_t bdof, min_sad, searched, pad_;
} orc_bipred_result;
void orc_bipred_block(int bd, const orc_bipred_job *job);
typedef struct orc_gpm_job {
    orc_bipred_job base;
    uint64_t weights;
    int32_t  step_x, step_y;
} orc_gpm_job;
void orc_gpm_block(int bd, const orc_gpm_job *job);

/* ---- regular inter prediction of a picture from the decoder's tables (orc_inter.c, "pred_regular_blk" over a list of coding units) ----
 * Same layouts as vvc355_ref_pic / vvc355_inter_pu / vvc355_inter_slice / vvc355_inter_frame of include/vvc_mi355.h, host addresses. */
typedef struct orc_mv_field { int32_t mv[2][2]; int8_t ref_idx[2]; uint8_t hpel_if_idx, bcw_idx, pred_flag, ciip_flag, pad_[2]; } orc_mv_field;
typedef struct orc_ref_pic { uint64_t plane[3]; int32_t stride[3]; int32_t pad_; } orc_ref_pic;
typedef struct orc_inter_pu {
    int16_t  x0, y0, cb_width, cb_height;
    uint8_t  num_sb_x, num_sb_y, dmvr_flag, bdof_flag, ciip_flag, hpel_if_idx, slice, pad_;
    uint32_t first_job;
} orc_inter_pu;
typedef struct orc_inter_slice {
    uint8_t  weighted_pred, weighted_bipred, log2_denom[2];
    uint8_t  lmcs_used, pad_;
    int16_t  weight[2][3][16], offset[2][3][16];
} orc_inter_slice;
typedef struct orc_inter_frame {
    uint64_t dst[3];
    uint64_t mvf, refs, pus, slices;
    uint64_t jobs_luma, jobs_chroma, records;
    uint64_t dmvr_mvf;
    int32_t  dst_stride[3];
    int32_t  mvf_stride;
    int32_t  n_pus, n_jobs;
    int32_t  width, height;
    uint8_t  hs, vs, chroma_format_idc, pixel_shift, pad_[4];
    uint64_t lmcs_fwd_lut;
} orc_inter_frame;
void orc_inter_frame_build(const orc_inter_frame *f);
void orc_inter_frame_pass(int bd, const orc_inter_frame *f);

/* ---- one 4x4 luma sub-block of an affine coding unit incl. PROF (orc_inter.c, "luma_prof_uni / luma_prof_bi") ----
 * Same layout as vvc355_affine_job of include/vvc_mi355.h, with host addresses. */
typedef struct orc_affine_job {
    uint64_t dst, ref0, ref1, diff_mv;
    int32_t  dst_stride, ref0_stride, ref1_stride;
    int32_t  mv[4];
    int16_t  x, y, pic_w, pic_h;
    int16_t  denom, w0, w1, o0, o1;
    uint8_t  pred_flag, prof0, prof1, weight_flag;
    uint8_t  pad_[6];
    uint64_t lmcs_lut;
} orc_affine_job;
void orc_affine_block(int bd, const orc_affine_job *job);

/* ---- one deblocking pass of a picture from the decoder's side tables (orc_filter.c, "ff_vvc_deblock_vertical/horizontal") ----
 * Same layout as vvc355_deblock_frame of include/vvc_mi355.h, with host addresses. */
typedef struct orc_deblock_frame {
    uint64_t plane[3];            /* component planes, filtered in place */
    uint64_t bs[3];               /* this pass's boundary strengths: fc->tab.vertical_bs[c] or horizontal_bs[c], uint8 per 4x4 luma unit */
    uint64_t max_len_p, max_len_q;/* luma: fc->tab.vertical_p / _q or horizontal_p / _q, uint8 per 4x4 luma unit */
    uint64_t tb_size_c;           /* chroma: fc->tab.tb_width[CHROMA] (vertical pass) or tb_height[CHROMA], uint8 per 4x4 luma unit */
    uint64_t qp_y;                /* fc->tab.qp[LUMA], int8 per minimum coding block */
    uint64_t qp_c[2];             /* fc->tab.qp[CB], [CR], int8 per 4x4 luma unit */
    uint64_t db_params;           /* fc->tab.deblock: int8 [ctb][6] = beta_offset[3], tc_offset[3] (DBParams, vvc_ps.h:89-92) */
    int32_t  stride[3];           /* bytes */
    int32_t  width, height;       /* luma picture size */
    int32_t  min_tu_width, min_cb_width, ctb_width;
    int32_t  ladf_lower_bound[5]; /* sps->ladf_interval_lower_bound */
    uint8_t  min_cb_log2, ctb_log2, hs, vs, n_comp, vertical, qp_bd_offset, ladf_enabled;
    uint8_t  num_ladf_intervals;
    int8_t   ladf_lowest_qp_offset, ladf_qp_offset[4];
    uint8_t  pad_[6];
} orc_deblock_frame;
void orc_deblock_frame_pass(int bd, const orc_deblock_frame *f);

/* ---- boundary strengths / max filter lengths of a picture (orc_filter.c, "vvc_deblock_bs"); layouts as in include/vvc_mi355.h ---- */
typedef struct orc_mvfield {             /* MvField, vvc_ctu.h:195-202 (same layout: 24 bytes) */
    int32_t mv[2][2];             /* [list][x, y] */
    int8_t  ref_idx[2];
    uint8_t hpel_if_idx, bcw_idx;
    uint8_t pred_flag;            /* PF_INTRA 0, PF_L0 1, PF_L1 2, PF_BI 3 (vvc_ctu.h:216-219) */
    uint8_t ciip_flag;
    uint8_t pad_[2];
} orc_mvfield;

typedef struct orc_bs_frame {
    /* inputs: the decoder's side tables (VVCFrameContext.tab, vvcdec.h:122-187), uploaded as they are */
    uint64_t mvf;                 /* orc_mvfield per 4x4 luma unit, row pitch min_pu_width */
    uint64_t ref_poc;             /* int32 [slice][2][32]: RefPicList.list[] (POCs) of the slice's two lists */
    uint64_t slice_idx;           /* int16 per CTB */
    uint64_t ctb_to_col_bd, ctb_to_row_bd;   /* int16 per CTB column (+1) / row (+1) */
    uint64_t tu_coded_flag[3];    /* uint8 per 4x4 luma unit */
    uint64_t tu_joint_cbcr;       /* tu_joint_cbcr_residual_flag, uint8 per 4x4 luma unit */
    uint64_t pcmf[2];             /* uint8 per 4x4 luma unit, [0] luma tree, [1] chroma tree */
    uint64_t tb_pos_x0[2], tb_pos_y0[2];     /* int32 per 4x4 luma unit, luma coordinates, per tree */
    uint64_t tb_width[2], tb_height[2];      /* uint8 per 4x4 luma unit, in samples of the component */
    uint64_t cb_pos_x, cb_pos_y;  /* int32 per minimum coding block (luma tree) */
    uint64_t cb_width, cb_height; /* uint8 per minimum coding block */
    uint64_t msf, iaf;            /* MergeSubblockFlag, InterAffineFlag: uint8 per minimum coding block */
    /* outputs, uint8 per 4x4 luma unit: [0] horizontal edges, [1] vertical edges */
    uint64_t bs[2][3];            /* fc->tab.horizontal_bs[c] / vertical_bs[c] */
    uint64_t max_len_p[2], max_len_q[2];     /* fc->tab.horizontal_p / _q, vertical_p / _q */
    int32_t  width, height;       /* luma picture size */
    int32_t  min_tu_width, min_pu_width, min_cb_width, ctb_width;
    uint8_t  ctb_log2, min_cb_log2, hs, vs, n_comp;
    uint8_t  lfase, lfate;        /* pps_loop_filter_across_slices / _tiles_enabled_flag */
    uint8_t  pad_;
} orc_bs_frame;
void orc_deblock_bs_pass(const orc_bs_frame *f);

/* ---- SAO of a picture from the decoder's per-CTB tables (orc_filter.c, "ff_vvc_sao_filter"); layouts as in include/vvc_mi355.h ---- */
typedef struct orc_sao_ctb {
    int16_t  offset_val[3][5];    /* SAOParams.offset_val */
    uint8_t  type_idx[3];         /* 0 not applied, 1 band, 2 edge (SAO_NOT_APPLIED / SAO_BAND / SAO_EDGE) */
    uint8_t  band_position[3], eo_class[3];
    uint8_t  pad_;
} orc_sao_ctb;

typedef struct orc_sao_frame {
    uint64_t dst[3], src[3];      /* post- and pre-SAO planes (the reference filters in place from saved border lines) */
    uint64_t sao;                 /* orc_sao_ctb per CTB, raster order (fc->tab.sao) */
    uint64_t slice_idx;           /* int16 per CTB (fc->tab.slice_idx) */
    uint64_t ctb_to_col_bd, ctb_to_row_bd;   /* int16 per CTB column (ctb_width + 1 entries) / row (pps->ctb_to_col_bd, _row_bd) */
    int32_t  dst_stride[3], src_stride[3];   /* bytes */
    int32_t  width, height, ctb_width, ctb_height;
    uint8_t  ctb_log2, hs, vs, n_comp;
    uint8_t  lfase;               /* pps_loop_filter_across_slices_enabled_flag */
    uint8_t  no_tile_filter;      /* num_tiles_in_pic > 1 && !pps_loop_filter_across_tiles_enabled_flag */
    uint8_t  pad_[2];
} orc_sao_frame;
void orc_sao_frame_pass(int bd, const orc_sao_frame *f);

/* ---- ALF of a picture from the decoder's per-CTB tables (orc_filter.c, "ff_vvc_alf_filter"); layouts as in include/vvc_mi355.h ---- */
typedef struct orc_alf_ctb {
    uint8_t ctb_flag[3];          /* alf_ctb_flag[] (ALFParams, vvc_ctu.h:453-459) */
    uint8_t filt_set_idx_y;       /* AlfCtbFiltSetIdxY: < 16 fixed filter sets, else 16 + index into the slice's luma APS list */
    uint8_t alt_idx[2];           /* alf_ctb_filter_alt_idx[] */
    uint8_t cc_idc[2];            /* alf_ctb_cc_cb_idc / _cr_idc */
} orc_alf_ctb;

/* what one slice signals (sh_alf_aps_id_luma[], sh_alf_aps_id_chroma, sh_alf_cc_cb / cr_aps_id resolved to the APS tables) */
typedef struct orc_alf_slice {
    uint64_t luma_coeff[8];       /* VVCALF.luma_coeff of sh_alf_aps_id_luma[k]: int16 [25][12] */
    uint64_t luma_clip_idx[8];    /* VVCALF.luma_clip_idx: uint8 [25][12] */
    uint64_t chroma_coeff;        /* VVCALF.chroma_coeff: int16 [8][6] */
    uint64_t chroma_clip_idx;     /* VVCALF.chroma_clip_idx: uint8 [8][6] */
    uint64_t cc_coeff[2];         /* VVCALF.cc_coeff[0 / 1]: int16 [4][7]; 0 = no APS */
} orc_alf_slice;

typedef struct orc_alf_frame {
    uint64_t dst[3], src[3];      /* post- and pre-ALF planes */
    uint64_t alf;                 /* orc_alf_ctb per CTB, raster order (fc->tab.alf) */
    uint64_t slices;              /* orc_alf_slice per slice */
    uint64_t slice_idx;           /* int16 per CTB (fc->tab.slice_idx) */
    uint64_t ctb_to_col_bd, ctb_to_row_bd;   /* int16 per CTB column (+1) / row (+1) */
    int32_t  dst_stride[3], src_stride[3];   /* bytes */
    int32_t  width, height, ctb_width, ctb_height;
    uint8_t  ctb_log2, hs, vs, n_comp;
    uint8_t  lfase, lfate;        /* pps_loop_filter_across_slices / _tiles_enabled_flag */
    uint8_t  pad_[2];
} orc_alf_frame;
void orc_alf_frame_pass(int bd, const orc_alf_frame *f);

typedef struct orc_intra_job {
    uint64_t plane;
    int32_t  stride;
    int16_t  x, y, w, h;
    int16_t  mode;
    int16_t  cb_width, cb_height;
    int16_t  left_avail, top_avail;
    int16_t  plane_w, plane_h;
    uint8_t  c_idx, ref_idx, is_mip, mip_mode, mip_transposed, isp_split, bdpcm_flag, cand_up_left;
    uint8_t  pad_[6];
} orc_intra_job;
void orc_intra_pred_flat(int bd, const orc_intra_job *job);
/* intra_cclm_pred / lmcs_scale_chroma flattened; same layouts as vvc355_cclm_job / vvc355_lmcs_scale_job */
typedef struct orc_cclm_job {
    uint64_t luma, cb, cr;
    int32_t  luma_stride, cb_stride, cr_stride;
    int16_t  x0, y0, width, height;
    int16_t  top_avail_c, left_avail_c;
    uint8_t  mode, hs, vs, avail_t, avail_l, collocated, ctu_boundary, pad_;
} orc_cclm_job;
typedef struct orc_lmcs_scale_job {
    uint64_t luma;
    int32_t  luma_stride;
    int16_t  x_vpdu, y_vpdu, pic_w, pic_h, size_y;
    uint8_t  avail_t, avail_l, min_bin_idx, max_bin_idx;
    uint16_t pivot[17];
    uint16_t chroma_scale_coeff[16];
    uint16_t pad_[6];
} orc_lmcs_scale_job;
void orc_intra_cclm_pred_flat(int bd, const orc_cclm_job *job);
int  orc_lmcs_chroma_scale_flat(int bd, const orc_lmcs_scale_job *job);
void orc_lmcs_scale_chroma_flat(int bd, const orc_lmcs_scale_job *job, int *dst, const int *coeff, int width, int height);
int  orc_intra_pred_angle(int mode);
int  orc_intra_inv_angle(int angle);
int  orc_intra_nscale(int w, int h, int mode);
int  orc_intra_need_pdpc(int w, int h, int bdpcm_flag, int mode, int ref_idx);
void orc_pred_planar(int bd, uint8_t *src, const uint8_t *top, const uint8_t *left, int w, int h, ptrdiff_t stride);
void orc_pred_dc(int bd, uint8_t *src, const uint8_t *top, const uint8_t *left, int w, int h, ptrdiff_t stride);
void orc_pred_v(int bd, uint8_t *src, const uint8_t *top, int w, int h, ptrdiff_t stride);
void orc_pred_h(int bd, uint8_t *src, const uint8_t *left, int w, int h, ptrdiff_t stride);
void orc_pred_angular_v(int bd, uint8_t *src, const uint8_t *top, const uint8_t *left, int w, int h, ptrdiff_t stride,
    int c_idx, int mode, int ref_idx, int filter_flag, int need_pdpc);
void orc_pred_angular_h(int bd, uint8_t *src, const uint8_t *top, const uint8_t *left, int w, int h, ptrdiff_t stride,
    int c_idx, int mode, int ref_idx, int filter_flag, int need_pdpc);
void orc_pred_mip(int bd, uint8_t *src, const uint8_t *top, const uint8_t *left, int w, int h, ptrdiff_t stride,
    int mode_id, int is_transpose);

/* ---- transform-side helpers between dequant and the inverse transform (orc_recon.c) ---- */
enum { ORC_TU_MTS_ENABLED = 1, ORC_TU_EXPLICIT_MTS_INTRA = 2, ORC_TU_ISP = 4, ORC_TU_SBT = 8, ORC_TU_SBT_HORIZONTAL = 16, ORC_TU_SBT_POS = 32,
       ORC_TU_INTRA = 64, ORC_TU_MIP = 128 };
int  orc_ilfnst_transform(int *coeffs, int w, int h, int pred_mode_intra, int lfnst_idx, int log2_transform_range);
int  orc_derive_transform_type(int flags, int mts_idx, int lfnst_idx, int c_idx, int w, int h);

/* ---- RECON of a picture from per-CTU command lists (orc_recon.c, "ff_vvc_reconstruct"); layouts as in include/vvc_mi355.h ---- */
enum { ORC_RECON_MARK = 0, ORC_RECON_PRED = 1, ORC_RECON_CCLM = 2, ORC_RECON_RESID = 3, ORC_RECON_CIIP = 4 };
typedef struct orc_recon_cmd {
    uint64_t resid;
    int16_t  x0, y0, w, h;
    int16_t  cu_x0, cu_y0, cb_width, cb_height;
    int8_t   mode;
    uint8_t  kind, c_idx, ref_idx, is_mip, mip_mode, mip_transposed, isp_split, bdpcm_flag, joint;
    uint8_t  pad_[6];
} orc_recon_cmd;
typedef struct orc_recon_ctu { uint32_t first_cmd, n_cmd, flags; } orc_recon_ctu;     /* flags: scheduling hints of the device pass, not read here */
typedef struct orc_recon_frame {
    uint64_t plane[3];
    uint64_t cmds, ctus, order, state;
    uint64_t slice_idx, ctb_to_col_bd, ctb_to_row_bd;
    int32_t  stride[3];
    int32_t  width, height, ctb_width, ctb_height, n_work;
    uint8_t  ctb_log2, hs, vs, wpp, collocated, pad_;
    uint16_t workgroups;          /* scheduling hint of the device pass; unused here */
    uint64_t lmcs_model;          /* 0, or orc_lmcs_model for RESID commands with joint bit 3 */
} orc_recon_frame;
typedef struct orc_lmcs_model {
    uint16_t pivot[17];
    uint16_t chroma_scale_coeff[16];
    uint8_t  min_bin_idx, max_bin_idx;
    uint8_t  pad_[4];
} orc_lmcs_model;
void orc_recon_frame_pass(int bd, const orc_recon_frame *f);
/* same layouts as vvc355_cu_rec / _tu_rec / _mv_rec / _tab_fill: the side tables from per-unit records (what set_cb_pos / set_cb_tab, set_tb_pos /
 * set_tb_tab and ff_vvc_set_mvf write per minimum unit, vvc_ctu.c:41-140, :1144-1250) */
typedef struct orc_cu_rec { int16_t x0, y0; uint8_t w, h, flags, pad_; } orc_cu_rec;
typedef struct orc_tu_rec { int16_t x0, y0; uint8_t w, h, flags, pad_; } orc_tu_rec;
typedef struct orc_mv_rec { int16_t x0, y0; uint8_t w, h, pad_[2]; int32_t mvf[6]; } orc_mv_rec;
typedef struct orc_tab_fill {
    uint64_t cu, tu, mv;
    int32_t  n_cu, n_tu, n_mv, unit_pitch, mvf_pitch;
    uint8_t  hs, vs, ctb_log2, pad_;
    uint64_t ctu_first_cu, ctu_first_tu, ctu_first_mv;     /* the device pass's grouping of the records per CTU; not read here */
    int32_t  width, height, ctb_width, ctb_height;
    uint64_t mvf;
    uint64_t tu_coded_flag[3], tu_joint_cbcr, pcmf[2];
    uint64_t tb_pos_x0[2], tb_pos_y0[2], tb_width[2], tb_height[2];
    uint64_t cb_pos_x, cb_pos_y, cb_width, cb_height, msf, iaf;
} orc_tab_fill;
void orc_tab_fill_pass(const orc_tab_fill *f);
/* same layout as vvc355_lmcs_resid_job: one chroma block's residual scaled (lmcs_scale_chroma) and added outside the in-order walk */
typedef struct orc_lmcs_resid_job {
    uint64_t dst, resid, luma;
    int32_t  dst_stride, luma_stride;
    int16_t  w, h, x_vpdu, y_vpdu, pic_w, pic_h, size_y;
    uint8_t  avail_l, avail_t, joint, pad_[7];
} orc_lmcs_resid_job;
void orc_lmcs_chroma_resid_block(int bd, const orc_lmcs_resid_job *job, const orc_lmcs_model *model);
/* same layout as vvc355_lmcs_scale_frame: the chroma residual scale of every 64x64 unit (lmcs_derive_chroma_scale for all units) */
typedef struct orc_lmcs_scale_frame {
    uint64_t luma, scale, model, slice_idx, ctb_to_col_bd, ctb_to_row_bd;
    int32_t  luma_stride, width, height, ctb_width;
    uint8_t  ctb_log2, size_y, pad_[6];
} orc_lmcs_scale_frame;
void orc_lmcs_vpdu_scale_pass(int bd, const orc_lmcs_scale_frame *f);
void orc_recon_debug_job(const orc_recon_frame *f, int rs, int k, orc_intra_job *out);

#ifdef __cplusplus
}
#endif
#endif
