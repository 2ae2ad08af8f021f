/*
 * vvc_mi355_dsp.h — the function-pointer table the ffvvc decoder dispatches its pixel kernels through, restated
 * as plain C so the MI355X installer can be compiled without the decoder's headers.
 *
 * LAYOUT CONTRACT: VVC355DSPContext must stay member-for-member layout-compatible with the reference's
 * VVCDSPContext (libavcodec/vvc/vvcdsp.h:48-168: sub-tables inter, intra, itx, lmcs, lf, sao, alf in that order, the
 * same array extents, every member a function pointer).  Inside an FFmpeg tree the installer is compiled against
 * vvcdsp.h itself; this header exists for the standalone build and its tests.
 */
#ifndef VVC_MI355_DSP_H
#define VVC_MI355_DSP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

struct VVCLocalContext;      /* opaque here: the three context-taking intra slots are installed by the in-tree shim only */
struct SAOParams;

/* ---- slot signatures (one typedef per distinct shape) ---- */
typedef void (*vvc355_put_fn)(int16_t *dst, const uint8_t *src, ptrdiff_t src_stride, int height,
                              const int8_t *hf, const int8_t *vf, int width);
typedef void (*vvc355_put_uni_fn)(uint8_t *dst, ptrdiff_t dst_stride, const uint8_t *src, ptrdiff_t src_stride, int height,
                                  const int8_t *hf, const int8_t *vf, int width);
typedef void (*vvc355_put_uni_w_fn)(uint8_t *dst, ptrdiff_t dst_stride, const uint8_t *src, ptrdiff_t src_stride, int height,
                                    int denom, int wx, int ox, const int8_t *hf, const int8_t *vf, int width);
typedef void (*vvc355_dmvr_fn)(int16_t *dst, const uint8_t *src, ptrdiff_t src_stride, int height,
                               intptr_t mx, intptr_t my, int width);
typedef void (*vvc355_itx_fn)(int *coeffs, size_t nzw, size_t nzh, intptr_t log2_transform_range, intptr_t bit_depth);
typedef int  (*vvc355_ladf_fn)(const uint8_t *pix, ptrdiff_t stride);
typedef void (*vvc355_lf_fn)(uint8_t *pix, ptrdiff_t stride, const int32_t *beta, const int32_t *tc,
                             const uint8_t *no_p, const uint8_t *no_q, const uint8_t *max_len_p, const uint8_t *max_len_q, int flag);
typedef void (*vvc355_sao_band_fn)(uint8_t *dst, const uint8_t *src, ptrdiff_t dst_stride, ptrdiff_t src_stride,
                                   const int16_t *sao_offset_val, int sao_left_class, int width, int height);
typedef void (*vvc355_sao_edge_fn)(uint8_t *dst, const uint8_t *src, ptrdiff_t dst_stride,
                                   const int16_t *sao_offset_val, int sao_eo_class, int width, int height);
typedef void (*vvc355_sao_restore_fn)(uint8_t *dst, const uint8_t *src, ptrdiff_t dst_stride, ptrdiff_t src_stride,
                                      const struct SAOParams *sao, const int *borders, int width, int height, int c_idx,
                                      const uint8_t *vert_edge, const uint8_t *horiz_edge, const uint8_t *diag_edge);
typedef void (*vvc355_alf_filter_fn)(uint8_t *dst, ptrdiff_t dst_stride, const uint8_t *src, ptrdiff_t src_stride,
                                     int width, int height, const int16_t *filter, const int16_t *clip, int vb_pos);

/* ---- sub-tables, in the reference's member order ---- */
typedef struct VVC355InterDSP {                 /* vvcdsp.h:48-93 */
    vvc355_put_fn       put[2][7][2][2];        /* [luma, chroma][log2(width) - 1][vertical frac][horizontal frac] */
    vvc355_put_uni_fn   put_uni[2][7][2][2];
    vvc355_put_uni_w_fn put_uni_w[2][7][2][2];
    void (*avg)(uint8_t *dst, ptrdiff_t dst_stride, const int16_t *src0, const int16_t *src1, int width, int height);
    void (*w_avg)(uint8_t *dst, ptrdiff_t dst_stride, const int16_t *src0, const int16_t *src1, int width, int height,
                  int denom, int w0, int w1, int o0, int o1);
    void (*put_ciip)(uint8_t *dst, ptrdiff_t dst_stride, int width, int height,
                     const uint8_t *inter, ptrdiff_t inter_stride, int inter_weight);
    void (*put_gpm)(uint8_t *dst, ptrdiff_t dst_stride, int width, int height,
                    const int16_t *src0, const int16_t *src1, const uint8_t *weights, int step_x, int step_y);
    void (*fetch_samples)(int16_t *dst, const uint8_t *src, ptrdiff_t src_stride, int x_frac, int y_frac);
    void (*bdof_fetch_samples)(int16_t *dst, const uint8_t *src, ptrdiff_t src_stride, int x_frac, int y_frac, int width, int height);
    void (*prof_grad_filter)(int16_t *gradient_h, int16_t *gradient_v, ptrdiff_t gradient_stride,
                             const int16_t *src, ptrdiff_t src_stride, int width, int height, int pad);
    void (*apply_prof)(int16_t *dst, const int16_t *src, const int16_t *diff_mv_x, const int16_t *diff_mv_y);
    void (*apply_prof_uni)(uint8_t *dst, ptrdiff_t dst_stride, const int16_t *src, const int16_t *diff_mv_x, const int16_t *diff_mv_y);
    void (*apply_prof_uni_w)(uint8_t *dst, ptrdiff_t dst_stride, const int16_t *src,
                             const int16_t *diff_mv_x, const int16_t *diff_mv_y, int denom, int wx, int ox);
    void (*apply_bdof)(uint8_t *dst, ptrdiff_t dst_stride, int16_t *src0, int16_t *src1, int block_w, int block_h);
    int  (*sad)(const int16_t *src0, const int16_t *src1, int dx, int dy, int block_w, int block_h);
    vvc355_dmvr_fn dmvr[2][2];
} VVC355InterDSP;

typedef struct VVC355IntraDSP {                 /* vvcdsp.h:97-111 */
    void (*intra_cclm_pred)(const struct VVCLocalContext *lc, int x0, int y0, int w, int h);
    void (*lmcs_scale_chroma)(struct VVCLocalContext *lc, int *dst, const int *coeff, int w, int h, int x0_cu, int y0_cu);
    void (*intra_pred)(const struct VVCLocalContext *lc, int x0, int y0, int w, int h, int c_idx);
    void (*pred_planar)(uint8_t *src, const uint8_t *top, const uint8_t *left, int w, int h, ptrdiff_t stride);
    void (*pred_mip)(uint8_t *src, const uint8_t *top, const uint8_t *left, int w, int h, ptrdiff_t stride, int mode_id, int is_transpose);
    void (*pred_dc)(uint8_t *src, const uint8_t *top, const uint8_t *left, int w, int h, ptrdiff_t stride);
    void (*pred_v)(uint8_t *src, const uint8_t *top, int w, int h, ptrdiff_t stride);
    void (*pred_h)(uint8_t *src, const uint8_t *left, int w, int h, ptrdiff_t stride);
    void (*pred_angular_v)(uint8_t *src, const uint8_t *top, const uint8_t *left, int w, int h, ptrdiff_t stride,
                           int c_idx, int mode, int ref_idx, int filter_flag, int need_pdpc);
    void (*pred_angular_h)(uint8_t *src, const uint8_t *top, const uint8_t *left, int w, int h, ptrdiff_t stride,
                           int c_idx, int mode, int ref_idx, int filter_flag, int need_pdpc);
} VVC355IntraDSP;

typedef struct VVC355ItxDSP {                   /* vvcdsp.h:113-121 */
    void (*add_residual)(uint8_t *dst, const int *res, int width, int height, ptrdiff_t stride);
    void (*add_residual_joint)(uint8_t *dst, const int *res, int width, int height, ptrdiff_t stride, int c_sign, int shift);
    void (*pred_residual_joint)(int *buf, int width, int height, int c_sign, int shift);
    vvc355_itx_fn itx[3][3][7][7];              /* [trh][trv][log2 w][log2 h]; NULL where the reference has no entry */
    void (*transform_bdpcm)(int *coeffs, int width, int height, int vertical, int log2_transform_range);
} VVC355ItxDSP;

typedef struct VVC355LmcsDSP {                  /* vvcdsp.h:123-125 */
    void (*filter)(uint8_t *dst, ptrdiff_t dst_stride, int width, int height, const uint8_t *lut);
} VVC355LmcsDSP;

typedef struct VVC355LfDSP {                    /* vvcdsp.h:127-134; index 0 = horizontal edge, 1 = vertical edge */
    vvc355_ladf_fn ladf_level[2];
    vvc355_lf_fn   filter_luma[2];
    vvc355_lf_fn   filter_chroma[2];
} VVC355LfDSP;

typedef struct VVC355SaoDSP {                   /* vvcdsp.h:137-146 */
    vvc355_sao_band_fn    band_filter[9];
    vvc355_sao_edge_fn    edge_filter[9];
    vvc355_sao_restore_fn edge_restore[2];
} VVC355SaoDSP;

typedef struct VVC355AlfDSP {                   /* vvcdsp.h:148-158 */
    vvc355_alf_filter_fn filter[2];
    void (*filter_cc)(uint8_t *dst, ptrdiff_t dst_stride, const uint8_t *luma, ptrdiff_t luma_stride,
                      int width, int height, int hs, int vs, const int16_t *filter, int vb_pos);
    void (*classify)(int *class_idx, int *transpose_idx, const uint8_t *src, ptrdiff_t src_stride, int width, int height,
                     int vb_pos, int *gradient_tmp);
    void (*recon_coeff_and_clip)(int16_t *coeff, int16_t *clip, const int *class_idx, const int *transpose_idx, int size,
                                 const int16_t *coeff_set, const uint8_t *clip_idx_set, const uint8_t *class_to_filt);
} VVC355AlfDSP;

typedef struct VVC355DSPContext {               /* vvcdsp.h:160-168 */
    VVC355InterDSP inter;
    VVC355IntraDSP intra;
    VVC355ItxDSP   itx;
    VVC355LmcsDSP  lmcs;
    VVC355LfDSP    lf;
    VVC355SaoDSP   sao;
    VVC355AlfDSP   alf;
} VVC355DSPContext;

/*
 * The arch hook: same shape as ff_vvc_dsp_init_x86(VVCDSPContext *, int bit_depth) (vvcdsp.h:172, called last by
 * ff_vvc_dsp_init, vvcdsp.c:254-256).  Overrides every slot it has a kernel for and leaves the rest untouched:
 * the three VVCLocalContext-taking intra slots and sao.edge_restore (which take decoder structs) are installed only
 * by the in-tree variant of this file.  bit_depth must be 8, 10 or 12.
 */
void ff_vvc_dsp_init_mi355(VVC355DSPContext *c, int bit_depth);

/* test hooks of the standalone build */
int  vvc355_dsp_count_slots(const VVC355DSPContext *c);      /* number of non-NULL pointers in the table */
int  vvc355_dsp_table_selftest(int bit_depth);               /* calls slots THROUGH the table vs directly; 0 = identical (needs a GPU) */

#ifdef __cplusplus
}
#endif
#endif
