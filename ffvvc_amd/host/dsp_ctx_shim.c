/*
 * dsp_ctx_shim.c — the four DSP slots that take decoder structs: intra.intra_pred, intra.intra_cclm_pred,
 * intra.lmcs_scale_chroma (libavcodec/vvc/vvc_intra_template.c:352-683) and sao.edge_restore[2]
 * (libavcodec/h26x/h2656_sao_template.c:81,131).  Each trampoline flattens what the slot reads through VVCLocalContext / SAOParams
 * into the POD job of include/vvc_mi355.h and calls the library's *_flat entry.
 *
 * Standalone build: compiled against include/vvc_mi355_ctx.h, a field-for-field mirror of the members read.  Inside an FFmpeg tree
 * the same file is compiled against vvc_ctu.h / vvcdec.h with VVC355_IN_TREE defined; the decoder's own ff_vvc_get_top_available /
 * _left_available / ff_vvc_wide_angle_mode_mapping are then used instead of the restatements below (they are the same functions).
 */
#include <string.h>

#include "vvc_mi355_ctx.h"

#define FFMIN(a, b) ((a) < (b) ? (a) : (b))
#define FFMAX(a, b) ((a) > (b) ? (a) : (b))

/* ------------------------------------------------------------------ availability process on the mirror (vvc_intra.c:574-648) */

static const ReconstructedArea *get_reconstructed_area(const VVCLocalContext *lc, int x, int y, int c_idx)
{
    const int ch_type = c_idx > 0;
    for (int i = lc->num_ras[ch_type] - 1; i >= 0; i--) {
        const ReconstructedArea *a = &lc->ras[ch_type][i];
        const int r = a->x + a->w, b = a->y + a->h;
        if (a->x <= x && x < r && a->y <= y && y < b)
            return a;
        if (x >= r && y >= b)
            break;
    }
    return NULL;
}

int vvc355_ctx_top_available(const VVCLocalContext *lc, int x, int y, int target_size, int c_idx)
{
    const VVCFrameContext *fc = lc->fc;
    const int hs = fc->hshift[c_idx], vs = fc->vshift[c_idx];
    const int log2_ctb_size_v = fc->ctb_log2_size_y - vs;
    const int end_of_ctb_x = ((lc->cu->x0 >> fc->ctb_log2_size_y) + 1) << fc->ctb_log2_size_y;
    const int y0b = y & ((1 << log2_ctb_size_v) - 1);
    const int max_x = FFMIN(fc->width, end_of_ctb_x) >> hs;
    const ReconstructedArea *a;
    int px = x;
    if (!y0b) {
        if (!lc->ctb_up_flag)
            return 0;
        target_size = FFMIN(target_size, (lc->end_of_tiles_x >> hs) - x);
        if (fc->sps_entropy_coding_sync_enabled_flag)
            target_size = FFMIN(target_size, (end_of_ctb_x >> hs) - x);
        return target_size;
    }
    target_size = FFMAX(0, FFMIN(target_size, max_x - x));
    while (target_size > 0 && (a = get_reconstructed_area(lc, px, y - 1, c_idx))) {
        const int sz = FFMIN(target_size, a->x + a->w - px);
        px += sz;
        target_size -= sz;
    }
    return px - x;
}

int vvc355_ctx_left_available(const VVCLocalContext *lc, int x, int y, int target_size, int c_idx)
{
    const VVCFrameContext *fc = lc->fc;
    const int hs = fc->hshift[c_idx], vs = fc->vshift[c_idx];
    const int log2_ctb_size_h = fc->ctb_log2_size_y - hs;
    const int x0b = x & ((1 << log2_ctb_size_h) - 1);
    const int end_of_ctb_y = ((lc->cu->y0 >> fc->ctb_log2_size_y) + 1) << fc->ctb_log2_size_y;
    const int max_y = FFMIN(fc->height, end_of_ctb_y) >> vs;
    const ReconstructedArea *a;
    int py = y;
    if (!x0b && !lc->ctb_left_flag)
        return 0;
    target_size = FFMAX(0, FFMIN(target_size, max_y - y));
    if (!x0b)
        return target_size;
    while (target_size > 0 && (a = get_reconstructed_area(lc, x - 1, py, c_idx))) {
        const int sz = FFMIN(target_size, a->y + a->h - py);
        py += sz;
        target_size -= sz;
    }
    return py - y;
}

static int ilog2(int v) { int r = 0; while (v > 1) { v >>= 1; r++; } return r; }

/* ff_vvc_wide_angle_mode_mapping, vvc_intra.c:693-714 */
static int wide_angle_mode_mapping(const CodingUnit *cu, int tb_width, int tb_height, int c_idx, int pred_mode_intra)
{
    const int no_isp = cu->isp_split_type == 0;
    const int nw = (no_isp || c_idx) ? tb_width : cu->cb_width, nh = (no_isp || c_idx) ? tb_height : cu->cb_height;
    const int d = ilog2(nw) - ilog2(nh), wh_ratio = d < 0 ? -d : d;
    const int max = wh_ratio > 1 ? 8 + 2 * wh_ratio : 8, min = wh_ratio > 1 ? 60 - 2 * wh_ratio : 60;
    if (nw > nh && pred_mode_intra >= 2 && pred_mode_intra < max)
        pred_mode_intra += 65;
    else if (nh > nw && pred_mode_intra <= 66 && pred_mode_intra > min)
        pred_mode_intra -= 67;
    return pred_mode_intra;
}

/* ------------------------------------------------------------------ flattening (vvc_intra_template.c:595-618, :352-366, :390-404) */

void vvc355_ctx_flatten_intra_pred(const VVCLocalContext *lc, int x0, int y0, int width, int height, int c_idx, vvc355_intra_job *j)
{
    const VVCFrameContext *fc = lc->fc;
    const CodingUnit *cu = lc->cu;
    const int hs = fc->hshift[c_idx], vs = fc->vshift[c_idx];
    const int x = x0 >> hs, y = y0 >> vs, w = width >> hs, h = height >> vs;
    const int x_cb = x0 >> fc->min_cb_log2_size_y, y_cb = y0 >> fc->min_cb_log2_size_y;
    const int at = y_cb * fc->min_cb_width + x_cb;                                   /* SAMPLE_CTB(tab, x_cb, y_cb) */
    const int pred_mode = c_idx ? cu->intra_pred_mode_c : cu->intra_pred_mode_y;
    memset(j, 0, sizeof(*j));
    j->plane = (uint64_t)(uintptr_t)fc->data[c_idx];
    j->stride = fc->linesize[c_idx];
    j->x = (int16_t)x; j->y = (int16_t)y; j->w = (int16_t)w; j->h = (int16_t)h;
    j->mode = (int16_t)wide_angle_mode_mapping(cu, w, h, c_idx, pred_mode);
    j->cb_width = (int16_t)cu->cb_width; j->cb_height = (int16_t)cu->cb_height;
    j->left_avail = (int16_t)vvc355_ctx_left_available(lc, x, y, 16384, c_idx);    /* unbounded request: the slot bounds it itself */
    j->top_avail = (int16_t)vvc355_ctx_top_available(lc, x, y, 16384, c_idx);
    j->plane_w = (int16_t)(fc->width >> hs); j->plane_h = (int16_t)(fc->height >> vs);
    j->c_idx = (uint8_t)c_idx;
    j->ref_idx = c_idx ? 0 : cu->intra_luma_ref_idx;
    j->is_mip = fc->imf[at] && (!c_idx || cu->mip_chroma_direct_flag);
    j->mip_mode = fc->imm[at];
    j->mip_transposed = fc->imtf[at];
    j->isp_split = cu->isp_split_type != 0;
    j->bdpcm_flag = cu->bdpcm_flag[c_idx];
    j->cand_up_left = (uint8_t)lc->na.cand_up_left;
}

void vvc355_ctx_flatten_cclm(const VVCLocalContext *lc, int x0, int y0, int width, int height, vvc355_cclm_job *j)
{
    const VVCFrameContext *fc = lc->fc;
    const int hs = fc->hshift[1], vs = fc->vshift[1];
    memset(j, 0, sizeof(*j));
    j->luma = (uint64_t)(uintptr_t)fc->data[0]; j->cb = (uint64_t)(uintptr_t)fc->data[1]; j->cr = (uint64_t)(uintptr_t)fc->data[2];
    j->luma_stride = fc->linesize[0]; j->cb_stride = fc->linesize[1]; j->cr_stride = fc->linesize[2];
    j->x0 = (int16_t)x0; j->y0 = (int16_t)y0; j->width = (int16_t)width; j->height = (int16_t)height;
    j->top_avail_c = (int16_t)vvc355_ctx_top_available(lc, x0 >> hs, y0 >> vs, 16384, 1);
    j->left_avail_c = (int16_t)vvc355_ctx_left_available(lc, x0 >> hs, y0 >> vs, 16384, 1);
    j->mode = (uint8_t)lc->cu->intra_pred_mode_c;
    j->hs = (uint8_t)hs; j->vs = (uint8_t)vs;
    j->avail_t = vvc355_ctx_top_available(lc, x0, y0, 1, 0) != 0;
    j->avail_l = vvc355_ctx_left_available(lc, x0, y0, 1, 0) != 0;
    j->collocated = fc->sps_chroma_vertical_collocated_flag;
    j->ctu_boundary = (y0 & ((1 << fc->ctb_log2_size_y) - 1)) == 0;
}

int vvc355_ctx_flatten_lmcs_scale(const VVCLocalContext *lc, int x0_cu, int y0_cu, vvc355_lmcs_scale_job *j)
{
    const VVCFrameContext *fc = lc->fc;
    const int size_y = FFMIN(1 << fc->ctb_log2_size_y, 64);
    const int x = x0_cu & ~(size_y - 1), y = y0_cu & ~(size_y - 1);
    if (lc->lmcs.x_vpdu == x && lc->lmcs.y_vpdu == y)
        return 1;
    memset(j, 0, sizeof(*j));
    j->luma = (uint64_t)(uintptr_t)fc->data[0];
    j->luma_stride = fc->linesize[0];
    j->x_vpdu = (int16_t)x; j->y_vpdu = (int16_t)y; j->pic_w = (int16_t)fc->width; j->pic_h = (int16_t)fc->height; j->size_y = (int16_t)size_y;
    j->avail_t = vvc355_ctx_top_available(lc, x, y, 1, 0) != 0;
    j->avail_l = vvc355_ctx_left_available(lc, x, y, 1, 0) != 0;
    j->min_bin_idx = fc->lmcs.min_bin_idx; j->max_bin_idx = fc->lmcs.max_bin_idx;
    memcpy(j->pivot, fc->lmcs.pivot, sizeof(j->pivot));
    memcpy(j->chroma_scale_coeff, fc->lmcs.chroma_scale_coeff, sizeof(j->chroma_scale_coeff));
    return 0;
}

/* ------------------------------------------------------------------ the slots */

static void intra_pred_bd(int bd, const VVCLocalContext *lc, int x0, int y0, int w, int h, int c_idx)
{
    vvc355_intra_job j;
    vvc355_ctx_flatten_intra_pred(lc, x0, y0, w, h, c_idx, &j);
    vvc355_intra_pred_flat(bd, &j);
}
static void cclm_bd(int bd, const VVCLocalContext *lc, int x0, int y0, int w, int h)
{
    vvc355_cclm_job j;
    vvc355_ctx_flatten_cclm(lc, x0, y0, w, h, &j);
    vvc355_intra_cclm_pred_flat(bd, &j, lc->fc->width, lc->fc->height);
}
static void lmcs_scale_bd(int bd, VVCLocalContext *lc, int *dst, const int *coeff, int w, int h, int x0_cu, int y0_cu)
{
    /* the scale of a VPDU does not change while the CTU is reconstructed (it reads neighbours outside the VPDU): the library derives
     * it and scales in one call; the reference's per-CTU cache (lc->lmcs) only keeps the VPDU origin so that the reset at the start of
     * every CTU (vvc_intra.c:509-510) keeps its meaning */
    vvc355_lmcs_scale_job j;
    lc->lmcs.x_vpdu = lc->lmcs.y_vpdu = -1;
    vvc355_ctx_flatten_lmcs_scale(lc, x0_cu, y0_cu, &j);
    vvc355_lmcs_scale_chroma_flat(bd, &j, dst, coeff, w, h);
}
static void edge_restore_bd(int bd, int variant, uint8_t *dst, const uint8_t *src, ptrdiff_t ds, ptrdiff_t ss, const SAOParams *sao,
                            const int *borders, int w, int h, int c_idx, const uint8_t *ve, const uint8_t *he, const uint8_t *de)
{
    vvc355_sao_edge_restore(bd, variant, dst, src, ds, ss, sao->offset_val[c_idx], sao->eo_class[c_idx], borders, w, h, ve, he, de);
}

#define CTX_SLOTS(BD)                                                                                                         \
    static void intra_pred_##BD(const VVCLocalContext *lc, int x0, int y0, int w, int h, int c) { intra_pred_bd(BD, lc, x0, y0, w, h, c); } \
    static void cclm_##BD(const VVCLocalContext *lc, int x0, int y0, int w, int h) { cclm_bd(BD, lc, x0, y0, w, h); }         \
    static void lmcs_scale_##BD(VVCLocalContext *lc, int *d, const int *c, int w, int h, int x, int y) { lmcs_scale_bd(BD, lc, d, c, w, h, x, y); } \
    static void restore0_##BD(uint8_t *d, const uint8_t *s, ptrdiff_t ds, ptrdiff_t ss, const SAOParams *sao, const int *b, int w, int h, \
                              int c, const uint8_t *ve, const uint8_t *he, const uint8_t *de)                                 \
    { edge_restore_bd(BD, 0, d, s, ds, ss, sao, b, w, h, c, ve, he, de); }                                                    \
    static void restore1_##BD(uint8_t *d, const uint8_t *s, ptrdiff_t ds, ptrdiff_t ss, const SAOParams *sao, const int *b, int w, int h, \
                              int c, const uint8_t *ve, const uint8_t *he, const uint8_t *de)                                 \
    { edge_restore_bd(BD, 1, d, s, ds, ss, sao, b, w, h, c, ve, he, de); }                                                    \
    static void install_ctx_##BD(VVC355DSPContext *c)                                                                         \
    {                                                                                                                         \
        c->intra.intra_pred = intra_pred_##BD; c->intra.intra_cclm_pred = cclm_##BD; c->intra.lmcs_scale_chroma = lmcs_scale_##BD; \
        c->sao.edge_restore[0] = restore0_##BD; c->sao.edge_restore[1] = restore1_##BD;                                       \
    }
CTX_SLOTS(8)
CTX_SLOTS(10)
CTX_SLOTS(12)

void ff_vvc_dsp_init_mi355_ctx(VVC355DSPContext *c, int bit_depth)
{
    switch (bit_depth) {
    case 12: install_ctx_12(c); break;
    case 10: install_ctx_10(c); break;
    default: install_ctx_8(c);  break;
    }
}
