/*
 * dsp_ctx_shim.c — the four DSP slots that take decoder structs: intra.intra_pred, intra.intra_cclm_pred,
 * intra.lmcs_scale_chroma (libavcodec/vvc/vvc_intra_template.c:352-683) and sao.edge_restore[2]
 * (libavcodec/h26x/h2656_sao_template.c:81,131).  Each trampoline flattens what the slot reads through VVCLocalContext / SAOParams
 * into the POD job of include/vvc_mi355.h and calls the library's *_flat entry.
 *
 * Every decoder member is read through the FC_* / LC_* accessors below.  The standalone build (what this repository compiles and
 * tests) resolves them against include/vvc_mi355_ctx.h, a mirror holding exactly those members.  With VVC355_IN_TREE defined they
 * resolve against the decoder's own headers (vvc_ctu.h, vvcdec.h, vvc_ps.h); that branch cannot be compiled in this repository's
 * environment (the headers need FFmpeg's configure output) and is the field mapping a maintainer starts from, not a tested build.
 *
 * Reference-sample availability (what ff_vvc_get_top_available / ff_vvc_get_left_available answer, vvc_intra.c:591-648) goes through
 * two function pointers.  In-tree the installer is handed the decoder's own two functions (vvc355_ctx_set_availability); the
 * standalone build answers from a coverage mask of the line next to the block, built from the CTU's list of reconstructed areas.
 */
#include <string.h>

#ifdef VVC355_IN_TREE
#include "libavcodec/vvc/vvc_ctu.h"
#include "libavcodec/vvc/vvcdec.h"
#include "libavcodec/vvc/vvc_intra.h"
#include "vvc_mi355.h"
#include "vvc_mi355_dsp.h"
#define FC_DATA(fc, c)        ((fc)->frame->data[c])
#define FC_LINESIZE(fc, c)    ((fc)->frame->linesize[c])
#define FC_WIDTH(fc)          ((fc)->ps.pps->width)
#define FC_HEIGHT(fc)         ((fc)->ps.pps->height)
#define FC_HSHIFT(fc, c)      ((fc)->ps.sps->hshift[c])
#define FC_VSHIFT(fc, c)      ((fc)->ps.sps->vshift[c])
#define FC_CTB_LOG2(fc)       ((fc)->ps.sps->ctb_log2_size_y)
#define FC_MIN_CB_LOG2(fc)    ((fc)->ps.sps->min_cb_log2_size_y)
#define FC_MIN_CB_WIDTH(fc)   ((fc)->ps.pps->min_cb_width)
#define FC_WPP(fc)            ((fc)->ps.sps->r->sps_entropy_coding_sync_enabled_flag)
#define FC_COLLOCATED(fc)     ((fc)->ps.sps->r->sps_chroma_vertical_collocated_flag)
#define FC_IMF(fc)            ((fc)->tab.imf)
#define FC_IMM(fc)            ((fc)->tab.imm)
#define FC_IMTF(fc)           ((fc)->tab.imtf)
#define FC_LMCS(fc)           (&(fc)->ps.lmcs)
typedef int (*vvc355_avail_fn)(const VVCLocalContext *lc, int x, int y, int target_size, int c_idx);
#else
#include "vvc_mi355_ctx.h"
#define FC_DATA(fc, c)        ((fc)->data[c])
#define FC_LINESIZE(fc, c)    ((fc)->linesize[c])
#define FC_WIDTH(fc)          ((fc)->width)
#define FC_HEIGHT(fc)         ((fc)->height)
#define FC_HSHIFT(fc, c)      ((fc)->hshift[c])
#define FC_VSHIFT(fc, c)      ((fc)->vshift[c])
#define FC_CTB_LOG2(fc)       ((fc)->ctb_log2_size_y)
#define FC_MIN_CB_LOG2(fc)    ((fc)->min_cb_log2_size_y)
#define FC_MIN_CB_WIDTH(fc)   ((fc)->min_cb_width)
#define FC_WPP(fc)            ((fc)->sps_entropy_coding_sync_enabled_flag)
#define FC_COLLOCATED(fc)     ((fc)->sps_chroma_vertical_collocated_flag)
#define FC_IMF(fc)            ((fc)->imf)
#define FC_IMM(fc)            ((fc)->imm)
#define FC_IMTF(fc)           ((fc)->imtf)
#define FC_LMCS(fc)           (&(fc)->lmcs)
#endif

/* ------------------------------------------------------------------ reference-sample availability
 *
 * How many of `want` reference samples next to a block exist already: along the row above it starting at (x, y - 1), or down the
 * column left of it starting at (x - 1, y); coordinates in samples of component c_idx.  Three cases per direction:
 *   - the line lies in the neighbouring CTU (the block touches the CTU's top / left edge): that CTU is complete if it may be used at
 *     all (lc->ctb_up_flag / ctb_left_flag); above, the run also stops at the end of the tile and, with wavefront entry points, at
 *     the end of the CTU above (the one further right is not decoded yet);
 *   - otherwise the line lies in this CTU: a sample exists if one of the areas reconstructed so far covers it.  The areas are
 *     marked into a bit mask of the line (bit i = sample i of the CTU's row / column) and the run is the count of consecutive set
 *     bits from the block's position on, cut at the picture edge.
 */
typedef struct LineMask { uint64_t bits[2]; } LineMask;           /* a CTU is at most 128 samples wide / tall */

static void mask_set(LineMask *m, int from, int to)               /* [from, to) */
{
    for (int i = from < 0 ? 0 : from; i < to && i < 128; i++)
        m->bits[i >> 6] |= 1ull << (i & 63);
}

static int mask_run(const LineMask *m, int from, int limit)       /* consecutive set bits from `from`, at most `limit` of them */
{
    int n = 0;
    while (n < limit && from + n < 128 && ((m->bits[(from + n) >> 6] >> ((from + n) & 63)) & 1))
        n++;
    return n;
}

/* coverage of row `line` (horizontal != 0) or column `line` of the CTU whose origin in this component's samples is (ox, oy) */
static LineMask covered(const VVCLocalContext *lc, int ch_type, int horizontal, int line, int ox, int oy)
{
    LineMask m = { { 0, 0 } };
    for (int i = 0; i < lc->num_ras[ch_type]; i++) {
        const ReconstructedArea *r = &lc->ras[ch_type][i];
        if (horizontal) {
            if (r->y <= line && line < r->y + r->h)
                mask_set(&m, r->x - ox, r->x + r->w - ox);
        } else if (r->x <= line && line < r->x + r->w) {
            mask_set(&m, r->y - oy, r->y + r->h - oy);
        }
    }
    return m;
}

static int imin(int a, int b) { return a < b ? a : b; }

static int mask_top_available(const VVCLocalContext *lc, int x, int y, int want, int c_idx)
{
    const VVCFrameContext *fc = lc->fc;
    const int hs = FC_HSHIFT(fc, c_idx), vs = FC_VSHIFT(fc, c_idx), log2 = FC_CTB_LOG2(fc);
    const int ctu_x = (lc->cu->x0 >> log2) << log2, ctu_y = (lc->cu->y0 >> log2) << log2;       /* luma samples */
    const int next_ctu_x = ctu_x + (1 << log2);
    if ((y & ((1 << (log2 - vs)) - 1)) == 0) {                     /* the row above belongs to the CTU above */
        if (!lc->ctb_up_flag)
            return 0;
        want = imin(want, (lc->end_of_tiles_x >> hs) - x);
        if (FC_WPP(fc))
            want = imin(want, (next_ctu_x >> hs) - x);
        return want;
    }
    const int room = (imin(FC_WIDTH(fc), next_ctu_x) >> hs) - x;   /* samples up to the CTU's (or the picture's) right edge */
    if (want > room) want = room;
    if (want <= 0)
        return 0;
    const LineMask m = covered(lc, c_idx > 0, 1, y - 1, ctu_x >> hs, ctu_y >> vs);
    return mask_run(&m, x - (ctu_x >> hs), want);
}

static int mask_left_available(const VVCLocalContext *lc, int x, int y, int want, int c_idx)
{
    const VVCFrameContext *fc = lc->fc;
    const int hs = FC_HSHIFT(fc, c_idx), vs = FC_VSHIFT(fc, c_idx), log2 = FC_CTB_LOG2(fc);
    const int ctu_x = (lc->cu->x0 >> log2) << log2, ctu_y = (lc->cu->y0 >> log2) << log2;
    const int at_ctu_edge = (x & ((1 << (log2 - hs)) - 1)) == 0;
    if (at_ctu_edge && !lc->ctb_left_flag)
        return 0;
    const int room = (imin(FC_HEIGHT(fc), ctu_y + (1 << log2)) >> vs) - y;
    if (want > room) want = room;
    if (want <= 0)
        return 0;
    if (at_ctu_edge)                                               /* the column belongs to the CTU on the left: complete */
        return want;
    const LineMask m = covered(lc, c_idx > 0, 0, x - 1, ctu_x >> hs, ctu_y >> vs);
    return mask_run(&m, y - (ctu_y >> vs), want);
}

static vvc355_avail_fn g_top_available = mask_top_available, g_left_available = mask_left_available;

/* in-tree: vvc355_ctx_set_availability(ff_vvc_get_top_available, ff_vvc_get_left_available) before the first slot call; NULL
 * restores the shim's own derivation */
void vvc355_ctx_set_availability(vvc355_avail_fn top, vvc355_avail_fn left)
{
    g_top_available = top ? top : mask_top_available;
    g_left_available = left ? left : mask_left_available;
}

int vvc355_ctx_top_available(const VVCLocalContext *lc, int x, int y, int target_size, int c_idx)
{
    return g_top_available(lc, x, y, target_size, c_idx);
}

int vvc355_ctx_left_available(const VVCLocalContext *lc, int x, int y, int target_size, int c_idx)
{
    return g_left_available(lc, x, y, target_size, c_idx);
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
    const int hs = FC_HSHIFT(fc, c_idx), vs = FC_VSHIFT(fc, c_idx);
    const int x = x0 >> hs, y = y0 >> vs, w = width >> hs, h = height >> vs;
    const int x_cb = x0 >> FC_MIN_CB_LOG2(fc), y_cb = y0 >> FC_MIN_CB_LOG2(fc);
    const int at = y_cb * FC_MIN_CB_WIDTH(fc) + x_cb;                                   /* SAMPLE_CTB(tab, x_cb, y_cb) */
    const int pred_mode = c_idx ? cu->intra_pred_mode_c : cu->intra_pred_mode_y;
    memset(j, 0, sizeof(*j));
    j->plane = (uint64_t)(uintptr_t)FC_DATA(fc, c_idx);
    j->stride = FC_LINESIZE(fc, c_idx);
    j->x = (int16_t)x; j->y = (int16_t)y; j->w = (int16_t)w; j->h = (int16_t)h;
    j->mode = (int16_t)wide_angle_mode_mapping(cu, w, h, c_idx, pred_mode);
    j->cb_width = (int16_t)cu->cb_width; j->cb_height = (int16_t)cu->cb_height;
    j->left_avail = (int16_t)vvc355_ctx_left_available(lc, x, y, 16384, c_idx);    /* unbounded request: the slot bounds it itself */
    j->top_avail = (int16_t)vvc355_ctx_top_available(lc, x, y, 16384, c_idx);
    j->plane_w = (int16_t)(FC_WIDTH(fc) >> hs); j->plane_h = (int16_t)(FC_HEIGHT(fc) >> vs);
    j->c_idx = (uint8_t)c_idx;
    j->ref_idx = c_idx ? 0 : cu->intra_luma_ref_idx;
    j->is_mip = FC_IMF(fc)[at] && (!c_idx || cu->mip_chroma_direct_flag);
    j->mip_mode = FC_IMM(fc)[at];
    j->mip_transposed = FC_IMTF(fc)[at];
    j->isp_split = cu->isp_split_type != 0;
    j->bdpcm_flag = cu->bdpcm_flag[c_idx];
    j->cand_up_left = (uint8_t)lc->na.cand_up_left;
}

void vvc355_ctx_flatten_cclm(const VVCLocalContext *lc, int x0, int y0, int width, int height, vvc355_cclm_job *j)
{
    const VVCFrameContext *fc = lc->fc;
    const int hs = FC_HSHIFT(fc, 1), vs = FC_VSHIFT(fc, 1);
    memset(j, 0, sizeof(*j));
    j->luma = (uint64_t)(uintptr_t)FC_DATA(fc, 0); j->cb = (uint64_t)(uintptr_t)FC_DATA(fc, 1); j->cr = (uint64_t)(uintptr_t)FC_DATA(fc, 2);
    j->luma_stride = FC_LINESIZE(fc, 0); j->cb_stride = FC_LINESIZE(fc, 1); j->cr_stride = FC_LINESIZE(fc, 2);
    j->x0 = (int16_t)x0; j->y0 = (int16_t)y0; j->width = (int16_t)width; j->height = (int16_t)height;
    j->top_avail_c = (int16_t)vvc355_ctx_top_available(lc, x0 >> hs, y0 >> vs, 16384, 1);
    j->left_avail_c = (int16_t)vvc355_ctx_left_available(lc, x0 >> hs, y0 >> vs, 16384, 1);
    j->mode = (uint8_t)lc->cu->intra_pred_mode_c;
    j->hs = (uint8_t)hs; j->vs = (uint8_t)vs;
    j->avail_t = vvc355_ctx_top_available(lc, x0, y0, 1, 0) != 0;
    j->avail_l = vvc355_ctx_left_available(lc, x0, y0, 1, 0) != 0;
    j->collocated = FC_COLLOCATED(fc);
    j->ctu_boundary = (y0 & ((1 << FC_CTB_LOG2(fc)) - 1)) == 0;
}

void vvc355_ctx_flatten_lmcs_scale(const VVCLocalContext *lc, int x0_cu, int y0_cu, vvc355_lmcs_scale_job *j)
{
    const VVCFrameContext *fc = lc->fc;
    const int size_y = imin(1 << FC_CTB_LOG2(fc), 64);
    const int x = x0_cu & ~(size_y - 1), y = y0_cu & ~(size_y - 1);
    memset(j, 0, sizeof(*j));
    j->luma = (uint64_t)(uintptr_t)FC_DATA(fc, 0);
    j->luma_stride = FC_LINESIZE(fc, 0);
    j->x_vpdu = (int16_t)x; j->y_vpdu = (int16_t)y; j->pic_w = (int16_t)FC_WIDTH(fc); j->pic_h = (int16_t)FC_HEIGHT(fc); j->size_y = (int16_t)size_y;
    j->avail_t = vvc355_ctx_top_available(lc, x, y, 1, 0) != 0;
    j->avail_l = vvc355_ctx_left_available(lc, x, y, 1, 0) != 0;
    j->min_bin_idx = FC_LMCS(fc)->min_bin_idx; j->max_bin_idx = FC_LMCS(fc)->max_bin_idx;
    memcpy(j->pivot, FC_LMCS(fc)->pivot, sizeof(j->pivot));
    memcpy(j->chroma_scale_coeff, FC_LMCS(fc)->chroma_scale_coeff, sizeof(j->chroma_scale_coeff));
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
    vvc355_intra_cclm_pred_flat(bd, &j, FC_WIDTH(lc->fc), FC_HEIGHT(lc->fc));
}
static void lmcs_scale_bd(int bd, VVCLocalContext *lc, int *dst, const int *coeff, int w, int h, int x0_cu, int y0_cu)
{
    /* The reference caches a VPDU's scale in lc->lmcs between the transform blocks of a CTU (vvc_intra_template.c:390-404).  This slot
     * derives it on every call instead — the samples it reads (the VPDU's left / upper neighbours) do not change while the CTU is
     * reconstructed, so the value is the same — and leaves lc->lmcs untouched. */
    vvc355_lmcs_scale_job j;
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
