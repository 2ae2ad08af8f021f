/*
 * oracle/orc_recon.c — the RECON stage of a picture as the decoder runs it per CTU, and the two transform-side helpers that sit
 * between dequant and the inverse transform (TEST INFRASTRUCTURE ONLY; PARITY UNPINNED, see orc_common.h).
 *
 *  - ilfnst_transform          libavcodec/vvc/vvc_intra.c:65-127
 *  - derive_transform_type     libavcodec/vvc/vvc_intra.c:130-164
 *  - ff_vvc_reconstruct        libavcodec/vvc/vvc_intra.c:480-527 (reconstruct, predict_intra :245-274, itransform's residual
 *                              tail :464-472, add_residual_for_joint_coding_chroma :166-186), with the availability process
 *                              ff_vvc_get_top_available / _left_available (:591-648) on the running list of reconstructed
 *                              areas (:188-206, :574-589), ff_vvc_decode_neighbour / ff_vvc_set_neighbour_available
 *                              (vvc_ctu.c:2468-2510) and the wide-angle mapping (:693-714).
 *
 * The CTU walk is flattened into a command list per CTU (what the parse stage leaves in CTU / CodingUnit / TransformUnit):
 * one command per reference call, in the reference's order.  Residuals come from the batched transform stage (the inverse
 * transform does not depend on neighbours), so a RESID command only adds.
 */
#include "orc_common.h"
#include "vvc_oracle.h"

extern const int8_t orc_tab_lfnst_8x8[4 * 2 * 16 * 48], orc_tab_lfnst_4x4[4 * 2 * 16 * 16];

/* 6.5.2 up-right diagonal scan of a 4x4 block (ff_vvc_diag_scan_x / _y [2][2], vvc_data.c:27,152) */
static const uint8_t diag4_x[16] = { 0, 0, 1, 0, 1, 2, 0, 1, 2, 3, 1, 2, 3, 2, 3, 3 };
static const uint8_t diag4_y[16] = { 0, 1, 0, 2, 1, 0, 3, 2, 1, 0, 3, 2, 1, 3, 2, 3 };

/* vvc_intra.c:65-127.  pred_mode_intra = what derive_ilfnst_pred_mode_intra (:34-62) returns (after the wide-angle mapping).
 * In place on coeffs[h][w]; returns the new max_scan + 1 (4 or 8). */
ORC_API int orc_ilfnst_transform(int *coeffs, int w, int h, int pred_mode_intra, int lfnst_idx, int log2_transform_range)
{
    const int big = w >= 8 && h >= 8;
    const int n_out = big ? 48 : 16, n_size = big ? 8 : 4;
    const int non_zero_size = ((w == 8 && h == 8) || (w == 4 && h == 4)) ? 8 : 16;
    const int transpose = pred_mode_intra > 34;
    int u[16], v[48];
    for (int x = 0; x < non_zero_size; x++)
        u[x] = coeffs[w * diag4_y[x] + diag4_x[x]];
    orc_inv_lfnst_1d(v, u, non_zero_size, n_out, pred_mode_intra, lfnst_idx, log2_transform_range);
    if (transpose) {
        if (n_size == 4) {
            for (int y = 0; y < 4; y++)
                for (int x = 0; x < 4; x++)
                    coeffs[y * w + x] = v[y + 4 * x];
        } else {
            for (int y = 0; y < 8; y++) {
                for (int x = 0; x < 4; x++)
                    coeffs[y * w + x] = v[y + 8 * x];
                if (y < 4)
                    for (int x = 4; x < 8; x++)
                        coeffs[y * w + x] = v[32 + y + 4 * (x - 4)];
            }
        }
    } else {
        const int *src = v;
        for (int y = 0; y < n_size; y++) {
            const int size = y < 4 ? n_size : 4;
            for (int x = 0; x < size; x++)
                coeffs[y * w + x] = src[x];
            src += size;
        }
    }
    return n_size;
}

/* vvc_intra.c:130-164.  flags: ORC_TU_* below; returns trh | trv << 4 (enum TxType values). */
ORC_API int orc_derive_transform_type(int flags, int mts_idx, int lfnst_idx, int c_idx, int w, int h)
{
    static const int mts_to_trh[5] = { ORC_DCT2, ORC_DST7, ORC_DCT8, ORC_DST7, ORC_DCT8 };
    static const int mts_to_trv[5] = { ORC_DCT2, ORC_DST7, ORC_DST7, ORC_DCT8, ORC_DCT8 };
    const int isp = !!(flags & ORC_TU_ISP), sbt = !!(flags & ORC_TU_SBT);
    if (c_idx || (isp && lfnst_idx))
        return ORC_DCT2 | (ORC_DCT2 << 4);
    int implicit = 0;
    if (flags & ORC_TU_MTS_ENABLED) {
        if (isp || (sbt && orc_max(w, h) <= 32) ||
            (!(flags & ORC_TU_EXPLICIT_MTS_INTRA) && (flags & ORC_TU_INTRA) && !lfnst_idx && !(flags & ORC_TU_MIP)))
            implicit = 1;
    }
    if (implicit) {
        int trh, trv;
        if (sbt) {
            const int hor = !!(flags & ORC_TU_SBT_HORIZONTAL), pos = !!(flags & ORC_TU_SBT_POS);
            trh = (hor || pos) ? ORC_DST7 : ORC_DCT8;
            trv = (!hor || pos) ? ORC_DST7 : ORC_DCT8;
        } else {
            trh = (w >= 4 && w <= 16) ? ORC_DST7 : ORC_DCT2;
            trv = (h >= 4 && h <= 16) ? ORC_DST7 : ORC_DCT2;
        }
        return trh | (trv << 4);
    }
    return mts_to_trh[mts_idx] | (mts_to_trv[mts_idx] << 4);
}

/* ------------------------------------------------------------------------------------------------ RECON of a picture */

typedef struct { int x, y, w, h; } Area;

typedef struct {
    const orc_recon_frame *f;
    int ctb_up_flag, ctb_left_flag, ctb_up_left_flag, end_of_tiles_x;
    int cu_x0, cu_y0;
    Area ras[2][1024];                      /* MAX_PARTS_IN_CTU, vvc_ctu.h:38 */
    int num_ras[2];
    struct { int x_vpdu, y_vpdu, chroma_scale; } lmcs;      /* lc->lmcs: the last 64x64 unit's scale (vvc_ctu.h:406-410) */
} Lc;

/* vvc_intra.c:188-206 */
static void add_area(Lc *lc, int ch_type, int x0, int y0, int w, int h)
{
    const int hs = ch_type ? lc->f->hs : 0, vs = ch_type ? lc->f->vs : 0;
    if (lc->num_ras[ch_type] >= 1024)
        abort();
    Area *a = &lc->ras[ch_type][lc->num_ras[ch_type]++];
    a->x = x0 >> hs; a->y = y0 >> vs; a->w = w >> hs; a->h = h >> vs;
}

/* vvc_intra.c:574-589 */
static const Area *get_area(const Lc *lc, int x, int y, int c_idx)
{
    const int ch_type = c_idx > 0;
    for (int i = lc->num_ras[ch_type] - 1; i >= 0; i--) {
        const Area *a = &lc->ras[ch_type][i];
        const int r = a->x + a->w, b = a->y + a->h;
        if (a->x <= x && x < r && a->y <= y && y < b)
            return a;
        if (x >= r && y >= b)               /* "it's too far away, no need check it" */
            break;
    }
    return NULL;
}

/* vvc_intra.c:591-620 */
static int top_available(const Lc *lc, int x, int y, int target_size, int c_idx)
{
    const orc_recon_frame *f = lc->f;
    const int hs = c_idx ? f->hs : 0, vs = c_idx ? f->vs : 0;
    const int log2_ctb_size_v = f->ctb_log2 - vs;
    const int end_of_ctb_x = ((lc->cu_x0 >> f->ctb_log2) + 1) << f->ctb_log2;
    const int y0b = y & ((1 << log2_ctb_size_v) - 1);
    const int max_x = orc_min(f->width, end_of_ctb_x) >> hs;
    const Area *a;
    int px = x;
    if (!y0b) {
        if (!lc->ctb_up_flag)
            return 0;
        target_size = orc_min(target_size, (lc->end_of_tiles_x >> hs) - x);
        if (f->wpp)
            target_size = orc_min(target_size, (end_of_ctb_x >> hs) - x);
        return target_size;
    }
    target_size = orc_max(0, orc_min(target_size, max_x - x));
    while (target_size > 0 && (a = get_area(lc, px, y - 1, c_idx))) {
        const int sz = orc_min(target_size, a->x + a->w - px);
        px += sz;
        target_size -= sz;
    }
    return px - x;
}

/* vvc_intra.c:622-648 */
static int left_available(const Lc *lc, int x, int y, int target_size, int c_idx)
{
    const orc_recon_frame *f = lc->f;
    const int hs = c_idx ? f->hs : 0, vs = c_idx ? f->vs : 0;
    const int log2_ctb_size_h = f->ctb_log2 - hs;
    const int x0b = x & ((1 << log2_ctb_size_h) - 1);
    const int end_of_ctb_y = ((lc->cu_y0 >> f->ctb_log2) + 1) << f->ctb_log2;
    const int max_y = orc_min(f->height, end_of_ctb_y) >> vs;
    const Area *a;
    int py = y;
    if (!x0b && !lc->ctb_left_flag)
        return 0;
    target_size = orc_max(0, orc_min(target_size, max_y - y));
    if (!x0b)
        return target_size;
    while (target_size > 0 && (a = get_area(lc, x - 1, py, c_idx))) {
        const int sz = orc_min(target_size, a->y + a->h - py);
        py += sz;
        target_size -= sz;
    }
    return py - y;
}

static int ilog2(int v) { int r = 0; while (v > 1) { v >>= 1; r++; } return r; }

/* vvc_intra.c:693-714 */
static int wide_angle_mode_mapping(int isp_split, int c_idx, int tb_width, int tb_height, int cb_width, int cb_height, int mode)
{
    const int nw = (!isp_split || c_idx) ? tb_width : cb_width, nh = (!isp_split || c_idx) ? tb_height : cb_height;
    const int wh_ratio = orc_abs(ilog2(nw) - ilog2(nh));
    const int max = wh_ratio > 1 ? 8 + 2 * wh_ratio : 8, min = wh_ratio > 1 ? 60 - 2 * wh_ratio : 60;
    if (nw > nh && mode >= 2 && mode < max)
        mode += 65;
    else if (nh > nw && mode <= 66 && mode > min)
        mode -= 67;
    return mode;
}

/* when set, the pass stops at command `dbg_k` of CTU `dbg_rs` (a PRED) and reports the flattened job instead of predicting */
static _Thread_local int dbg_rs = -1, dbg_k = -1;
static _Thread_local orc_intra_job *dbg_job;

ORC_API void orc_recon_frame_pass(int bd, const orc_recon_frame *f)
{
    const int wide = bd > 8;
    const orc_recon_cmd *cmds = (const orc_recon_cmd *)(uintptr_t)f->cmds;
    const orc_recon_ctu *ctus = (const orc_recon_ctu *)(uintptr_t)f->ctus;
    const int16_t *slice_idx = (const int16_t *)(uintptr_t)f->slice_idx;
    const int16_t *col_bd = (const int16_t *)(uintptr_t)f->ctb_to_col_bd, *row_bd = (const int16_t *)(uintptr_t)f->ctb_to_row_bd;
    const int ctb_size = 1 << f->ctb_log2;
    static _Thread_local Lc lc;
    lc.f = f;
    for (int ry = 0; ry < f->ctb_height; ry++)
        for (int rx = 0; rx < f->ctb_width; rx++) {
            const int rs = ry * f->ctb_width + rx;
            const orc_recon_ctu *ctu = &ctus[rs];
            if (!ctu->n_cmd)
                continue;
            /* ff_vvc_decode_neighbour, vvc_ctu.c:2468-2495 */
            const int left_tile = rx > 0 && col_bd[rx] != col_bd[rx - 1];
            const int upper_tile = ry > 0 && row_bd[ry] != row_bd[ry - 1];
            const int upper_slice = ry > 0 && slice_idx[rs] != slice_idx[rs - f->ctb_width];
            lc.end_of_tiles_x = f->width;
            if (col_bd[rx] != col_bd[rx + 1])
                lc.end_of_tiles_x = orc_min(rx * ctb_size + ctb_size, lc.end_of_tiles_x);
            lc.ctb_left_flag = rx > 0 && !left_tile;
            lc.ctb_up_flag = ry > 0 && !upper_tile && !upper_slice;
            lc.ctb_up_left_flag = lc.ctb_left_flag && lc.ctb_up_flag;
            lc.num_ras[0] = lc.num_ras[1] = 0;
            lc.lmcs.x_vpdu = lc.lmcs.y_vpdu = -1;                   /* vvc_intra.c:509-510 */
            for (uint32_t k = 0; k < ctu->n_cmd; k++) {
                const orc_recon_cmd *c = &cmds[ctu->first_cmd + k];
                lc.cu_x0 = c->cu_x0;
                lc.cu_y0 = c->cu_y0;
                if (c->kind == ORC_RECON_MARK) {
                    add_area(&lc, c->c_idx > 0, c->x0, c->y0, c->w, c->h);
                } else if (c->kind == ORC_RECON_PRED) {
                    const int c_idx = c->c_idx, hs = c_idx ? f->hs : 0, vs = c_idx ? f->vs : 0;
                    orc_intra_job j;
                    memset(&j, 0, sizeof(j));
                    j.plane = f->plane[c_idx];
                    j.stride = f->stride[c_idx];
                    j.x = c->x0 >> hs; j.y = c->y0 >> vs; j.w = c->w >> hs; j.h = c->h >> vs;
                    j.mode = (int16_t)wide_angle_mode_mapping(c->isp_split, c_idx, j.w, j.h, c->cb_width, c->cb_height, c->mode);
                    j.cb_width = c->cb_width; j.cb_height = c->cb_height;
                    j.left_avail = (int16_t)left_available(&lc, j.x, j.y, 16384, c_idx);
                    j.top_avail = (int16_t)top_available(&lc, j.x, j.y, 16384, c_idx);
                    j.plane_w = (int16_t)(f->width >> hs); j.plane_h = (int16_t)(f->height >> vs);
                    j.c_idx = (uint8_t)c_idx; j.ref_idx = c_idx ? 0 : c->ref_idx;
                    j.is_mip = c->is_mip; j.mip_mode = c->mip_mode; j.mip_transposed = c->mip_transposed;
                    j.isp_split = c->isp_split; j.bdpcm_flag = c->bdpcm_flag;
                    {   /* ff_vvc_set_neighbour_available, vvc_ctu.c:2497-2510 (luma coordinates) */
                        const int x0b = c->x0 & (ctb_size - 1), y0b = c->y0 & (ctb_size - 1);
                        const int cand_up = lc.ctb_up_flag || y0b, cand_left = lc.ctb_left_flag || x0b;
                        j.cand_up_left = (uint8_t)((x0b || y0b) ? (cand_left && cand_up) : lc.ctb_up_left_flag);
                    }
                    if (dbg_rs == rs && dbg_k == (int)k) {
                        *dbg_job = j;
                        return;
                    }
                    if (dbg_rs < 0)
                        orc_intra_pred_flat(bd, &j);
                } else if (c->kind == ORC_RECON_CCLM) {
                    orc_cclm_job j;
                    memset(&j, 0, sizeof(j));
                    j.luma = f->plane[0]; j.cb = f->plane[1]; j.cr = f->plane[2];
                    j.luma_stride = f->stride[0]; j.cb_stride = f->stride[1]; j.cr_stride = f->stride[2];
                    j.x0 = c->x0; j.y0 = c->y0; j.width = c->w; j.height = c->h;
                    j.top_avail_c = (int16_t)top_available(&lc, c->x0 >> f->hs, c->y0 >> f->vs, 16384, 1);
                    j.left_avail_c = (int16_t)left_available(&lc, c->x0 >> f->hs, c->y0 >> f->vs, 16384, 1);
                    j.mode = (uint8_t)c->mode; j.hs = f->hs; j.vs = f->vs;
                    j.avail_t = top_available(&lc, c->x0, c->y0, 1, 0) != 0;
                    j.avail_l = left_available(&lc, c->x0, c->y0, 1, 0) != 0;
                    j.collocated = f->collocated;
                    j.ctu_boundary = (c->y0 & (ctb_size - 1)) == 0;
                    if (dbg_rs < 0)
                        orc_intra_cclm_pred_flat(bd, &j);
                } else if (c->kind == ORC_RECON_CIIP) {
                    /* put_ciip (vvc_inter_template.c:60) as pred_regular_luma / _chroma call it after the intra prediction (vvc_inter.c:570-575) */
                    const int c_idx = c->c_idx, hs = c_idx ? f->hs : 0, vs = c_idx ? f->vs : 0;
                    uint8_t *dst = (uint8_t *)(uintptr_t)f->plane[c_idx] + (ptrdiff_t)(c->y0 >> vs) * f->stride[c_idx] + (((ptrdiff_t)c->x0 >> hs) << wide);
                    if (dbg_rs >= 0)
                        continue;
                    orc_put_ciip(bd, dst, f->stride[c_idx], c->w >> hs, c->h >> vs, (const uint8_t *)(uintptr_t)c->resid, (ptrdiff_t)(c->w >> hs) << wide, c->joint);
                } else if (c->kind == ORC_RECON_RESID) {
                    /* itransform's tail :464-472 / add_residual_for_joint_coding_chroma :166-186: x0, y0 = tb->x0, tb->y0 (luma
                     * coordinates), w, h = tb_width, tb_height (component samples) */
                    const int c_idx = c->c_idx, hs = c_idx ? f->hs : 0, vs = c_idx ? f->vs : 0;
                    uint8_t *dst = (uint8_t *)(uintptr_t)f->plane[c_idx] + (ptrdiff_t)(c->y0 >> vs) * f->stride[c_idx] + (((ptrdiff_t)c->x0 >> hs) << wide);
                    const int *res = (const int *)(uintptr_t)c->resid;
                    if (dbg_rs >= 0)
                        continue;
                    if (c->joint & 8) {
                        /* itransform with chroma_scale (:449-472) / add_residual_for_joint_coding_chroma (:179-183): the residual, after the
                         * joint sign / shift if any, goes through lmcs_scale_chroma with the 64x64 unit's scale */
                        static _Thread_local int tmp[64 * 64];
                        const orc_lmcs_model *m = (const orc_lmcs_model *)(uintptr_t)f->lmcs_model;
                        const int size_y = orc_min(ctb_size, 64);
                        const int xv = c->cu_x0 & ~(size_y - 1), yv = c->cu_y0 & ~(size_y - 1);
                        if (lc.lmcs.x_vpdu != xv || lc.lmcs.y_vpdu != yv) {             /* lmcs_derive_chroma_scale, vvc_intra_template.c:390-429 */
                            orc_lmcs_scale_job sj;
                            memset(&sj, 0, sizeof(sj));
                            sj.luma = f->plane[0]; sj.luma_stride = f->stride[0];
                            sj.x_vpdu = (int16_t)xv; sj.y_vpdu = (int16_t)yv; sj.pic_w = (int16_t)f->width; sj.pic_h = (int16_t)f->height; sj.size_y = (int16_t)size_y;
                            sj.avail_t = top_available(&lc, xv, yv, 1, 0) != 0;
                            sj.avail_l = left_available(&lc, xv, yv, 1, 0) != 0;
                            sj.min_bin_idx = m->min_bin_idx; sj.max_bin_idx = m->max_bin_idx;
                            memcpy(sj.pivot, m->pivot, sizeof(sj.pivot));
                            memcpy(sj.chroma_scale_coeff, m->chroma_scale_coeff, sizeof(sj.chroma_scale_coeff));
                            lc.lmcs.chroma_scale = orc_lmcs_chroma_scale_flat(bd, &sj);
                            lc.lmcs.x_vpdu = xv; lc.lmcs.y_vpdu = yv;
                        }
                        memcpy(tmp, res, sizeof(int) * c->w * c->h);
                        if (c->joint & 1)
                            orc_pred_residual_joint(tmp, c->w, c->h, (c->joint & 2) ? -1 : 1, (c->joint >> 2) & 1);
                        for (int i = 0; i < c->w * c->h; i++) {
                            const int v = orc_clip_intp2(tmp[i], bd);
                            tmp[i] = v > 0 ? (v * lc.lmcs.chroma_scale + (1 << 10)) >> 11 : -((-v * lc.lmcs.chroma_scale + (1 << 10)) >> 11);
                        }
                        orc_add_residual(bd, dst, tmp, c->w, c->h, f->stride[c_idx]);
                    } else if (c->joint & 1)
                        orc_add_residual_joint(bd, dst, res, c->w, c->h, f->stride[c_idx], (c->joint & 2) ? -1 : 1, (c->joint >> 2) & 1);
                    else
                        orc_add_residual(bd, dst, res, c->w, c->h, f->stride[c_idx]);
                } else {
                    abort();
                }
            }
        }
}

/* the job RECON derives for PRED command k of CTU rs (availability, wide-angle mode, cand_up_left): no pixel is touched */
ORC_API void orc_recon_debug_job(const orc_recon_frame *f, int rs, int k, orc_intra_job *out)
{
    dbg_rs = rs; dbg_k = k; dbg_job = out;
    orc_recon_frame_pass(10, f);
    dbg_rs = dbg_k = -1;
}

/* The tail of itransform with chroma_scale (vvc_intra.c:449-472; joint blocks :179-183) for one chroma block outside the walk: the scale
 * of the coding unit's 64x64 unit from the luma plane as it stands (lmcs_derive_chroma_scale, vvc_intra_template.c:390-429), then
 * pred_residual_joint (if joint), lmcs_scale_chroma, add_residual. */
ORC_API void orc_lmcs_chroma_resid_block(int bd, const orc_lmcs_resid_job *j, const orc_lmcs_model *m)
{
    static _Thread_local int tmp[64 * 64];
    const int n = j->w * j->h;
    const int *res = (const int *)(uintptr_t)j->resid;
    uint8_t *dst = (uint8_t *)(uintptr_t)j->dst;
    memcpy(tmp, res, sizeof(int) * n);
    if (j->joint & 1)
        orc_pred_residual_joint(tmp, j->w, j->h, (j->joint & 2) ? -1 : 1, (j->joint >> 2) & 1);
    if (j->joint & 16) {
        const int scale = *(const int16_t *)(uintptr_t)j->luma;           /* the unit's entry of orc_lmcs_vpdu_scale_pass's table */
        for (int i = 0; i < n; i++) {
            const int v = orc_clip_intp2(tmp[i], bd);
            tmp[i] = v > 0 ? (v * scale + (1 << 10)) >> 11 : -((-v * scale + (1 << 10)) >> 11);
        }
    } else if (j->joint & 8) {
        orc_lmcs_scale_job sj;
        memset(&sj, 0, sizeof(sj));
        sj.luma = j->luma; sj.luma_stride = j->luma_stride;
        sj.x_vpdu = j->x_vpdu; sj.y_vpdu = j->y_vpdu; sj.pic_w = j->pic_w; sj.pic_h = j->pic_h; sj.size_y = j->size_y;
        sj.avail_t = j->avail_t; sj.avail_l = j->avail_l;
        sj.min_bin_idx = m->min_bin_idx; sj.max_bin_idx = m->max_bin_idx;
        memcpy(sj.pivot, m->pivot, sizeof(sj.pivot));
        memcpy(sj.chroma_scale_coeff, m->chroma_scale_coeff, sizeof(sj.chroma_scale_coeff));
        const int scale = orc_lmcs_chroma_scale_flat(bd, &sj);
        for (int i = 0; i < n; i++) {
            const int v = orc_clip_intp2(tmp[i], bd);
            tmp[i] = v > 0 ? (v * scale + (1 << 10)) >> 11 : -((-v * scale + (1 << 10)) >> 11);
        }
    }
    orc_add_residual(bd, dst, tmp, j->w, j->h, j->dst_stride);
}

/* The side tables from per-unit records, one table entry per 4x4 luma unit of every record's rectangle — the loops of set_cb_pos /
 * set_cb_tab (vvc_ctu.c:124-140, :1144-1160), set_tb_pos / set_tb_tab (:41-75) and ff_vvc_set_mvf (vvc_mvs.c) with the values the call
 * sites pass (:395-400, :511, :1230-1250). */
ORC_API void orc_tab_fill_pass(const orc_tab_fill *f)
{
#define TAB(type, addr) ((type *)(uintptr_t)(addr))
    const orc_cu_rec *cu = TAB(const orc_cu_rec, f->cu);
    for (int i = 0; i < f->n_cu; i++)
        for (int y = cu[i].y0 >> 2; y < (cu[i].y0 + cu[i].h) >> 2; y++)
            for (int x = cu[i].x0 >> 2; x < (cu[i].x0 + cu[i].w) >> 2; x++) {
                const int u = y * f->unit_pitch + x;
                TAB(int, f->cb_pos_x)[u] = cu[i].x0; TAB(int, f->cb_pos_y)[u] = cu[i].y0;
                TAB(uint8_t, f->cb_width)[u] = cu[i].w; TAB(uint8_t, f->cb_height)[u] = cu[i].h;
                TAB(uint8_t, f->msf)[u] = cu[i].flags & 1; TAB(uint8_t, f->iaf)[u] = (cu[i].flags >> 1) & 1;
            }
    const orc_tu_rec *tu = TAB(const orc_tu_rec, f->tu);
    for (int i = 0; i < f->n_tu; i++) {
        const int tree = tu[i].flags >> 7;
        for (int y = tu[i].y0 >> 2; y < (tu[i].y0 + tu[i].h) >> 2; y++)
            for (int x = tu[i].x0 >> 2; x < (tu[i].x0 + tu[i].w) >> 2; x++) {
                const int u = y * f->unit_pitch + x;
                TAB(int, f->tb_pos_x0[tree])[u] = tu[i].x0; TAB(int, f->tb_pos_y0[tree])[u] = tu[i].y0;
                TAB(uint8_t, f->tb_width[tree])[u] = (uint8_t)(tree ? tu[i].w >> f->hs : tu[i].w);
                TAB(uint8_t, f->tb_height[tree])[u] = (uint8_t)(tree ? tu[i].h >> f->vs : tu[i].h);
                TAB(uint8_t, f->pcmf[tree])[u] = (tu[i].flags >> 4) & 1;
                if (!tree) {
                    TAB(uint8_t, f->tu_coded_flag[0])[u] = tu[i].flags & 1;
                } else {
                    TAB(uint8_t, f->tu_coded_flag[1])[u] = (tu[i].flags >> 1) & 1; TAB(uint8_t, f->tu_coded_flag[2])[u] = (tu[i].flags >> 2) & 1;
                    TAB(uint8_t, f->tu_joint_cbcr)[u] = (tu[i].flags >> 3) & 1;
                }
            }
    }
    const orc_mv_rec *mv = TAB(const orc_mv_rec, f->mv);
    for (int i = 0; i < f->n_mv; i++)
        for (int y = mv[i].y0 >> 2; y < (mv[i].y0 + mv[i].h) >> 2; y++)
            for (int x = mv[i].x0 >> 2; x < (mv[i].x0 + mv[i].w) >> 2; x++)
                memcpy(TAB(uint8_t, f->mvf) + (size_t)(y * f->mvf_pitch + x) * 24, mv[i].mvf, 24);
#undef TAB
}

/* lmcs_derive_chroma_scale (vvc_intra_template.c:390-429) for every 64x64 unit of the picture; a unit's first sample has a left / upper
 * neighbour inside its CTU always (an earlier unit), at the CTU's edge when ff_vvc_decode_neighbour (vvc_ctu.c:2468-2495) sets
 * ctb_left_flag / ctb_up_flag — what ff_vvc_get_left / top_available(lc, x, y, 1, 0) return there */
ORC_API void orc_lmcs_vpdu_scale_pass(int bd, const orc_lmcs_scale_frame *f)
{
    const orc_lmcs_model *m = (const orc_lmcs_model *)(uintptr_t)f->model;
    const int16_t *slice_idx = (const int16_t *)(uintptr_t)f->slice_idx;
    const int16_t *col_bd = (const int16_t *)(uintptr_t)f->ctb_to_col_bd, *row_bd = (const int16_t *)(uintptr_t)f->ctb_to_row_bd;
    int16_t *out = (int16_t *)(uintptr_t)f->scale;
    const int size = f->size_y, ux = (f->width + size - 1) / size, uy = (f->height + size - 1) / size, ctb = 1 << f->ctb_log2;
    for (int vy = 0; vy < uy; vy++)
        for (int vx = 0; vx < ux; vx++) {
            const int x = vx * size, y = vy * size, rx = x >> f->ctb_log2, ry = y >> f->ctb_log2, rs = ry * f->ctb_width + rx;
            orc_lmcs_scale_job sj;
            memset(&sj, 0, sizeof(sj));
            sj.luma = f->luma; sj.luma_stride = f->luma_stride;
            sj.x_vpdu = (int16_t)x; sj.y_vpdu = (int16_t)y; sj.pic_w = (int16_t)f->width; sj.pic_h = (int16_t)f->height; sj.size_y = (int16_t)size;
            sj.avail_l = (x & (ctb - 1)) ? 1 : (rx > 0 && col_bd[rx] == col_bd[rx - 1]);
            sj.avail_t = (y & (ctb - 1)) ? 1 : (ry > 0 && row_bd[ry] == row_bd[ry - 1] && slice_idx[rs] == slice_idx[rs - f->ctb_width]);
            sj.min_bin_idx = m->min_bin_idx; sj.max_bin_idx = m->max_bin_idx;
            memcpy(sj.pivot, m->pivot, sizeof(sj.pivot));
            memcpy(sj.chroma_scale_coeff, m->chroma_scale_coeff, sizeof(sj.chroma_scale_coeff));
            out[vy * ux + vx] = (int16_t)orc_lmcs_chroma_scale_flat(bd, &sj);
        }
}
