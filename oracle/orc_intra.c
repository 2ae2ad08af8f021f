/*
 * oracle/orc_intra.c — CPU restatement of the intra-prediction DSP slots.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see orc_common.h).
 *
 * Follows, by reading:
 *   libavcodec/vvc/vvc_intra_template.c  (ref_filter :450, prepare_intra_edge_params :467, intra_pred :595, planar :686,
 *                                         MIP :708-824, DC :826-864, V/H :866-887, angular :894-1001, CCLM :29-388,
 *                                         LMCS chroma scaling :390-446)
 *   libavcodec/vvc/vvc_intra.c           (mip size id :529, nscale :538, need_pdpc :557, ref filter modes :655,
 *                                         angle tables :661-690)
 * Leaf predictors keep the reference's convention that `stride` counts PIXELS (the POS() macro, :27).
 */
#include "vvc_oracle.h"
#include "orc_common.h"

extern const int8_t orc_tab_intra_luma_filter[2 * 32 * 4];
extern const uint8_t orc_tab_mip_matrix_4x4[16 * 16 * 4], orc_tab_mip_matrix_8x8[8 * 16 * 8], orc_tab_mip_matrix_16x16[6 * 64 * 7];

#define PXL(p, i) orc_ld((const uint8_t *)(p), (i), wide)
#define PUT(x, y, v) orc_st(src, (ptrdiff_t)(x) + stride * (ptrdiff_t)(y), (v), wide)
#define GET(x, y) orc_ld(src, (ptrdiff_t)(x) + stride * (ptrdiff_t)(y), wide)

/* ------------------------------------------------------------------ mode helpers (vvc_intra.c) */

ORC_API int orc_intra_pred_angle(int mode)          /* :661 */
{
    static const int angles[31] = { 0, 1, 2, 3, 4, 6, 8, 10, 12, 14, 16, 18, 20, 23, 26, 29,
                                    32, 35, 39, 45, 51, 57, 64, 73, 86, 102, 128, 171, 256, 341, 512 };
    int idx = mode > 34 ? mode - 50 : mode > 0 ? 18 - mode : 16 - mode;
    return idx < 0 ? -angles[-idx] : angles[idx];
}

ORC_API int orc_intra_inv_angle(int angle)          /* :683 — round(16384 / angle), halves away from zero */
{
    const int a = angle < 0 ? -angle : angle;
    const int r = (16384 + a / 2) / a;
    return angle < 0 ? -r : r;
}

ORC_API int orc_intra_nscale(int w, int h, int mode)   /* :538 */
{
    if (mode == 0 || mode == 1 || mode == 18 || mode == 50)
        return (orc_log2(w) + orc_log2(h) - 2) >> 2;
    const int inv = orc_intra_inv_angle(orc_intra_pred_angle(mode));
    const int side = mode >= 50 ? h : w;
    return orc_min(2, orc_log2(side) - orc_log2(3 * inv - 2) + 8);
}

ORC_API int orc_intra_need_pdpc(int w, int h, int bdpcm_flag, int mode, int ref_idx)   /* :557 */
{
    if (w >= 4 && h >= 4 && !ref_idx && !bdpcm_flag) {
        if (mode == 0 || mode == 1 || mode == 18 || mode == 50)
            return 1;
        if (mode > 18 && mode < 50)
            return 0;
        return orc_intra_nscale(w, h, mode) >= 0;
    }
    return 0;
}

static int ref_filter_mode(int mode)                /* :655 */
{
    static const int modes[12] = { -14, -12, -10, -6, 0, 2, 34, 66, 72, 76, 78, 80 };
    for (int i = 0; i < 12; i++)
        if (modes[i] == mode)
            return 1;
    return 0;
}

/* ------------------------------------------------------------------ leaf predictors */

ORC_INLINE void planar_body(const int bd, uint8_t *src, const uint8_t *top, const uint8_t *left, int w, int h, ptrdiff_t stride)
{
    const int wide = bd > 8, lw = orc_log2(w), lh = orc_log2(h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int pv = ((h - 1 - y) * PXL(top, x) + (y + 1) * PXL(left, h)) << lw;
            const int ph = ((w - 1 - x) * PXL(left, y) + (x + 1) * PXL(top, w)) << lh;
            PUT(x, y, (pv + ph + w * h) >> (lw + lh + 1));
        }
}

ORC_API void orc_pred_planar(int bd, uint8_t *src, const uint8_t *top, const uint8_t *left, int w, int h, ptrdiff_t stride)
{
    ORC_BD_SWITCH(bd, planar_body(8, src, top, left, w, h, stride), planar_body(10, src, top, left, w, h, stride),
                  planar_body(12, src, top, left, w, h, stride));
}

/* :826 — mean of the longer side, or of both when square; the store covers whole groups of 4 samples (:856) */
ORC_API void orc_pred_dc(int bd, uint8_t *src, const uint8_t *top, const uint8_t *left, int w, int h, ptrdiff_t stride)
{
    const int wide = bd > 8;
    unsigned offset = w == h ? (unsigned)w << 1 : (unsigned)orc_max(w, h);
    const int shift = orc_log2(offset);
    int sum = 0;
    if (w >= h) for (int i = 0; i < w; i++) sum += PXL(top, i);
    if (w <= h) for (int i = 0; i < h; i++) sum += PXL(left, i);
    const int dc = (sum + (int)(offset >> 1)) >> shift;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < ((w + 3) & ~3); x++)
            PUT(x, y, dc);
}

ORC_API void orc_pred_v(int bd, uint8_t *src, const uint8_t *top, int w, int h, ptrdiff_t stride)
{
    const int wide = bd > 8;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            PUT(x, y, PXL(top, x));
}

ORC_API void orc_pred_h(int bd, uint8_t *src, const uint8_t *left, int w, int h, ptrdiff_t stride)
{
    const int wide = bd > 8;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < ((w + 3) & ~3); x++)       /* stores whole groups of 4 (:885) */
            PUT(x, y, PXL(left, y));
}

/* reference sample interpolation shared by both angular directions: p points at ref[idx] */
ORC_INLINE int angular_sample(const int bd, const uint8_t *ref, ptrdiff_t i, int fact, int c_idx, int filter_flag)
{
    const int wide = bd > 8;
    if (!fact && (c_idx || !filter_flag))
        return PXL(ref, i + 1);
    if (!c_idx) {
        const int8_t *f = orc_tab_intra_luma_filter + (filter_flag * 32 + fact) * 4;
        return orc_clip_px((PXL(ref, i) * f[0] + PXL(ref, i + 1) * f[1] + PXL(ref, i + 2) * f[2] + PXL(ref, i + 3) * f[3] + 32) >> 6, bd);
    }
    return ((32 - fact) * PXL(ref, i + 1) + fact * PXL(ref, i + 2) + 16) >> 5;
}

ORC_INLINE void angular_body(const int bd, const int vertical, uint8_t *src, const uint8_t *top, const uint8_t *left,
    int w, int h, ptrdiff_t stride, int c_idx, int mode, int ref_idx, int filter_flag, int need_pdpc)
{
    const int wide = bd > 8;
    const int angle = orc_intra_pred_angle(mode);
    int inv = 0, nscale = 0;
    if (need_pdpc) {
        inv = orc_intra_inv_angle(angle);
        nscale = orc_intra_nscale(w, h, mode);
    }
    /* main reference (top for vertical modes, left for horizontal ones) shifted by 1 + ref_idx samples (:899,:955) */
    const uint8_t *mainref = vertical ? top : left, *side = vertical ? left : top;
    const ptrdiff_t base = -(1 + ref_idx);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int along = vertical ? x : y, across = vertical ? y : x;      /* `across` walks away from the main reference */
            const int pos = (1 + ref_idx + across) * angle;
            const int idx = (pos >> 5) + ref_idx, fact = pos & 31;
            int pred = angular_sample(bd, mainref, base + along + idx, fact, c_idx, filter_flag);
            if (need_pdpc) {
                if (vertical) {
                    if (x < orc_min(w, 3 << nscale)) {
                        const int l = PXL(side, y + ((256 + (x + 1) * inv) >> 9));
                        const int wl = 32 >> ((x << 1) >> nscale);
                        pred = orc_clip_px(pred + (((l - pred) * wl + 32) >> 6), bd);
                    }
                } else if (y < (3 << nscale)) {
                    const int t = PXL(side, x + ((256 + (y + 1) * inv) >> 9));
                    const int wt = 32 >> orc_min(31, (y * 2) >> nscale);
                    pred = orc_clip_px(pred + (((t - pred) * wt + 32) >> 6), bd);
                }
            }
            PUT(x, y, pred);
        }
}

ORC_API void orc_pred_angular_v(int bd, uint8_t *src, const uint8_t *top, const uint8_t *left, int w, int h, ptrdiff_t stride,
    int c_idx, int mode, int ref_idx, int filter_flag, int need_pdpc)
{
    ORC_BD_SWITCH(bd, angular_body(8, 1, src, top, left, w, h, stride, c_idx, mode, ref_idx, filter_flag, need_pdpc),
                  angular_body(10, 1, src, top, left, w, h, stride, c_idx, mode, ref_idx, filter_flag, need_pdpc),
                  angular_body(12, 1, src, top, left, w, h, stride, c_idx, mode, ref_idx, filter_flag, need_pdpc));
}

ORC_API void orc_pred_angular_h(int bd, uint8_t *src, const uint8_t *top, const uint8_t *left, int w, int h, ptrdiff_t stride,
    int c_idx, int mode, int ref_idx, int filter_flag, int need_pdpc)
{
    ORC_BD_SWITCH(bd, angular_body(8, 0, src, top, left, w, h, stride, c_idx, mode, ref_idx, filter_flag, need_pdpc),
                  angular_body(10, 0, src, top, left, w, h, stride, c_idx, mode, ref_idx, filter_flag, need_pdpc),
                  angular_body(12, 0, src, top, left, w, h, stride, c_idx, mode, ref_idx, filter_flag, need_pdpc));
}

/* ------------------------------------------------------------------ MIP (:708-824) */

static void mip_reduce(int *out, int count, const uint8_t *ref, int len, int wide)
{
    const int per = len / count;
    if (per == 1) {
        for (int i = 0; i < len; i++)
            out[i] = PXL(ref, i);
        return;
    }
    const int lg = orc_log2(per);
    for (int i = 0; i < count; i++) {
        int s = 0;
        for (int j = 0; j < per; j++)
            s += PXL(ref, i * per + j);
        out[i] = (s + (1 << (lg - 1))) >> lg;
    }
}

ORC_API void orc_pred_mip(int bd, uint8_t *src, const uint8_t *top, const uint8_t *left, int w, int h, ptrdiff_t stride,
    int mode_id, int is_transposed)
{
    const int wide = bd > 8;
    const int size_id = (w == 4 && h == 4) ? 0 : ((w == 4 || h == 4) || (w == 8 && h == 8)) ? 1 : 2;
    const int bsize = size_id == 0 ? 2 : 4, psize = size_id == 2 ? 8 : 4;
    const int in_size = 2 * bsize - (size_id == 2);
    const uint8_t *matrix = size_id == 0 ? orc_tab_mip_matrix_4x4 + mode_id * 16 * 4
                          : size_id == 1 ? orc_tab_mip_matrix_8x8 + mode_id * 16 * 8
                                         : orc_tab_mip_matrix_16x16 + mode_id * 64 * 7;
    const int up_h = w / psize, up_v = h / psize;
    int red[16];
    mip_reduce(is_transposed ? red + bsize : red, bsize, top, w, wide);
    mip_reduce(is_transposed ? red : red + bsize, bsize, left, h, wide);

    const int t0 = red[0];
    int ow, off = 1;
    if (size_id != 2) {
        off = 0;
        ow = (1 << (bd - 1)) - t0;
    } else {
        ow = red[1] - t0;
    }
    red[0] = ow;
    for (int i = 1; i < in_size; i++) {
        red[i] = red[i + off] - t0;
        ow += red[i];
    }
    ow = 32 - 32 * ow;

    /* reduced prediction lands on the bottom-right sample of every up_h x up_v cell (:727-747) */
    for (int y = 0; y < psize; y++)
        for (int x = 0; x < psize; x++) {
            int p = 0;
            for (int i = 0; i < in_size; i++)
                p += red[i] * matrix[(y * psize + x) * in_size + i];
            p = orc_clip3(((p + ow) >> 6) + t0, 0, (1 << bd) - 1);
            const int cx = is_transposed ? y : x, cy = is_transposed ? x : y;
            PUT(up_h - 1 + cx * up_h, up_v - 1 + cy * up_v, p);
        }
    /* horizontal interpolation on the rows that hold reduced samples, then vertical on every column (:749-771,:817-822) */
    if (up_h > 1)
        for (int i = 0; i < psize; i++) {
            const int row = up_v - 1 + i * up_v;
            int before = PXL(left, row);
            for (int j = 0; j < psize; j++) {
                const int after = GET((j + 1) * up_h - 1, row);
                for (int k = 1; k < up_h; k++)
                    PUT(j * up_h + k - 1, row, ((up_h - k) * before + k * after + up_h / 2) / up_h);
                before = after;
            }
        }
    if (up_v > 1)
        for (int x = 0; x < w; x++) {
            int before = PXL(top, x);
            for (int j = 0; j < psize; j++) {
                const int after = GET(x, (j + 1) * up_v - 1);
                for (int k = 1; k < up_v; k++)
                    PUT(x, j * up_v + k - 1, ((up_v - k) * before + k * after + up_v / 2) / up_v);
                before = after;
            }
        }
}

/* ------------------------------------------------------------------ intra_pred with the decoder context flattened */

#define EDGE_ORG (64 + 3)               /* MAX_TB_SIZE + 3 (:482-485) */
#define EDGE_LEN (6 * 64 + 5)

/*
 * vvc_intra_template.c:467-592 + :595-683.  What the reference pulls out of VVCLocalContext arrives in `j`:
 * the mode after wide-angle mapping, the CU flags, and the neighbour availability that ff_vvc_get_left/top_available
 * (vvc_intra.c:591-648) would grant for an unbounded request (both are min(request, limit) shaped).
 */
ORC_API void orc_intra_pred_flat(int bd, const orc_intra_job *j)
{
    const int wide = bd > 8;
    uint8_t *plane = (uint8_t *)(uintptr_t)j->plane;
    const ptrdiff_t stride = j->stride >> wide;
    const int w = j->w, h = j->h, c_idx = j->c_idx, mode = j->mode, ref_idx = j->ref_idx;
    const int is_mip = j->is_mip, no_isp = !j->isp_split;
    uint8_t *src = plane + (((ptrdiff_t)j->y * stride + j->x) << wide);
    const int need_pdpc = orc_intra_need_pdpc(w, h, j->bdpcm_flag, mode, ref_idx);
    uint16_t arr[4][EDGE_LEN];
    memset(arr, 0, sizeof(arr));
    /* edge arrays are handled as uint16 here and converted for the 8-bit leaf calls below */
    uint16_t *left = arr[0] + EDGE_ORG, *top = arr[1] + EDGE_ORG, *fleft = arr[2] + EDGE_ORG, *ftop = arr[3] + EDGE_ORG;

    const int ref_filter_flag = is_mip ? 0 : ref_filter_mode(mode);
    const int smooth = !ref_idx && w * h > 32 && !c_idx && no_isp && ref_filter_flag;
    const int ref_line = ref_idx == 3 ? -4 : -1 - ref_idx;
    int left_size, top_size, uleft, utop, refw = 0, refh = 0, angle = 0, inv = 0;

    if (is_mip || mode == 0) {
        left_size = h + 1; top_size = w + 1;
        uleft = left_size + smooth; utop = top_size + smooth;
    } else if (mode == 1) {
        uleft = left_size = h; utop = top_size = w;
    } else if (mode == 50) {
        uleft = left_size = need_pdpc ? h : 1; utop = top_size = w;
    } else if (mode == 18) {
        uleft = left_size = h; utop = top_size = need_pdpc ? w : 1;
    } else {
        if (no_isp || c_idx) { refw = w * 2; refh = h * 2; }
        else { refw = j->cb_width + w; refh = j->cb_height + h; }
        angle = orc_intra_pred_angle(mode);
        inv = orc_intra_inv_angle(angle);
        utop = top_size = refw; uleft = left_size = refh;
    }

    const int la = orc_min(uleft, j->left_avail), ta = orc_min(utop, j->top_avail);
    for (int i = 0; i < la; i++) left[i] = GET(ref_line, i);
    for (int i = 0; i < ta; i++) top[i] = GET(i, ref_line);
    for (int i = -1; i >= ref_line; i--) {
        if (j->cand_up_left) { left[i] = GET(ref_line, i); top[i] = GET(i, ref_line); }
        else if (la) left[i] = top[i] = left[0];
        else if (ta) left[i] = top[i] = top[0];
        else left[i] = top[i] = 1 << (bd - 1);
    }
    /* EXTEND reads element [avail - 1], which is element [-1] when nothing is available (:527-528) */
    for (int i = ta; i < utop; i++) top[i] = top[ta - 1];
    for (int i = la; i < uleft; i++) left[i] = left[la - 1];

    if (ref_filter_flag && smooth) {                /* ref_filter :450 */
        const int keep_last = left_size == uleft;
        fleft[-1] = ftop[-1] = (left[0] + 2 * left[-1] + top[0] + 2) >> 2;
        for (int i = 0; i < uleft - keep_last; i++) fleft[i] = (left[i - 1] + 2 * left[i] + left[i + 1] + 2) >> 2;
        for (int i = 0; i < utop - keep_last; i++) ftop[i] = (top[i - 1] + 2 * top[i] + top[i + 1] + 2) >> 2;
        if (keep_last) { ftop[utop - 1] = top[utop - 1]; fleft[uleft - 1] = left[uleft - 1]; }
        left = fleft; top = ftop;
    }
    int filter_flag = 0;
    if (!is_mip && mode != 0 && mode != 1) {
        if (!(ref_filter_flag || ref_idx || !no_isp)) {
            static const int thres[5] = { 24, 14, 2, 0, 0 };
            const int dist = orc_min(orc_abs(mode - 50), orc_abs(mode - 18));
            filter_flag = dist > thres[orc_max(0, ((orc_log2(w) + orc_log2(h)) >> 1) - 2)];   /* index < 0 only for 2xN chroma, where the flag is unused */
        }
        if (mode != 50 && mode != 18) {
            if (mode >= 34) {
                if (angle < 0) {
                    uint16_t *p = top - (ref_idx + 1);
                    for (int x = -h; x < 0; x++)
                        p[x] = left[-1 - ref_idx + orc_min((x * inv + 256) >> 9, h)];
                } else {
                    for (int i = refw; i <= refw + orc_max(1, w / h) * ref_idx + 1; i++) top[i] = top[refw - 1];
                }
            } else {
                if (angle < 0) {
                    uint16_t *p = left - (ref_idx + 1);
                    for (int x = -w; x < 0; x++)
                        p[x] = top[-1 - ref_idx + orc_min((x * inv + 256) >> 9, w)];
                } else {
                    for (int i = refh; i <= refh + orc_max(1, h / w) * ref_idx + 1; i++) left[i] = left[refh - 1];
                }
            }
        }
    }

    /* leaf predictors take pixel-typed edge arrays: narrow for 8-bit */
    uint8_t top8[EDGE_LEN], left8[EDGE_LEN];
    const uint8_t *tp = (const uint8_t *)top, *lp = (const uint8_t *)left;
    if (!wide) {
        for (int i = -EDGE_ORG; i < EDGE_LEN - EDGE_ORG; i++) { top8[EDGE_ORG + i] = (uint8_t)top[i]; left8[EDGE_ORG + i] = (uint8_t)left[i]; }
        tp = top8 + EDGE_ORG; lp = left8 + EDGE_ORG;
    }
    if (is_mip) orc_pred_mip(bd, src, tp, lp, w, h, stride, j->mip_mode, j->mip_transposed);
    else if (mode == 0) orc_pred_planar(bd, src, tp, lp, w, h, stride);
    else if (mode == 1) orc_pred_dc(bd, src, tp, lp, w, h, stride);
    else if (mode == 50) orc_pred_v(bd, src, tp, w, h, stride);
    else if (mode == 18) orc_pred_h(bd, src, lp, w, h, stride);
    else if (mode >= 34) orc_pred_angular_v(bd, src, tp, lp, w, h, stride, c_idx, mode, ref_idx, filter_flag, need_pdpc);
    else orc_pred_angular_h(bd, src, tp, lp, w, h, stride, c_idx, mode, ref_idx, filter_flag, need_pdpc);

    if (need_pdpc && !is_mip && (mode == 0 || mode == 1 || mode == 50 || mode == 18)) {       /* :654-682 */
        const int scale = (orc_log2(w) + orc_log2(h) - 2) >> 2;
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                const int val = GET(x, y);
                int l, t, wl, wt;
                if (mode == 0 || mode == 1) {
                    l = left[y]; t = top[x];
                    wl = 32 >> orc_min((x << 1) >> scale, 31);
                    wt = 32 >> orc_min((y << 1) >> scale, 31);
                } else {
                    l = left[y] - left[-1] + val; t = top[x] - top[-1] + val;
                    wl = mode == 50 ? 32 >> orc_min((x << 1) >> scale, 31) : 0;
                    wt = mode == 18 ? 32 >> orc_min((y << 1) >> scale, 31) : 0;
                }
                PUT(x, y, orc_clip_px(val + ((wl * (l - val) + wt * (t - val) + 32) >> 6), bd));
            }
    }
}

/* ------------------------------------------------------------------ CCLM with the decoder context flattened */

/* down-sampled luma sample co-located with chroma (cx, cy) of the block (vvc_intra_template.c:278-334) */
static int cclm_ds_luma(const orc_cclm_job *j, int wide, int cx, int cy)
{
    const uint8_t *luma = (const uint8_t *)(uintptr_t)j->luma;
    const ptrdiff_t s = j->luma_stride >> wide;
    const int hs = j->hs, vs = j->vs;
    const ptrdiff_t o = (ptrdiff_t)(j->y0 + (cy << vs)) * s + j->x0 + (cx << hs);
#define L(dx, dy) orc_ld(luma, o + (dx) + (dy) * s, wide)
    if (!hs && !vs)
        return L(0, 0);
    /* left neighbour: one luma sample to the left, except in column 0 without a left neighbour (:286,:300,:308,:319) */
    const int lx = (cx || j->avail_l) ? -1 : 0;
    if (!vs)
        return (L(lx, 0) + 2 * L(0, 0) + L(1, 0) + 2) >> 2;
    if (j->collocated) {
        /* sample above: the row above, except in row 0 without a top neighbour (:287,:333) */
        const int ty = (cy || j->avail_t) ? -1 : 0;
        return (L(lx, 0) + L(0, ty) + 4 * L(0, 0) + L(1, 0) + L(0, 1) + 4) >> 3;
    }
    return (L(lx, 0) + L(lx, 1) + 2 * L(0, 0) + 2 * L(0, 1) + L(1, 0) + L(1, 1) + 4) >> 3;
#undef L
}

/* vvc_intra_template.c:352 (+ :29-277).  Returns nothing; writes the Cb and Cr predictions of the block. */
ORC_API void orc_intra_cclm_pred_flat(int bd, const orc_cclm_job *j)
{
    const int wide = bd > 8;
    const int hs = j->hs, vs = j->vs;
    const int x = j->x0 >> hs, y = j->y0 >> vs, w = j->width >> hs, h = j->height >> vs;
    const int avail_t = j->avail_t, avail_l = j->avail_l;
    uint8_t *cpl[2] = { (uint8_t *)(uintptr_t)j->cb, (uint8_t *)(uintptr_t)j->cr };
    const ptrdiff_t cs[2] = { j->cb_stride >> wide, j->cr_stride >> wide };
    const uint8_t *luma = (const uint8_t *)(uintptr_t)j->luma;
    const ptrdiff_t ls = j->luma_stride >> wide;
    int a[2] = { 0, 0 }, b[2], k[2] = { 0, 0 };
    b[0] = b[1] = 1 << (bd - 1);

    if (!avail_t && !avail_l) {                                   /* cclm_pred_default :335 */
        for (int c = 0; c < 2; c++)
            for (int yy = 0; yy < h; yy++)
                for (int xx = 0; xx < w; xx++)
                    orc_st(cpl[c], (ptrdiff_t)(y + yy) * cs[c] + x + xx, 1 << (bd - 1), wide);
        return;
    }

    /* --- parameter derivation: cclm_get_select_pos :58, cclm_select_* :86-211, min/max :213, a/b/k :238 */
    int cnt[2] = { 0, 0 }, pos[2][4], have = 0;
    {
        const int lt = j->mode == 81;
        const int is4 = !avail_t || !avail_l || !lt;
        int num[2];
        if (lt) { num[0] = avail_t ? w : 0; num[1] = avail_l ? h : 0; }
        else {
            num[0] = (avail_t && j->mode == 83) ? orc_min(w + orc_min(w, h), j->top_avail_c) : 0;
            num[1] = (avail_l && j->mode == 82) ? orc_min(h + orc_min(w, h), j->left_avail_c) : 0;
        }
        if (num[0] || num[1]) {
            have = 1;
            for (int i = 0; i < 2; i++) {
                const int start = num[i] >> (2 + is4), step = orc_max(1, num[i] >> (1 + is4));
                cnt[i] = orc_min(num[i], (1 + is4) << 1);
                for (int c = 0; c < cnt[i]; c++)
                    pos[i][c] = start + c * step;
            }
        }
    }
    if (have) {
        int sel[3][8] = { { 0 } };
        const ptrdiff_t lo = (ptrdiff_t)j->y0 * ls + j->x0;
#define LP(off) orc_ld(luma, (off), wide)
        for (int i = 0; i < cnt[0]; i++) {                        /* top luma */
            if (!hs && !vs) { sel[0][i] = LP(lo - avail_t * ls + pos[0][i]); continue; }
            const int xx = pos[0][i] << hs;
            const int has_left = xx || avail_l;
            if (vs && !j->ctu_boundary) {
                const ptrdiff_t o = lo - 2 * ls + xx;
                const int l = has_left ? LP(o - 1) : LP(o);
                if (j->collocated)
                    sel[0][i] = (LP(o - ls) + l + 4 * LP(o) + LP(o + 1) + LP(o + ls) + 4) >> 3;
                else {
                    const int l1 = has_left ? LP(o - 1 + ls) : LP(o + ls);
                    sel[0][i] = (l + l1 + 2 * (LP(o) + LP(o + ls)) + LP(o + 1) + LP(o + 1 + ls) + 4) >> 3;
                }
            } else {
                const ptrdiff_t o = lo - ls + xx;
                const int l = has_left ? LP(o - 1) : LP(o);
                sel[0][i] = (l + 2 * LP(o) + LP(o + 1) + 2) >> 2;
            }
        }
        for (int i = 0; i < cnt[1]; i++) {                        /* left luma */
            if (!hs && !vs) { sel[0][cnt[0] + i] = LP(lo - avail_l + (ptrdiff_t)pos[1][i] * ls); continue; }
            const int yy = pos[1][i] << vs;
            const ptrdiff_t o = lo - (1 + hs) * avail_l + (ptrdiff_t)yy * ls, l = o - avail_l;
            int p;
            if (!vs)
                p = (LP(l) + 2 * LP(o) + LP(o + 1) + 2) >> 2;
            else if (j->collocated) {
                const int t = (yy || avail_t) ? LP(o - ls) : LP(o);
                p = (LP(l) + t + 4 * LP(o) + LP(o + 1) + LP(o + ls) + 4) >> 3;
            } else
                p = (LP(l) + LP(l + ls) + 2 * LP(o) + 2 * LP(o + ls) + LP(o + 1) + LP(o + 1 + ls) + 4) >> 3;
            sel[0][cnt[0] + i] = p;
        }
#undef LP
        for (int c = 0; c < 2; c++) {                             /* chroma neighbours :171 */
            for (int i = 0; i < cnt[0]; i++)
                sel[c + 1][i] = orc_ld(cpl[c], (ptrdiff_t)(y - 1) * cs[c] + x + pos[0][i], wide);
            for (int i = 0; i < cnt[1]; i++)
                sel[c + 1][cnt[0] + i] = orc_ld(cpl[c], (ptrdiff_t)(y + pos[1][i]) * cs[c] + x - 1, wide);
        }
        if (cnt[0] + cnt[1] == 2)
            for (int c = 0; c < 3; c++) {
                sel[c][3] = sel[c][0]; sel[c][2] = sel[c][1]; sel[c][0] = sel[c][1]; sel[c][1] = sel[c][3];
            }
        int mn[2] = { 0, 2 }, mx[2] = { 1, 3 }, t;
#define SWAP(p, q) do { t = p; p = q; q = t; } while (0)
        if (sel[0][mn[0]] > sel[0][mn[1]]) SWAP(mn[0], mn[1]);
        if (sel[0][mx[0]] > sel[0][mx[1]]) SWAP(mx[0], mx[1]);
        if (sel[0][mn[0]] > sel[0][mx[1]]) { SWAP(mn[0], mx[0]); SWAP(mn[1], mx[1]); }
        if (sel[0][mn[1]] > sel[0][mx[0]]) SWAP(mn[1], mx[0]);
#undef SWAP
        int vmax[3], vmin[3];
        for (int c = 0; c < 3; c++) {
            vmax[c] = (sel[c][mx[0]] + sel[c][mx[1]] + 1) >> 1;
            vmin[c] = (sel[c][mn[0]] + sel[c][mn[1]] + 1) >> 1;
        }
        const int diff = vmax[0] - vmin[0];
        for (int i = 0; i < 2; i++) {
            if (!diff) { a[i] = k[i] = 0; b[i] = vmin[i + 1]; continue; }
            static const int div_sig[16] = { 0, 7, 6, 5, 5, 4, 4, 3, 3, 2, 2, 1, 1, 1, 1, 0 };
            const int diffc = vmax[i + 1] - vmin[i + 1];
            int xl = orc_log2(diff);
            const int norm = ((diff << 4) >> xl) & 15;
            xl += norm ? 1 : 0;
            const int yl = orc_abs(diffc) > 0 ? orc_log2(orc_abs(diffc)) + 1 : 0;
            const int v = div_sig[norm] | 8;
            a[i] = (diffc * v + ((1 << yl) >> 1)) >> yl;
            k[i] = orc_max(1, 3 + xl - yl);
            if (3 + xl - yl < 1)
                a[i] = orc_sign(a[i]) * 15;
            b[i] = vmin[i + 1] - ((a[i] * vmin[0]) >> k[i]);
        }
    }
    /* --- linear model on the down-sampled luma (:29); the reference keeps dsy in a pixel-typed array */
    for (int yy = 0; yy < h; yy++)
        for (int xx = 0; xx < w; xx++) {
            const int dsy = cclm_ds_luma(j, wide, xx, yy);
            for (int c = 0; c < 2; c++)
                orc_st(cpl[c], (ptrdiff_t)(y + yy) * cs[c] + x + xx, orc_clip_px(((dsy * a[c]) >> k[c]) + b[c], bd), wide);
        }
}

/* ------------------------------------------------------------------ LMCS chroma residual scaling, flattened (:390-446) */

ORC_API int orc_lmcs_chroma_scale_flat(int bd, const orc_lmcs_scale_job *j)
{
    const int wide = bd > 8;
    const uint8_t *luma = (const uint8_t *)(uintptr_t)j->luma;
    const ptrdiff_t s = j->luma_stride >> wide;
    const int size = j->size_y, x = j->x_vpdu, y = j->y_vpdu;
    int cnt = 0, sum = 0;
    if (j->avail_l) {
        const int n = orc_min(j->pic_h - y, size);
        for (int i = 0; i < n; i++) sum += orc_ld(luma, (ptrdiff_t)(y + i) * s + x - 1, wide);
        sum += orc_ld(luma, (ptrdiff_t)(y + n - 1) * s + x - 1, wide) * (size - n);
        cnt = size;
    }
    if (j->avail_t) {
        const int n = orc_min(j->pic_w - x, size);
        for (int i = 0; i < n; i++) sum += orc_ld(luma, (ptrdiff_t)(y - 1) * s + x + i, wide);
        sum += orc_ld(luma, (ptrdiff_t)(y - 1) * s + x + n - 1, wide) * (size - n);
        cnt += size;
    }
    const int avg = cnt ? (sum + (cnt >> 1)) >> orc_log2(cnt) : 1 << (bd - 1);
    int i;
    for (i = j->min_bin_idx; i <= j->max_bin_idx; i++)
        if (avg < j->pivot[i + 1])
            break;
    return j->chroma_scale_coeff[orc_min(i, 15)];
}

ORC_API void orc_lmcs_scale_chroma_flat(int bd, const orc_lmcs_scale_job *j, int *dst, const int *coeff, int width, int height)
{
    const int scale = orc_lmcs_chroma_scale_flat(bd, j);
    for (int i = 0; i < width * height; i++) {
        const int c = orc_clip_intp2(coeff[i], bd);
        dst[i] = c > 0 ? (c * scale + (1 << 10)) >> 11 : -((-c * scale + (1 << 10)) >> 11);
    }
}
