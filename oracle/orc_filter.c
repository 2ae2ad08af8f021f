/*
 * oracle/orc_filter.c — CPU restatement of the in-loop filter DSP slots (LMCS, ALF, SAO, deblock).
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see orc_common.h).
 *
 * Follows, by reading:
 *   libavcodec/vvc/vvc_filter_template.c   (lmcs :25, alf :38-408, deblock :466-804)
 *   libavcodec/h26x/h2656_sao_template.c   (band :24, edge :50, restore :81,:131)
 *   libavcodec/h26x/h2656_deblock_template.c (strong :25, weak :52, chroma weak :83)
 */
#include "vvc_oracle.h"
#include "orc_common.h"

/* ------------------------------------------------------------------ LMCS */

/* vvc_filter_template.c:25 — in-place LUT map of a luma block */
ORC_INLINE void lmcs_body(const int bd, uint8_t *dst, ptrdiff_t stride, int w, int h, const uint8_t *lut)
{
    const int wide = bd > 8;
    for (int y = 0; y < h; y++, dst += stride)
        for (int x = 0; x < w; x++)
            orc_st(dst, x, orc_ld(lut, orc_ld(dst, x, wide), wide), wide);
}

ORC_API void orc_lmcs_filter(int bd, uint8_t *dst, ptrdiff_t dst_stride, int width, int height, const uint8_t *lut)
{
    ORC_BD_SWITCH(bd, lmcs_body(8, dst, dst_stride, width, height, lut),
                      lmcs_body(10, dst, dst_stride, width, height, lut),
                      lmcs_body(12, dst, dst_stride, width, height, lut));
}

/* ------------------------------------------------------------------ ALF */

/* vvc_filter_template.c:38 — the pair sum is narrowed to int16_t by the return type */
ORC_INLINE int alf_pair(int cur, int a, int b, int lim)
{
    return (int16_t)(orc_clip3(a - cur, -lim, lim) + orc_clip3(b - cur, -lim, lim));
}

/*
 * Rows within 3 of the virtual boundary fold their vertical taps toward the centre row
 * (vvc_filter_template.c:80-96 luma, :174-190 chroma): tap row distance k becomes min(k, dist)
 * where dist is the number of rows between this row and the boundary on its own side.
 */
ORC_INLINE int alf_vb_dist(int y, int vb_pos)
{
    return y < vb_pos ? vb_pos - 1 - y : y - vb_pos;
}

/* luma diamond: tap k pairs (+dy,+dx) with (-dy,-dx); vvc_filter_template.c:102-113 */
static const int8_t alf_luma_tap[12][2] = {
    { 3, 0 }, { 2, 1 }, { 2, 0 }, { 2, -1 }, { 1, 2 }, { 1, 1 },
    { 1, 0 }, { 1, -1 }, { 1, -2 }, { 0, 3 }, { 0, 2 }, { 0, 1 },
};
/* chroma diamond; vvc_filter_template.c:196-201 */
static const int8_t alf_chroma_tap[6][2] = {
    { 2, 0 }, { 1, 1 }, { 1, 0 }, { 1, -1 }, { 0, 2 }, { 0, 1 },
};

ORC_INLINE void alf_filter_body(const int bd, const int luma, uint8_t *dst, ptrdiff_t dst_stride,
    const uint8_t *src, ptrdiff_t src_stride, int w, int h,
    const int16_t *filter, const int16_t *clip, int vb_pos)
{
    const int wide = bd > 8;
    const int ntap = luma ? 12 : 6;
    const int8_t (*tap)[2] = luma ? alf_luma_tap : alf_chroma_tap;
    const ptrdiff_t ss = src_stride >> wide;

    for (int y = 0; y < h; y++) {
        const int dist = alf_vb_dist(y, vb_pos);
        const int near_vb = dist == 0;
        for (int x = 0; x < w; x++) {
            /* luma carries one 12-tap set per 4x4 block, raster order (:131-132); chroma one set per call */
            const int blk = luma ? ((y >> 2) * (w >> 2) + (x >> 2)) * 12 : 0;
            const int16_t *f = filter + blk, *c = clip + blk;
            const ptrdiff_t o = (ptrdiff_t)y * ss + x;
            const int cur = orc_ld(src, o, wide);
            int sum = 0;
            for (int k = 0; k < ntap; k++) {
                const int dy = orc_min(tap[k][0], dist), dx = tap[k][1];
                const int a = orc_ld(src, o + dy * ss + dx, wide);
                const int b = orc_ld(src, o - dy * ss - dx, wide);
                sum += f[k] * alf_pair(cur, a, b, c[k]);
            }
            sum = near_vb ? (sum + 512) >> 10 : (sum + 64) >> 7;     /* :115-118 */
            orc_st(dst + (ptrdiff_t)y * dst_stride, x, orc_clip_px(sum + cur, bd), wide);
        }
    }
}

ORC_API void orc_alf_filter_luma(int bd, uint8_t *dst, ptrdiff_t dst_stride, const uint8_t *src, ptrdiff_t src_stride,
    int width, int height, const int16_t *filter, const int16_t *clip, int vb_pos)
{
    ORC_BD_SWITCH(bd,
        alf_filter_body(8, 1, dst, dst_stride, src, src_stride, width, height, filter, clip, vb_pos),
        alf_filter_body(10, 1, dst, dst_stride, src, src_stride, width, height, filter, clip, vb_pos),
        alf_filter_body(12, 1, dst, dst_stride, src, src_stride, width, height, filter, clip, vb_pos));
}

ORC_API void orc_alf_filter_chroma(int bd, uint8_t *dst, ptrdiff_t dst_stride, const uint8_t *src, ptrdiff_t src_stride,
    int width, int height, const int16_t *filter, const int16_t *clip, int vb_pos)
{
    ORC_BD_SWITCH(bd,
        alf_filter_body(8, 0, dst, dst_stride, src, src_stride, width, height, filter, clip, vb_pos),
        alf_filter_body(10, 0, dst, dst_stride, src, src_stride, width, height, filter, clip, vb_pos),
        alf_filter_body(12, 0, dst, dst_stride, src, src_stride, width, height, filter, clip, vb_pos));
}

/* vvc_filter_template.c:223 — cross-component ALF: 7 luma taps correct one chroma sample */
ORC_INLINE void alf_cc_body(const int bd, uint8_t *dst, ptrdiff_t dst_stride, const uint8_t *luma, ptrdiff_t luma_stride,
    int w, int h, int hs, int vs, const int16_t *f, int vb_pos)
{
    const int wide = bd > 8;
    const ptrdiff_t ls = luma_stride >> wide;
    for (int y = 0; y < h; y++) {
        const int ly = y << vs;
        if (!vs && (ly == vb_pos || ly == vb_pos + 1))            /* :242 */
            continue;
        /* row offsets of the above / below / second-below taps, folded at the boundary (:245-248) */
        int up = -1, dn = 1, dn2 = 2;
        if (ly == vb_pos - 2 || ly == vb_pos + 1)
            dn2 = 1;
        else if (ly == vb_pos - 1 || ly == vb_pos)
            up = dn = dn2 = 0;
        for (int x = 0; x < w; x++) {
            const ptrdiff_t o = (ptrdiff_t)ly * ls + (x << hs);
            const int c = orc_ld(luma, o, wide);
            int sum = 0;
            sum += f[0] * (orc_ld(luma, o + up * ls, wide) - c);
            sum += f[1] * (orc_ld(luma, o - 1, wide) - c);
            sum += f[2] * (orc_ld(luma, o + 1, wide) - c);
            sum += f[3] * (orc_ld(luma, o + dn * ls - 1, wide) - c);
            sum += f[4] * (orc_ld(luma, o + dn * ls, wide) - c);
            sum += f[5] * (orc_ld(luma, o + dn * ls + 1, wide) - c);
            sum += f[6] * (orc_ld(luma, o + dn2 * ls, wide) - c);
            sum = orc_clip3((sum + 64) >> 7, -(1 << (bd - 1)), (1 << (bd - 1)) - 1);
            uint8_t *d = dst + (ptrdiff_t)y * dst_stride;
            orc_st(d, x, orc_clip_px(sum + orc_ld(d, x, wide), bd), wide);
        }
    }
}

ORC_API void orc_alf_filter_cc(int bd, uint8_t *dst, ptrdiff_t dst_stride, const uint8_t *luma, ptrdiff_t luma_stride,
    int width, int height, int hs, int vs, const int16_t *filter, int vb_pos)
{
    ORC_BD_SWITCH(bd,
        alf_cc_body(8, dst, dst_stride, luma, luma_stride, width, height, hs, vs, filter, vb_pos),
        alf_cc_body(10, dst, dst_stride, luma, luma_stride, width, height, hs, vs, filter, vb_pos),
        alf_cc_body(12, dst, dst_stride, luma, luma_stride, width, height, hs, vs, filter, vb_pos));
}

/* vvc_filter_template.c:270 — direction sums {V,H,D0,D1} of one 4x4 -> class 0..24, transpose 0..3 */
ORC_INLINE void alf_block_class(const int bd, const int *sum, int ac, int *class_idx, int *transpose_idx)
{
    static const uint8_t var_tab[16] = { 0, 1, 2, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 3, 3, 4 };
    const int v = sum[0], h = sum[1], d0 = sum[2], d1 = sum[3];
    const int dir_hv = v <= h, dir_d = d0 <= d1;
    const int hv_hi = orc_max(v, h), hv_lo = orc_min(v, h);
    const int d_hi = orc_max(d0, d1), d_lo = orc_min(d0, d1);
    /* 64-bit unsigned cross-multiply, :285 */
    const int main_is_hv = (uint64_t)(uint32_t)d_hi * (uint32_t)hv_lo <= (uint64_t)(uint32_t)hv_hi * (uint32_t)d_lo;
    const int hi = main_is_hv ? hv_hi : d_hi;
    const int lo = main_is_hv ? hv_lo : d_lo;
    int cls = var_tab[orc_clip_uintp2(((h + v) * ac) >> (bd - 1), 4)];
    if (hi * 2 > 9 * lo)
        cls += ((main_is_hv << 1) + 2) * 5;
    else if (hi > 2 * lo)
        cls += ((main_is_hv << 1) + 1) * 5;
    *class_idx = cls;
    *transpose_idx = dir_d * 2 + dir_hv;
}

/* vvc_filter_template.c:299 — gradient_tmp layout: ((h+4)/2) rows x ((w+4)/2) cells x 4 ints */
ORC_INLINE void alf_classify_body(const int bd, int *class_idx, int *transpose_idx, const uint8_t *src, ptrdiff_t src_stride,
    int w, int h, int vb_pos, int *grad)
{
    const int wide = bd > 8;
    const ptrdiff_t ss = src_stride >> wide;
    const int gw = (w + 4) >> 1, gh = (h + 4) >> 1;

    for (int gy = 0; gy < gh; gy++) {
        const int y = 2 * gy;                 /* row y of the (h+4)-row window == picture row y-2 */
        /* rows r0..r3 = picture rows y-3 .. y, with the virtual-boundary substitutions of :321-324 */
        int r0 = y - 3, r1 = y - 2, r2 = y - 1, r3 = y;
        if (y == vb_pos)
            r3 = r2;
        else if (y == vb_pos + 2)
            r0 = r1;
        for (int gx = 0; gx < gw; gx++) {
            const int xa = 2 * gx - 2, xb = xa + 1;      /* sample A at (r1, xa), sample B at (r2, xb) */
#define PX(r, c) orc_ld(src, (ptrdiff_t)(r) * ss + (c), wide)
            const int a2 = PX(r1, xa) << 1, b2 = PX(r2, xb) << 1;
            int *g = grad + ((ptrdiff_t)gy * gw + gx) * 4;
            g[0] = orc_abs(a2 - PX(r0, xa) - PX(r2, xa)) + orc_abs(b2 - PX(r1, xb) - PX(r3, xb));
            g[1] = orc_abs(a2 - PX(r1, xa - 1) - PX(r1, xa + 1)) + orc_abs(b2 - PX(r2, xb - 1) - PX(r2, xb + 1));
            g[2] = orc_abs(a2 - PX(r0, xa - 1) - PX(r2, xa + 1)) + orc_abs(b2 - PX(r1, xb - 1) - PX(r3, xb + 1));
            g[3] = orc_abs(a2 - PX(r0, xa + 1) - PX(r2, xa - 1)) + orc_abs(b2 - PX(r1, xb + 1) - PX(r3, xb - 1));
#undef PX
        }
    }

    for (int y = 0; y < h; y += 4) {
        /* 4 gradient rows per block, trimmed to 3 next to the virtual boundary (:346-356) */
        int first = 0, last = 4, ac = 2;
        if (y + 4 == vb_pos) {
            last = 3;
            ac = 3;
        } else if (y == vb_pos) {
            first = 1;
            ac = 3;
        }
        for (int x = 0; x < w; x += 4) {
            int sum[4] = { 0, 0, 0, 0 };
            for (int i = first; i < last; i++)
                for (int j = 0; j < 4; j++) {
                    const int *g = grad + ((ptrdiff_t)((y >> 1) + i) * gw + (x >> 1) + j) * 4;
                    sum[0] += g[0]; sum[1] += g[1]; sum[2] += g[2]; sum[3] += g[3];
                }
            alf_block_class(bd, sum, ac, class_idx++, transpose_idx++);
        }
    }
}

ORC_API void orc_alf_classify(int bd, int *class_idx, int *transpose_idx, const uint8_t *src, ptrdiff_t src_stride,
    int width, int height, int vb_pos, int *gradient_tmp)
{
    ORC_BD_SWITCH(bd,
        alf_classify_body(8, class_idx, transpose_idx, src, src_stride, width, height, vb_pos, gradient_tmp),
        alf_classify_body(10, class_idx, transpose_idx, src, src_stride, width, height, vb_pos, gradient_tmp),
        alf_classify_body(12, class_idx, transpose_idx, src, src_stride, width, height, vb_pos, gradient_tmp));
}

/* vvc_filter_template.c:383 — gather 12 coefficients + clip values per 4x4 under the transpose permutation */
const uint8_t orc_alf_transpose_perm[4][12] = {
    { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11 },
    { 9, 4, 10, 8, 1, 5, 11, 7, 3, 0, 2, 6 },
    { 0, 3, 2, 1, 8, 7, 6, 5, 4, 9, 10, 11 },
    { 9, 8, 10, 4, 3, 7, 11, 5, 1, 0, 2, 6 },
};

ORC_API void orc_alf_recon_coeff_and_clip(int bd, int16_t *coeff, int16_t *clip, const int *class_idx, const int *transpose_idx,
    int size, const int16_t *coeff_set, const uint8_t *clip_idx_set, const uint8_t *class_to_filt)
{
    const int16_t clip_val[4] = { (int16_t)(1 << bd), (int16_t)(1 << (bd - 3)), (int16_t)(1 << (bd - 5)), (int16_t)(1 << (bd - 7)) };
    for (int i = 0; i < size; i++) {
        const int16_t *cs = coeff_set + class_to_filt[class_idx[i]] * 12;
        const uint8_t *ci = clip_idx_set + class_idx[i] * 12;
        const uint8_t *perm = orc_alf_transpose_perm[transpose_idx[i]];
        for (int j = 0; j < 12; j++) {
            coeff[i * 12 + j] = cs[perm[j]];
            clip[i * 12 + j]  = clip_val[ci[perm[j]]];
        }
    }
}

/* ------------------------------------------------------------------ SAO */

/* h2656_sao_template.c:24 */
ORC_INLINE void sao_band_body(const int bd, uint8_t *dst, const uint8_t *src, ptrdiff_t dst_stride, ptrdiff_t src_stride,
    const int16_t *offset_val, int left_class, int w, int h)
{
    const int wide = bd > 8;
    int table[32] = { 0 };
    for (int k = 0; k < 4; k++)
        table[(k + left_class) & 31] = offset_val[k + 1];
    for (int y = 0; y < h; y++, dst += dst_stride, src += src_stride)
        for (int x = 0; x < w; x++) {
            const int s = orc_ld(src, x, wide);
            orc_st(dst, x, orc_clip_px(s + table[(s >> (bd - 5)) & 31], bd), wide);
        }
}

ORC_API void orc_sao_band_filter(int bd, uint8_t *dst, const uint8_t *src, ptrdiff_t dst_stride, ptrdiff_t src_stride,
    const int16_t *sao_offset_val, int sao_left_class, int width, int height)
{
    ORC_BD_SWITCH(bd,
        sao_band_body(8, dst, src, dst_stride, src_stride, sao_offset_val, sao_left_class, width, height),
        sao_band_body(10, dst, src, dst_stride, src_stride, sao_offset_val, sao_left_class, width, height),
        sao_band_body(12, dst, src, dst_stride, src_stride, sao_offset_val, sao_left_class, width, height));
}

/* h2656_sao_template.c:50 — the source is the caller's padded CTB copy with the fixed byte stride
 * 2*MAX_PB_SIZE + AV_INPUT_BUFFER_PADDING_SIZE (vvcdsp.h:140) */
ORC_INLINE void sao_edge_body(const int bd, uint8_t *dst, const uint8_t *src, ptrdiff_t dst_stride,
    const int16_t *offset_val, int eo, int w, int h)
{
    static const uint8_t cat[5] = { 1, 2, 0, 3, 4 };
    /* neighbour a/b displacement (dx, dy) per edge-offset class */
    static const int8_t nb[4][4] = { { -1, 0, 1, 0 }, { 0, -1, 0, 1 }, { -1, -1, 1, 1 }, { 1, -1, -1, 1 } };
    const int wide = bd > 8;
    const ptrdiff_t ss = ORC_SAO_EDGE_SRC_STRIDE >> wide;
    const ptrdiff_t oa = nb[eo][0] + nb[eo][1] * ss, ob = nb[eo][2] + nb[eo][3] * ss;
    for (int y = 0; y < h; y++, dst += dst_stride)
        for (int x = 0; x < w; x++) {
            const ptrdiff_t o = (ptrdiff_t)y * ss + x;
            const int s = orc_ld(src, o, wide);
            const int k = 2 + orc_sign(s - orc_ld(src, o + oa, wide)) + orc_sign(s - orc_ld(src, o + ob, wide));
            orc_st(dst, x, orc_clip_px(s + offset_val[cat[k]], bd), wide);
        }
}

ORC_API void orc_sao_edge_filter(int bd, uint8_t *dst, const uint8_t *src, ptrdiff_t dst_stride,
    const int16_t *sao_offset_val, int eo, int width, int height)
{
    ORC_BD_SWITCH(bd,
        sao_edge_body(8, dst, src, dst_stride, sao_offset_val, eo, width, height),
        sao_edge_body(10, dst, src, dst_stride, sao_offset_val, eo, width, height),
        sao_edge_body(12, dst, src, dst_stride, sao_offset_val, eo, width, height));
}

/* h2656_sao_template.c:81 (variant 0) and :131 (variant 1).  The reference reads offset_val / eo_class from
 * SAOParams (vvc_ctu.h:440); the flattened form takes the component's offset table and class directly. */
ORC_INLINE void sao_restore_body(const int bd, const int variant, uint8_t *dst, const uint8_t *src, ptrdiff_t dst_stride, ptrdiff_t src_stride,
    const int16_t *offset_val, int eo, const int *borders, int w, int h,
    const uint8_t *vert_edge, const uint8_t *horiz_edge, const uint8_t *diag_edge)
{
    const int wide = bd > 8;
    const ptrdiff_t ds = dst_stride >> wide, ss = src_stride >> wide;
    const int off0 = offset_val[0];
    int x0 = 0, y0 = 0, x1 = w, y1 = h;
#define SRC(y, x) orc_ld(src, (ptrdiff_t)(y) * ss + (x), wide)
#define PUT(y, x, v) orc_st(dst, (ptrdiff_t)(y) * ds + (x), (v), wide)
    if (eo != 1 /* SAO_EO_VERT */) {
        if (borders[0]) {
            for (int y = 0; y < h; y++)
                PUT(y, 0, orc_clip_px(SRC(y, 0) + off0, bd));
            x0 = 1;
        }
        if (borders[2]) {
            for (int y = 0; y < h; y++)
                PUT(y, w - 1, orc_clip_px(SRC(y, w - 1) + off0, bd));
            x1--;
        }
    }
    if (eo != 0 /* SAO_EO_HORIZ */) {
        if (borders[1]) {
            for (int x = x0; x < x1; x++)
                PUT(0, x, orc_clip_px(SRC(0, x) + off0, bd));
            if (variant)
                y0 = 1;
        }
        if (borders[3]) {
            for (int x = x0; x < x1; x++)
                PUT(h - 1, x, orc_clip_px(SRC(h - 1, x) + off0, bd));
            y1--;
        }
    }
    if (variant) {
        const int keep_ul = !diag_edge[0] && eo == 2 /* 135D */ && !borders[0] && !borders[1];
        const int keep_ur = !diag_edge[1] && eo == 3 /* 45D  */ && !borders[1] && !borders[2];
        const int keep_lr = !diag_edge[2] && eo == 2 && !borders[2] && !borders[3];
        const int keep_ll = !diag_edge[3] && eo == 3 && !borders[0] && !borders[3];
        if (vert_edge[0] && eo != 1)
            for (int y = y0 + keep_ul; y < y1 - keep_ll; y++)
                PUT(y, 0, SRC(y, 0));
        if (vert_edge[1] && eo != 1)
            for (int y = y0 + keep_ur; y < y1 - keep_lr; y++)
                PUT(y, x1 - 1, SRC(y, x1 - 1));
        if (horiz_edge[0] && eo != 0)
            for (int x = x0 + keep_ul; x < x1 - keep_ur; x++)
                PUT(0, x, SRC(0, x));
        if (horiz_edge[1] && eo != 0)
            for (int x = x0 + keep_ll; x < x1 - keep_lr; x++)
                PUT(y1 - 1, x, SRC(y1 - 1, x));
        if (diag_edge[0] && eo == 2)
            PUT(0, 0, SRC(0, 0));
        if (diag_edge[1] && eo == 3)
            PUT(0, x1 - 1, SRC(0, x1 - 1));
        if (diag_edge[2] && eo == 2)
            PUT(y1 - 1, x1 - 1, SRC(y1 - 1, x1 - 1));
        if (diag_edge[3] && eo == 3)
            PUT(y1 - 1, 0, SRC(y1 - 1, 0));
    }
#undef SRC
#undef PUT
}

ORC_API void orc_sao_edge_restore(int bd, int variant, uint8_t *dst, const uint8_t *src, ptrdiff_t dst_stride, ptrdiff_t src_stride,
    const int16_t *offset_val, int eo_class, const int *borders, int width, int height,
    const uint8_t *vert_edge, const uint8_t *horiz_edge, const uint8_t *diag_edge)
{
    ORC_BD_SWITCH(bd,
        sao_restore_body(8, variant, dst, src, dst_stride, src_stride, offset_val, eo_class, borders, width, height, vert_edge, horiz_edge, diag_edge),
        sao_restore_body(10, variant, dst, src, dst_stride, src_stride, offset_val, eo_class, borders, width, height, vert_edge, horiz_edge, diag_edge),
        sao_restore_body(12, variant, dst, src, dst_stride, src_stride, offset_val, eo_class, borders, width, height, vert_edge, horiz_edge, diag_edge));
}

/* ------------------------------------------------------------------ deblock */

/*
 * One filtered line: samples p[0..7] lie at pix[-(i+1)*xs], q[0..7] at pix[i*xs]; `xs` steps across the
 * edge, `ys` along it (both in pixels).  Helper views keep the code independent of the direction.
 */
typedef struct { uint8_t *pix; ptrdiff_t xs, ys; int wide; } dbk_t;

ORC_INLINE int dbk_p(const dbk_t *d, int line, int i) { return orc_ld(d->pix, line * d->ys - (i + 1) * d->xs, d->wide); }
ORC_INLINE int dbk_q(const dbk_t *d, int line, int i) { return orc_ld(d->pix, line * d->ys + i * d->xs, d->wide); }
/* stores truncate to the pixel type without clipping, as the reference's P0 = ... assignments do */
ORC_INLINE void dbk_sp(const dbk_t *d, int line, int i, int v) { orc_st(d->pix, line * d->ys - (i + 1) * d->xs, v, d->wide); }
ORC_INLINE void dbk_sq(const dbk_t *d, int line, int i, int v) { orc_st(d->pix, line * d->ys + i * d->xs, v, d->wide); }
ORC_INLINE int dbk_d2(int a, int b, int c) { return orc_abs(a - 2 * b + c); }

/* h2656_deblock_template.c:25 */
ORC_INLINE void dbk_luma_strong(const dbk_t *d, int tc, int no_p, int no_q)
{
    const int tc2 = tc << 1, tc3 = tc * 3;
    for (int l = 0; l < 4; l++) {
        const int p3 = dbk_p(d, l, 3), p2 = dbk_p(d, l, 2), p1 = dbk_p(d, l, 1), p0 = dbk_p(d, l, 0);
        const int q0 = dbk_q(d, l, 0), q1 = dbk_q(d, l, 1), q2 = dbk_q(d, l, 2), q3 = dbk_q(d, l, 3);
        if (!no_p) {
            dbk_sp(d, l, 0, p0 + orc_clip3(((p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3) - p0, -tc3, tc3));
            dbk_sp(d, l, 1, p1 + orc_clip3(((p2 + p1 + p0 + q0 + 2) >> 2) - p1, -tc2, tc2));
            dbk_sp(d, l, 2, p2 + orc_clip3(((2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3) - p2, -tc, tc));
        }
        if (!no_q) {
            dbk_sq(d, l, 0, q0 + orc_clip3(((p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3) - q0, -tc3, tc3));
            dbk_sq(d, l, 1, q1 + orc_clip3(((p0 + q0 + q1 + q2 + 2) >> 2) - q1, -tc2, tc2));
            dbk_sq(d, l, 2, q2 + orc_clip3(((2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3) - q2, -tc, tc));
        }
    }
}

/* h2656_deblock_template.c:52 */
ORC_INLINE void dbk_luma_weak(const dbk_t *d, int bd, int tc, int no_p, int no_q, int nd_p, int nd_q)
{
    const int tc_2 = tc >> 1;
    for (int l = 0; l < 4; l++) {
        const int p2 = dbk_p(d, l, 2), p1 = dbk_p(d, l, 1), p0 = dbk_p(d, l, 0);
        const int q0 = dbk_q(d, l, 0), q1 = dbk_q(d, l, 1), q2 = dbk_q(d, l, 2);
        int delta = (9 * (q0 - p0) - 3 * (q1 - p1) + 8) >> 4;
        if (orc_abs(delta) >= 10 * tc)
            continue;
        delta = orc_clip3(delta, -tc, tc);
        if (!no_p)
            dbk_sp(d, l, 0, orc_clip_px(p0 + delta, bd));
        if (!no_q)
            dbk_sq(d, l, 0, orc_clip_px(q0 - delta, bd));
        if (!no_p && nd_p > 1)
            dbk_sp(d, l, 1, orc_clip_px(p1 + orc_clip3((((p2 + p0 + 1) >> 1) - p1 + delta) >> 1, -tc_2, tc_2), bd));
        if (!no_q && nd_q > 1)
            dbk_sq(d, l, 1, orc_clip_px(q1 + orc_clip3((((q2 + q0 + 1) >> 1) - q1 - delta) >> 1, -tc_2, tc_2), bd));
    }
}

/* vvc_filter_template.c:466 — long-tap luma filter; per-side interpolation weights toward the far reference */
static const uint8_t dbk_w3[3] = { 53, 32, 11 }, dbk_w5[5] = { 58, 45, 32, 19, 6 }, dbk_w7[7] = { 59, 50, 41, 32, 23, 14, 5 };
static const uint8_t dbk_t3[3] = { 6, 4, 2 }, dbk_t5[5] = { 6, 5, 4, 3, 2 }, dbk_t7[7] = { 6, 5, 4, 3, 2, 1, 1 };

ORC_INLINE void dbk_luma_large(const dbk_t *d, int tc, int no_p, int no_q, int len_p, int len_q)
{
    for (int l = 0; l < 4; l++) {
        int p[8], q[8], m;
        for (int i = 0; i < 8; i++) {
            p[i] = dbk_p(d, l, i);
            q[i] = dbk_q(d, l, i);
        }
        if (len_p == 5 && len_q == 5)
            m = (p[4] + p[3] + 2 * (p[2] + p[1] + p[0] + q[0] + q[1] + q[2]) + q[3] + q[4] + 8) >> 4;
        else if (len_p == len_q)
            m = (p[6] + p[5] + p[4] + p[3] + p[2] + p[1] + 2 * (p[0] + q[0]) + q[1] + q[2] + q[3] + q[4] + q[5] + q[6] + 8) >> 4;
        else if (len_p + len_q == 12)
            m = (p[5] + p[4] + p[3] + p[2] + 2 * (p[1] + p[0] + q[0] + q[1]) + q[2] + q[3] + q[4] + q[5] + 8) >> 4;
        else if (len_p + len_q == 8)
            m = (p[3] + p[2] + p[1] + p[0] + q[0] + q[1] + q[2] + q[3] + 4) >> 3;
        else if (len_q == 7)
            m = (2 * (p[2] + p[1] + p[0] + q[0]) + p[0] + p[1] + q[1] + q[2] + q[3] + q[4] + q[5] + q[6] + 8) >> 4;
        else
            m = (p[6] + p[5] + p[4] + p[3] + p[2] + p[1] + 2 * (q[2] + q[1] + q[0] + p[0]) + q[0] + q[1] + 8) >> 4;
        if (!no_p) {
            const int ref = (p[len_p] + p[len_p - 1] + 1) >> 1;
            const int n = len_p == 3 ? 3 : len_p == 5 ? 5 : 7;
            const uint8_t *w = n == 3 ? dbk_w3 : n == 5 ? dbk_w5 : dbk_w7;
            const uint8_t *t = n == 3 ? dbk_t3 : n == 5 ? dbk_t5 : dbk_t7;
            for (int i = 0; i < n; i++) {
                const int lim = (tc * t[i]) >> 1;
                dbk_sp(d, l, i, p[i] + orc_clip3(((m * w[i] + ref * (64 - w[i]) + 32) >> 6) - p[i], -lim, lim));
            }
        }
        if (!no_q) {
            const int ref = (q[len_q] + q[len_q - 1] + 1) >> 1;
            const int n = len_q == 3 ? 3 : len_q == 5 ? 5 : 7;
            const uint8_t *w = n == 3 ? dbk_w3 : n == 5 ? dbk_w5 : dbk_w7;
            const uint8_t *t = n == 3 ? dbk_t3 : n == 5 ? dbk_t5 : dbk_t7;
            for (int i = 0; i < n; i++) {
                const int lim = (tc * t[i]) >> 1;
                dbk_sq(d, l, i, q[i] + orc_clip3(((m * w[i] + ref * (64 - w[i]) + 32) >> 6) - q[i], -lim, lim));
            }
        }
    }
}

/* vvc_filter_template.c:546 — 8 lines = 2 segments of 4; decisions from lines 0 and 3 of each segment */
ORC_INLINE void dbk_luma_body(const int bd, uint8_t *pix, ptrdiff_t xs_b, ptrdiff_t ys_b,
    const int32_t *beta_in, const int32_t *tc_in, const uint8_t *no_p_in, const uint8_t *no_q_in,
    const uint8_t *max_len_p_in, const uint8_t *max_len_q_in, int hor_ctu_edge)
{
    const int wide = bd > 8;
    const ptrdiff_t xs = xs_b >> wide, ys = ys_b >> wide;
    for (int seg = 0; seg < 2; seg++) {
        const int tc = bd < 10 ? (tc_in[seg] + (1 << (9 - bd))) >> (10 - bd) : tc_in[seg] << (bd - 10);
        if (!tc)
            continue;
        dbk_t d = { pix + ((seg * 4 * ys) << wide), xs, ys, wide };
        const int no_p = no_p_in[seg], no_q = no_q_in[seg];
        int len_p = max_len_p_in[seg], len_q = max_len_q_in[seg];
#define P(i) dbk_p(&d, 0, i)
#define Q(i) dbk_q(&d, 0, i)
#define TP(i) dbk_p(&d, 3, i)
#define TQ(i) dbk_q(&d, 3, i)
        const int dp0 = dbk_d2(P(2), P(1), P(0)), dq0 = dbk_d2(Q(2), Q(1), Q(0));
        const int dp3 = dbk_d2(TP(2), TP(1), TP(0)), dq3 = dbk_d2(TQ(2), TQ(1), TQ(0));
        const int d0 = dp0 + dq0, d3 = dp3 + dq3;
        const int tc25 = (tc * 5 + 1) >> 1;
        const int large_p = len_p > 3 && !hor_ctu_edge, large_q = len_q > 3;
        const int beta = beta_in[seg] << (bd - 8);
        int done = 0;

        if (large_p || large_q) {
            const int dp0l = large_p ? (dp0 + dbk_d2(P(5), P(4), P(3)) + 1) >> 1 : dp0;
            const int dq0l = large_q ? (dq0 + dbk_d2(Q(5), Q(4), Q(3)) + 1) >> 1 : dq0;
            const int dp3l = large_p ? (dp3 + dbk_d2(TP(5), TP(4), TP(3)) + 1) >> 1 : dp3;
            const int dq3l = large_q ? (dq3 + dbk_d2(TQ(5), TQ(4), TQ(3)) + 1) >> 1 : dq3;
            const int d0l = dp0l + dq0l, d3l = dp3l + dq3l;
            const int beta53 = (beta * 3) >> 5, beta_4 = beta >> 4;
            len_p = large_p ? len_p : 3;
            len_q = large_q ? len_q : 3;
            if (d0l + d3l < beta) {
                const int sp0l = orc_abs(P(3) - P(0)) + (len_p == 7 ? orc_abs(P(7) - P(6) - P(5) + P(4)) : 0);
                const int sq0l = orc_abs(Q(0) - Q(3)) + (len_q == 7 ? orc_abs(Q(4) - Q(5) - Q(6) + Q(7)) : 0);
                const int sp3l = orc_abs(TP(3) - TP(0)) + (len_p == 7 ? orc_abs(TP(7) - TP(6) - TP(5) + TP(4)) : 0);
                const int sq3l = orc_abs(TQ(0) - TQ(3)) + (len_q == 7 ? orc_abs(TQ(4) - TQ(5) - TQ(6) + TQ(7)) : 0);
                const int sp0 = large_p ? (sp0l + orc_abs(P(3) - P(len_p)) + 1) >> 1 : sp0l;
                const int sp3 = large_p ? (sp3l + orc_abs(TP(3) - TP(len_p)) + 1) >> 1 : sp3l;
                const int sq0 = large_q ? (sq0l + orc_abs(Q(3) - Q(len_q)) + 1) >> 1 : sq0l;
                const int sq3 = large_q ? (sq3l + orc_abs(TQ(3) - TQ(len_q)) + 1) >> 1 : sq3l;
                if (sp0 + sq0 < beta53 && orc_abs(P(0) - Q(0)) < tc25 &&
                    sp3 + sq3 < beta53 && orc_abs(TP(0) - TQ(0)) < tc25 &&
                    (d0l << 1) < beta_4 && (d3l << 1) < beta_4) {
                    dbk_luma_large(&d, tc, no_p, no_q, len_p, len_q);
                    done = 1;
                }
            }
        }
        if (!done && d0 + d3 < beta) {
            const int beta_3 = beta >> 3, beta_2 = beta >> 2;
            if (len_p > 2 && len_q > 2 &&
                orc_abs(P(3) - P(0)) + orc_abs(Q(3) - Q(0)) < beta_3 && orc_abs(P(0) - Q(0)) < tc25 &&
                orc_abs(TP(3) - TP(0)) + orc_abs(TQ(3) - TQ(0)) < beta_3 && orc_abs(TP(0) - TQ(0)) < tc25 &&
                (d0 << 1) < beta_2 && (d3 << 1) < beta_2) {
                dbk_luma_strong(&d, tc, no_p, no_q);
            } else {
                int nd_p = 1, nd_q = 1;
                if (len_p > 1 && len_q > 1) {
                    const int side = (beta + (beta >> 1)) >> 3;
                    if (dp0 + dp3 < side) nd_p = 2;
                    if (dq0 + dq3 < side) nd_q = 2;
                }
                dbk_luma_weak(&d, bd, tc, no_p, no_q, nd_p, nd_q);
            }
        }
#undef P
#undef Q
#undef TP
#undef TQ
    }
}

/* dir 0 = "h" slot (xstride = stride: filters a horizontal edge), dir 1 = "v" slot; vvc_filter_template.c:772-786 */
ORC_API void orc_lf_filter_luma(int bd, int dir, uint8_t *pix, ptrdiff_t stride, const int32_t *beta, const int32_t *tc,
    const uint8_t *no_p, const uint8_t *no_q, const uint8_t *max_len_p, const uint8_t *max_len_q, int hor_ctu_edge)
{
    const ptrdiff_t px = bd > 8 ? 2 : 1;
    const ptrdiff_t xs = dir == 0 ? stride : px, ys = dir == 0 ? px : stride;
    ORC_BD_SWITCH(bd,
        dbk_luma_body(8, pix, xs, ys, beta, tc, no_p, no_q, max_len_p, max_len_q, hor_ctu_edge),
        dbk_luma_body(10, pix, xs, ys, beta, tc, no_p, no_q, max_len_p, max_len_q, hor_ctu_edge),
        dbk_luma_body(12, pix, xs, ys, beta, tc, no_p, no_q, max_len_p, max_len_q, hor_ctu_edge));
}

/* vvc_filter_template.c:633,659 and h2656_deblock_template.c:83 */
ORC_INLINE void dbk_chroma_lines(const dbk_t *d, int bd, int kind, int lines, int tc, int no_p, int no_q)
{
    for (int l = 0; l < lines; l++) {
        const int p3 = dbk_p(d, l, 3), p2 = dbk_p(d, l, 2), p1 = dbk_p(d, l, 1), p0 = dbk_p(d, l, 0);
        const int q0 = dbk_q(d, l, 0), q1 = dbk_q(d, l, 1), q2 = dbk_q(d, l, 2), q3 = dbk_q(d, l, 3);
        if (kind == 2) {            /* strong both sides */
            if (!no_p) {
                dbk_sp(d, l, 0, orc_clip3((p3 + p2 + p1 + 2 * p0 + q0 + q1 + q2 + 4) >> 3, p0 - tc, p0 + tc));
                dbk_sp(d, l, 1, orc_clip3((2 * p3 + p2 + 2 * p1 + p0 + q0 + q1 + 4) >> 3, p1 - tc, p1 + tc));
                dbk_sp(d, l, 2, orc_clip3((3 * p3 + 2 * p2 + p1 + p0 + q0 + 4) >> 3, p2 - tc, p2 + tc));
            }
            if (!no_q) {
                dbk_sq(d, l, 0, orc_clip3((p2 + p1 + p0 + 2 * q0 + q1 + q2 + q3 + 4) >> 3, q0 - tc, q0 + tc));
                dbk_sq(d, l, 1, orc_clip3((p1 + p0 + q0 + 2 * q1 + q2 + 2 * q3 + 4) >> 3, q1 - tc, q1 + tc));
                dbk_sq(d, l, 2, orc_clip3((p0 + q0 + q1 + 2 * q2 + 3 * q3 + 4) >> 3, q2 - tc, q2 + tc));
            }
        } else if (kind == 1) {     /* strong on q, one sample on p */
            if (!no_p)
                dbk_sp(d, l, 0, orc_clip3((3 * p1 + 2 * p0 + q0 + q1 + q2 + 4) >> 3, p0 - tc, p0 + tc));
            if (!no_q) {
                dbk_sq(d, l, 0, orc_clip3((2 * p1 + p0 + 2 * q0 + q1 + q2 + q3 + 4) >> 3, q0 - tc, q0 + tc));
                dbk_sq(d, l, 1, orc_clip3((p1 + p0 + q0 + 2 * q1 + q2 + 2 * q3 + 4) >> 3, q1 - tc, q1 + tc));
                dbk_sq(d, l, 2, orc_clip3((p0 + q0 + q1 + 2 * q2 + 3 * q3 + 4) >> 3, q2 - tc, q2 + tc));
            }
        } else {                    /* weak */
            const int delta = orc_clip3((((q0 - p0) * 4) + p1 - q1 + 4) >> 3, -tc, tc);
            if (!no_p)
                dbk_sp(d, l, 0, orc_clip_px(p0 + delta, bd));
            if (!no_q)
                dbk_sq(d, l, 0, orc_clip_px(q0 - delta, bd));
        }
    }
}

/* vvc_filter_template.c:681 — 8 samples per call: 2 segments x 4 lines, or 4 x 2 when `shift` (subsampled) */
ORC_INLINE void dbk_chroma_body(const int bd, uint8_t *pix, ptrdiff_t xs_b, ptrdiff_t ys_b,
    const int32_t *beta_in, const int32_t *tc_in, const uint8_t *no_p_in, const uint8_t *no_q_in,
    const uint8_t *max_len_p_in, const uint8_t *max_len_q_in, int shift)
{
    const int wide = bd > 8;
    const ptrdiff_t xs = xs_b >> wide, ys = ys_b >> wide;
    const int lines = shift ? 2 : 4, nseg = 8 / lines;
    const int l2 = shift ? 1 : 3;      /* second decision line */
    for (int seg = 0; seg < nseg; seg++) {
        const int tc = bd < 10 ? (tc_in[seg] + (1 << (9 - bd))) >> (10 - bd) : tc_in[seg] << (bd - 10);
        if (!tc)
            continue;
        dbk_t d = { pix + ((seg * lines * ys) << wide), xs, ys, wide };
        const int no_p = no_p_in[seg], no_q = no_q_in[seg];
        const int beta = beta_in[seg] << (bd - 8);
        const int beta_3 = beta >> 3, beta_2 = beta >> 2, tc25 = (tc * 5 + 1) >> 1;
        int len_p = max_len_p_in[seg], len_q = max_len_q_in[seg];
        if (!len_p || !len_q)
            continue;
        if (len_q == 3) {
            const int one = len_p == 1;
            /* with a 1-sample p side, p2/p3 alias p1 (:715-721) */
            const int p0 = dbk_p(&d, 0, 0), p1 = dbk_p(&d, 0, 1);
            const int p2 = one ? p1 : dbk_p(&d, 0, 2), p3 = one ? p1 : dbk_p(&d, 0, 3);
            const int p0n = dbk_p(&d, l2, 0), p1n = dbk_p(&d, l2, 1);
            const int p2n = one ? p1n : dbk_p(&d, l2, 2);
            const int q0 = dbk_q(&d, 0, 0), q1 = dbk_q(&d, 0, 1), q2 = dbk_q(&d, 0, 2), q3 = dbk_q(&d, 0, 3);
            const int q0n = dbk_q(&d, l2, 0), q1n = dbk_q(&d, l2, 1), q2n = dbk_q(&d, l2, 2);
            const int d0 = dbk_d2(p2, p1, p0) + dbk_d2(q2, q1, q0);
            const int d1 = dbk_d2(p2n, p1n, p0n) + dbk_d2(q2n, q1n, q0n);
            int strong = 0;
            if (d0 + d1 < beta) {
                const int p3n = one ? p1n : dbk_p(&d, l2, 3), q3n = dbk_q(&d, l2, 3);
                const int ok0 = (d0 << 1) < beta_2 && orc_abs(p3 - p0) + orc_abs(q0 - q3) < beta_3 && orc_abs(p0 - q0) < tc25;
                const int ok1 = (d1 << 1) < beta_2 && orc_abs(p3n - p0n) + orc_abs(q0n - q3n) < beta_3 && orc_abs(p0n - q0n) < tc25;
                strong = ok0 && ok1;
            }
            if (!strong)
                len_p = len_q = 1;
        }
        dbk_chroma_lines(&d, bd, (len_p == 3 && len_q == 3) ? 2 : (len_q == 3) ? 1 : 0, lines, tc, no_p, no_q);
    }
}

ORC_API void orc_lf_filter_chroma(int bd, int dir, uint8_t *pix, ptrdiff_t stride, const int32_t *beta, const int32_t *tc,
    const uint8_t *no_p, const uint8_t *no_q, const uint8_t *max_len_p, const uint8_t *max_len_q, int shift)
{
    const ptrdiff_t px = bd > 8 ? 2 : 1;
    const ptrdiff_t xs = dir == 0 ? stride : px, ys = dir == 0 ? px : stride;
    ORC_BD_SWITCH(bd,
        dbk_chroma_body(8, pix, xs, ys, beta, tc, no_p, no_q, max_len_p, max_len_q, shift),
        dbk_chroma_body(10, pix, xs, ys, beta, tc, no_p, no_q, max_len_p, max_len_q, shift),
        dbk_chroma_body(12, pix, xs, ys, beta, tc, no_p, no_q, max_len_p, max_len_q, shift));
}

/* vvc_filter_template.c:788 — luma level for the luma-adaptive deblocking offset */
ORC_API int orc_lf_ladf_level(int bd, int dir, const uint8_t *pix, ptrdiff_t stride)
{
    const int wide = bd > 8;
    const ptrdiff_t px = wide ? 2 : 1;
    const ptrdiff_t xs = (dir == 0 ? stride : px) >> wide, ys = (dir == 0 ? px : stride) >> wide;
    return (orc_ld(pix, -xs, wide) + orc_ld(pix, -xs + 3 * ys, wide) + orc_ld(pix, 0, wide) + orc_ld(pix, 3 * ys, wide)) >> 2;
}


/* ------------------------------------------------------------------ callers: one deblocking pass of a picture
 *
 * ff_vvc_deblock_vertical (vvc_filter.c:864-932) / ff_vvc_deblock_horizontal (:934-1003) with the per-CTU loop flattened to the
 * picture: for every component, every edge position on its grid and every 8-sample unit along it, the four (two for luma)
 * boundary strengths are read; where one is set the unit's QP (get_qp :850-855), beta (betatable), tc (TC_CALC :823-826) and
 * the maximum filter lengths (max_filter_length :814-821) follow, then the lf.filter_* slot runs.
 */
static const uint16_t orc_tctable[66] = {       /* Table 43, vvc_filter.c:38 */
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 3, 4, 4, 4, 4, 5, 5, 5, 5, 7, 7, 8, 9, 10,
    10, 11, 13, 14, 15, 17, 19, 21, 24, 25, 29, 33, 36, 41, 45, 51, 57, 64, 71, 80, 89, 100, 112, 125, 141, 157, 177, 198, 222, 250, 280, 314,
    352, 395,
};
static const uint8_t orc_betatable[64] = {      /* Table 43, vvc_filter.c:47 */
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 20, 22, 24,
    26, 28, 30, 32, 34, 36, 38, 40, 42, 44, 46, 48, 50, 52, 54, 56, 58, 60, 62, 64, 66, 68, 70, 72, 74, 76, 78, 80, 82, 84, 86, 88,
};

ORC_API void orc_deblock_frame_pass(int bd, const orc_deblock_frame *f)
{
    const int wide = bd > 8, vertical = f->vertical;
    const uint8_t no_pq[4] = { 0, 0, 0, 0 };
    const int8_t *dbp = (const int8_t *)(uintptr_t)f->db_params;
    const int8_t *qp_y = (const int8_t *)(uintptr_t)f->qp_y;
    const uint8_t *len_p = (const uint8_t *)(uintptr_t)f->max_len_p, *len_q = (const uint8_t *)(uintptr_t)f->max_len_q;
    const uint8_t *tbs = (const uint8_t *)(uintptr_t)f->tb_size_c;
#define TU(t, x, y) (t)[((y) >> 2) * f->min_tu_width + ((x) >> 2)]
    for (int c = 0; c < f->n_comp; c++) {
        const int hs = c ? f->hs : 0, vs = c ? f->vs : 0;
        uint8_t *plane = (uint8_t *)(uintptr_t)f->plane[c];
        const uint8_t *bs_tab = (const uint8_t *)(uintptr_t)f->bs[c];
        const int8_t *qp_c = c ? (const int8_t *)(uintptr_t)f->qp_c[c - 1] : NULL;
        /* across the edges: grid; along them: units of 8 samples of this component */
        const int grid = c ? (8 << (vertical ? hs : vs)) : 4;
        const int step = 8 << (vertical ? vs : hs);
        const int across_end = vertical ? f->width : f->height, along_end = vertical ? f->height : f->width;
        const int nseg = 8 >> (2 - (vertical ? vs : hs));
        for (int e = grid; e < across_end; e += grid)
            for (int u = 0; u < along_end; u += step) {
                int32_t bs[4], beta[4] = { 0, 0, 0, 0 }, tc[4];
                uint8_t max_len_p[4] = { 0, 0, 0, 0 }, max_len_q[4] = { 0, 0, 0, 0 };
                int any = 0;
                const int ux = vertical ? e : u, uy = vertical ? u : e;
                const int hor_ctu_edge = !vertical && !(e % (1 << f->ctb_log2));
                const int ctb = (ux >> f->ctb_log2) + (uy >> f->ctb_log2) * f->ctb_width;
                const int beta_offset = dbp[ctb * 6 + c], tc_offset = dbp[ctb * 6 + 3 + c];
                for (int i = 0; i < nseg; i++) {
                    const int x = vertical ? e : u + 4 * i, y = vertical ? u + 4 * i : e;
                    int qp = 0;
                    bs[i] = (vertical ? y < f->height : x < f->width) ? TU(bs_tab, x, y) : 0;
                    if (bs[i]) {
                        const int xp = x - vertical, yp = y - !vertical;
                        if (!c) {
                            const int a = qp_y[(xp >> f->min_cb_log2) + (yp >> f->min_cb_log2) * f->min_cb_width];
                            const int b = qp_y[(x >> f->min_cb_log2) + (y >> f->min_cb_log2) * f->min_cb_width];
                            qp = (a + b + 1) >> 1;
                            if (f->ladf_enabled) {
                                const uint8_t *src = plane + (ptrdiff_t)y * f->stride[0] + ((ptrdiff_t)x << wide);
                                const int level = orc_lf_ladf_level(bd, vertical, src, f->stride[0]);
                                int qp_offset = f->ladf_lowest_qp_offset;
                                for (int k = 0; k < f->num_ladf_intervals - 1 && level > f->ladf_lower_bound[k + 1]; k++)
                                    qp_offset = f->ladf_qp_offset[k];
                                qp += qp_offset;
                            }
                            max_len_p[i] = TU(len_p, x, y);
                            max_len_q[i] = TU(len_q, x, y);
                        } else {
                            qp = (TU(qp_c, xp, yp) + TU(qp_c, x, y) - 2 * f->qp_bd_offset + 1) >> 1;
                            const int size_p = TU(tbs, xp, yp), size_q = TU(tbs, x, y);
                            if (size_p >= 8 && size_q >= 8) {
                                max_len_p[i] = max_len_q[i] = 3;
                                if (hor_ctu_edge)
                                    max_len_p[i] = 1;
                            } else {
                                max_len_p[i] = max_len_q[i] = bs[i] == 2;
                            }
                        }
                        beta[i] = orc_betatable[orc_clip3(qp + beta_offset, 0, 63)];
                        any = 1;
                    }
                    tc[i] = bs[i] ? orc_tctable[orc_clip3(qp + 2 * (bs[i] - 1) + (tc_offset & -2), 0, 65)] : 0;
                }
                if (!any)
                    continue;
                uint8_t *src = plane + (ptrdiff_t)(uy >> vs) * f->stride[c] + ((ptrdiff_t)(ux >> hs) << wide);
                if (!c)
                    orc_lf_filter_luma(bd, vertical, src, f->stride[0], beta, tc, no_pq, no_pq, max_len_p, max_len_q, hor_ctu_edge);
                else
                    orc_lf_filter_chroma(bd, vertical, src, f->stride[c], beta, tc, no_pq, no_pq, max_len_p, max_len_q, vertical ? vs : hs);
            }
    }
#undef TU
}


/* ------------------------------------------------------------------ callers: SAO of a picture
 *
 * ff_vvc_sao_filter (vvc_filter.c:154-300) per CTB, flattened: edges[] = picture borders (:172-175), vert / horiz / diag_edge from
 * slice indices and tile boundaries when filtering across them is off (:177-215), then per component band_filter, or the
 * padded copy (:243-287, here taken from the pre-SAO plane instead of the saved border lines) + edge_filter + edge_restore.
 */
#define SAO_EDGE_STRIDE (2 * ORC_PB + 64)          /* 2*MAX_PB_SIZE + AV_INPUT_BUFFER_PADDING_SIZE, bytes (vvcdsp.h:140) */
ORC_API void orc_sao_frame_pass(int bd, const orc_sao_frame *f)
{
    const int wide = bd > 8;
    const orc_sao_ctb *tab = (const orc_sao_ctb *)(uintptr_t)f->sao;
    const int16_t *slice = (const int16_t *)(uintptr_t)f->slice_idx;
    const int16_t *col_bd = (const int16_t *)(uintptr_t)f->ctb_to_col_bd, *row_bd = (const int16_t *)(uintptr_t)f->ctb_to_row_bd;
    const int restore = f->no_tile_filter || !f->lfase;
    static _Thread_local uint8_t buf[(ORC_PB + 2) * SAO_EDGE_STRIDE];
#define SL(x, y) slice[(y) * f->ctb_width + (x)]
    for (int yc = 0; yc < f->ctb_height; yc++)
        for (int xc = 0; xc < f->ctb_width; xc++) {
            const orc_sao_ctb *sao = tab + yc * f->ctb_width + xc;
            int edges[4] = { xc == 0, yc == 0, xc == f->ctb_width - 1, yc == f->ctb_height - 1 };
            uint8_t vert_edge[2] = { 0, 0 }, horiz_edge[2] = { 0, 0 }, diag_edge[4] = { 0, 0, 0, 0 };
            int lt = 0, rt = 0, ut = 0, bt = 0;
            if (restore) {
                const int me = SL(xc, yc);
                if (!edges[0]) { lt = f->no_tile_filter && col_bd[xc] == xc; vert_edge[0] = (!f->lfase && me != SL(xc - 1, yc)) || lt; }
                if (!edges[2]) { rt = f->no_tile_filter && col_bd[xc] != col_bd[xc + 1]; vert_edge[1] = (!f->lfase && me != SL(xc + 1, yc)) || rt; }
                if (!edges[1]) { ut = f->no_tile_filter && row_bd[yc] == yc; horiz_edge[0] = (!f->lfase && me != SL(xc, yc - 1)) || ut; }
                if (!edges[3]) { bt = f->no_tile_filter && row_bd[yc] != row_bd[yc + 1]; horiz_edge[1] = (!f->lfase && me != SL(xc, yc + 1)) || bt; }
                if (!edges[0] && !edges[1]) diag_edge[0] = (!f->lfase && me != SL(xc - 1, yc - 1)) || lt || ut;
                if (!edges[1] && !edges[2]) diag_edge[1] = (!f->lfase && me != SL(xc + 1, yc - 1)) || rt || ut;
                if (!edges[2] && !edges[3]) diag_edge[2] = (!f->lfase && me != SL(xc + 1, yc + 1)) || rt || bt;
                if (!edges[0] && !edges[3]) diag_edge[3] = (!f->lfase && me != SL(xc - 1, yc + 1)) || lt || bt;
            }
            for (int c = 0; c < f->n_comp; c++) {
                const int hs = c ? f->hs : 0, vs = c ? f->vs : 0;
                const int pw = f->width >> hs, ph = f->height >> vs;
                const int x0 = (xc << f->ctb_log2) >> hs, y0 = (yc << f->ctb_log2) >> vs;
                const int w = orc_min((1 << f->ctb_log2) >> hs, pw - x0), h = orc_min((1 << f->ctb_log2) >> vs, ph - y0);
                const uint8_t *splane = (const uint8_t *)(uintptr_t)f->src[c];
                uint8_t *dst = (uint8_t *)(uintptr_t)f->dst[c] + (ptrdiff_t)y0 * f->dst_stride[c] + ((ptrdiff_t)x0 << wide);
                const uint8_t *src = splane + (ptrdiff_t)y0 * f->src_stride[c] + ((ptrdiff_t)x0 << wide);
                if (sao->type_idx[c] == 1) {
                    orc_sao_band_filter(bd, dst, src, f->dst_stride[c], f->src_stride[c], sao->offset_val[c], sao->band_position[c], w, h);
                } else if (sao->type_idx[c] == 2) {
                    /* the CTB with the one-sample apron that exists inside the picture (:243-287) */
                    uint8_t *b0 = buf + SAO_EDGE_STRIDE + 64;
                    for (int y = -1; y <= h; y++)
                        for (int x = -1; x <= w; x++) {
                            const int sx = x0 + x, sy = y0 + y;
                            if (sx < 0 || sy < 0 || sx >= pw || sy >= ph)
                                continue;
                            orc_st(b0 + (ptrdiff_t)y * SAO_EDGE_STRIDE, x, orc_ld(splane + (ptrdiff_t)sy * f->src_stride[c], sx, wide), wide);
                        }
                    orc_sao_edge_filter(bd, dst, b0, f->dst_stride[c], sao->offset_val[c], sao->eo_class[c], w, h);
                    orc_sao_edge_restore(bd, restore, dst, b0, f->dst_stride[c], SAO_EDGE_STRIDE, sao->offset_val[c], sao->eo_class[c], edges,
                                         w, h, vert_edge, horiz_edge, diag_edge);
                } else {
                    for (int y = 0; y < h; y++)
                        memcpy(dst + (ptrdiff_t)y * f->dst_stride[c], src + (ptrdiff_t)y * f->src_stride[c], (size_t)w << wide);
                }
            }
        }
#undef SL
}


/* ------------------------------------------------------------------ callers: ALF of a picture
 *
 * ff_vvc_alf_filter (vvc_filter.c:1254-1318) per CTB, flattened.  alf_prepare_buffer (:1105-1137) builds the padded source: on
 * a side whose edges[] flag is set the CTB's own border samples are replicated, otherwise the neighbouring samples are copied —
 * i.e. per axis, coordinates are clamped to the CTB on flagged sides.  Then alf_filter_luma (:1171-1186: classify,
 * recon_coeff_and_clip, filter[LUMA]), alf_filter_chroma (:1195-1210), alf_filter_cc (:1212-1227).
 */
extern const int16_t orc_tab_alf_fix_filt_coeff[64 * 12];
extern const uint8_t orc_tab_alf_class_to_filt_map[16 * 25], orc_tab_alf_aps_class_to_filt_map[25];

#define ALF_PAD 8
#define ALF_PSTRIDE (ORC_PB + 2 * ALF_PAD)
static void alf_padded(int wide, uint8_t *buf, const uint8_t *plane, ptrdiff_t stride, int x0, int y0, int w, int h, int pw, int ph,
                       const int *edges, int border)
{
    for (int y = -border; y < h + border; y++)
        for (int x = -border; x < w + border; x++) {
            int sx = x0 + x, sy = y0 + y;
            if (x < 0 && edges[0]) sx = x0;
            if (x >= w && edges[2]) sx = x0 + w - 1;
            if (y < 0 && edges[1]) sy = y0;
            if (y >= h && edges[3]) sy = y0 + h - 1;
            sx = orc_clip3(sx, 0, pw - 1);
            sy = orc_clip3(sy, 0, ph - 1);
            orc_st(buf, (ptrdiff_t)(y + ALF_PAD) * ALF_PSTRIDE + x + ALF_PAD, orc_ld(plane, (ptrdiff_t)sy * (stride >> wide) + sx, wide), wide);
        }
}

ORC_API void orc_alf_frame_pass(int bd, const orc_alf_frame *f)
{
    const int wide = bd > 8, ctb_size = 1 << f->ctb_log2;
    const orc_alf_ctb *tab = (const orc_alf_ctb *)(uintptr_t)f->alf;
    const orc_alf_slice *slices = (const orc_alf_slice *)(uintptr_t)f->slices;
    const int16_t *slice = (const int16_t *)(uintptr_t)f->slice_idx;
    const int16_t *col_bd = (const int16_t *)(uintptr_t)f->ctb_to_col_bd, *row_bd = (const int16_t *)(uintptr_t)f->ctb_to_row_bd;
    static _Thread_local uint8_t pad_l[ALF_PSTRIDE * ALF_PSTRIDE * 2], pad_c[ALF_PSTRIDE * ALF_PSTRIDE * 2];
    static _Thread_local int cls[1024], tr[1024], grad[(ORC_PB + 4) * (ORC_PB + 4)];
    static _Thread_local int16_t coeff[1024 * 12], clip[1024 * 12];
    static const uint8_t zero_clip[25 * 12] = { 0 };
#define SL(x, y) slice[(y) * f->ctb_width + (x)]
    for (int yc = 0; yc < f->ctb_height; yc++)
        for (int xc = 0; xc < f->ctb_width; xc++) {
            const orc_alf_ctb *alf = tab + yc * f->ctb_width + xc;
            const orc_alf_slice *sl = slices + SL(xc, yc);
            int edges[4] = { xc == 0, yc == 0, xc == f->ctb_width - 1, yc == f->ctb_height - 1 };
            if (!f->lfate) {
                edges[0] = edges[0] || col_bd[xc] == xc;                      /* BOUNDARY_LEFT_TILE */
                edges[1] = edges[1] || row_bd[yc] == yc;                      /* BOUNDARY_UPPER_TILE */
                edges[2] = edges[2] || col_bd[xc] != col_bd[xc + 1];
                edges[3] = edges[3] || row_bd[yc] != row_bd[yc + 1];
            }
            if (!f->lfase) {
                edges[0] = edges[0] || SL(xc, yc) != SL(xc - 1, yc);          /* BOUNDARY_LEFT_SLICE (edges[0] already set at xc == 0) */
                edges[1] = edges[1] || SL(xc, yc) != SL(xc, yc - 1);
                edges[2] = edges[2] || SL(xc, yc) != SL(xc + 1, yc);
                edges[3] = edges[3] || SL(xc, yc) != SL(xc, yc + 1);
            }
            for (int c = 0; c < f->n_comp; c++) {
                const int hs = c ? f->hs : 0, vs = c ? f->vs : 0;
                const int pw = f->width >> hs, ph = f->height >> vs;
                const int x0 = (xc * ctb_size) >> hs, y0 = (yc * ctb_size) >> vs;
                const int w = orc_min(pw - x0, ctb_size >> hs), h = orc_min(ph - y0, ctb_size >> vs);
                const uint8_t *splane = (const uint8_t *)(uintptr_t)f->src[c];
                uint8_t *dst = (uint8_t *)(uintptr_t)f->dst[c] + (ptrdiff_t)y0 * f->dst_stride[c] + ((ptrdiff_t)x0 << wide);
                uint8_t *pad = c ? pad_c : pad_l;
                const uint8_t *psrc = pad + (((ptrdiff_t)ALF_PAD * ALF_PSTRIDE + ALF_PAD) << wide);
                const ptrdiff_t pstride = (ptrdiff_t)ALF_PSTRIDE << wide;
                alf_padded(wide, pad, splane, f->src_stride[c], x0, y0, w, h, pw, ph, edges, c ? 2 : 3);
                if (!alf->ctb_flag[c]) {
                    for (int y = 0; y < h; y++)
                        memcpy(dst + (ptrdiff_t)y * f->dst_stride[c], splane + (ptrdiff_t)(y0 + y) * f->src_stride[c] + ((ptrdiff_t)x0 << wide), (size_t)w << wide);
                } else if (!c) {
                    const int16_t *coeff_set;
                    const uint8_t *clip_idx_set, *class_to_filt;
                    if (alf->filt_set_idx_y < 16) {
                        coeff_set = orc_tab_alf_fix_filt_coeff;
                        clip_idx_set = zero_clip;
                        class_to_filt = orc_tab_alf_class_to_filt_map + alf->filt_set_idx_y * 25;
                    } else {
                        coeff_set = (const int16_t *)(uintptr_t)sl->luma_coeff[alf->filt_set_idx_y - 16];
                        clip_idx_set = (const uint8_t *)(uintptr_t)sl->luma_clip_idx[alf->filt_set_idx_y - 16];
                        class_to_filt = orc_tab_alf_aps_class_to_filt_map;
                    }
                    const int vb_pos = ctb_size - 4, n = (w / 4) * (h / 4);
                    orc_alf_classify(bd, cls, tr, psrc, pstride, w, h, vb_pos, grad);
                    orc_alf_recon_coeff_and_clip(bd, coeff, clip, cls, tr, n, coeff_set, clip_idx_set, class_to_filt);
                    orc_alf_filter_luma(bd, dst, f->dst_stride[0], psrc, pstride, w, h, coeff, clip, vb_pos);
                } else {
                    const int idx = alf->alt_idx[c - 1];
                    const int16_t *cf = (const int16_t *)(uintptr_t)sl->chroma_coeff + idx * 6;
                    const uint8_t *ci = (const uint8_t *)(uintptr_t)sl->chroma_clip_idx + idx * 6;
                    static const int off[4] = { 0, 3, 5, 7 };
                    int16_t cl[6];
                    for (int i = 0; i < 6; i++) cl[i] = (int16_t)(1 << (bd - off[ci[i]]));
                    orc_alf_filter_chroma(bd, dst, f->dst_stride[c], psrc, pstride, w, h, cf, cl, (ctb_size >> vs) - 2);
                }
                if (c && alf->cc_idc[c - 1] && sl->cc_coeff[c - 1]) {
                    /* the luma buffer still holds this CTB's padded luma (component 0 ran first) */
                    const int16_t *cf = (const int16_t *)(uintptr_t)sl->cc_coeff[c - 1] + (alf->cc_idc[c - 1] - 1) * 7;
                    const uint8_t *lsrc = pad_l + (((ptrdiff_t)ALF_PAD * ALF_PSTRIDE + ALF_PAD) << wide);
                    orc_alf_filter_cc(bd, dst, f->dst_stride[c], lsrc, pstride, w, h, hs, vs, cf, ctb_size - 4);
                }
            }
        }
#undef SL
}

/* ------------------------------------------------------------------ callers: boundary strengths of a picture
 *
 * vvc_deblock_bs (vvc_filter.c:756-783) for every CTB and both directions, in the reference's own shape: walk the transform
 * units of each tree, and let each one write the entries of its left / upper edge and of its internal sub-block edges.
 * dir 1 = vertical edges (vvc_deblock_bs_luma_vertical :477, _chroma_vertical :642, subblock :399),
 * dir 0 = horizontal edges (:560, :698, :437).  The output tables are cleared first (the decoder clears them per frame).
 */
static int bs_mv_far(const int32_t *a, const int32_t *b) { return abs(a[0] - b[0]) >= 8 || abs(a[1] - b[1]) >= 8; }

/* boundary_strength (:308-372); rpl / nrpl = int32 [2][32] POC lists */
static int bs_boundary_strength(const orc_mvfield *curr, const orc_mvfield *neigh, const int32_t *rpl, const int32_t *nrpl)
{
    if (curr->pred_flag == 3 && neigh->pred_flag == 3) {
        const int c0 = rpl[curr->ref_idx[0]], c1 = rpl[32 + curr->ref_idx[1]];
        const int n0 = nrpl[neigh->ref_idx[0]], n1 = nrpl[32 + neigh->ref_idx[1]];
        if (c0 == n0 && c0 == c1 && n0 == n1)
            return (bs_mv_far(neigh->mv[0], curr->mv[0]) || bs_mv_far(neigh->mv[1], curr->mv[1])) &&
                   (bs_mv_far(neigh->mv[1], curr->mv[0]) || bs_mv_far(neigh->mv[0], curr->mv[1]));
        if (n0 == c0 && n1 == c1)
            return bs_mv_far(neigh->mv[0], curr->mv[0]) || bs_mv_far(neigh->mv[1], curr->mv[1]);
        if (n1 == c0 && n0 == c1)
            return bs_mv_far(neigh->mv[1], curr->mv[0]) || bs_mv_far(neigh->mv[0], curr->mv[1]);
        return 1;
    }
    if (curr->pred_flag != 3 && neigh->pred_flag != 3) {
        const int la = (curr->pred_flag & 1) ? 0 : 1, lb = (neigh->pred_flag & 1) ? 0 : 1;
        const int ref_a = rpl[la * 32 + curr->ref_idx[la]], ref_b = nrpl[lb * 32 + neigh->ref_idx[lb]];
        if (ref_a != ref_b)
            return 1;
        return bs_mv_far(curr->mv[la], neigh->mv[lb]);
    }
    return 1;
}

typedef struct bs_ctx {
    const orc_bs_frame *f;
    const orc_mvfield *mvf;
    const int32_t *rpl;               /* the current CTU's slice */
    int left_slice, upper_slice, left_tile, upper_tile;     /* lc->boundary_flags of the current CTU (vvc_ctu.c:2481-2489) */
    int ctb_x, ctb_y;
} bs_ctx;

#define BS_U8(p)  ((uint8_t *)(uintptr_t)(p))
#define BS_I32(p) ((const int32_t *)(uintptr_t)(p))
#define BS_TU(f, x, y) (((y) >> 2) * (f)->min_tu_width + ((x) >> 2))

static void bs_luma_tu(const bs_ctx *c, int dir, int x0, int y0, int width, int height)
{
    const orc_bs_frame *f = c->f;
    const int mpw = f->min_pu_width;
    const int is_intra = c->mvf[(y0 >> 2) * mpw + (x0 >> 2)].pred_flag == 0;
    const int off_q = (y0 >> f->min_cb_log2) * f->min_cb_width + (x0 >> f->min_cb_log2);
    const int cb_x = BS_I32(f->cb_pos_x)[off_q], cb_y = BS_I32(f->cb_pos_y)[off_q];
    const int cb_size = dir ? BS_U8(f->cb_width)[off_q] : BS_U8(f->cb_height)[off_q];
    const int sb_cu = !is_intra && (BS_U8(f->msf)[off_q] || BS_U8(f->iaf)[off_q]);
    const int has_sb = sb_cu && cb_size > 8;
    const int a0 = dir ? x0 : y0, along = dir ? height : width, across = dir ? width : height;
    const int ctb_mask = (1 << f->ctb_log2) - 1;
    uint8_t *tab_bs = BS_U8(f->bs[dir][0]), *tab_p = BS_U8(f->max_len_p[dir]), *tab_q = BS_U8(f->max_len_q[dir]);
    const uint8_t *tb_size = dir ? BS_U8(f->tb_width[0]) : BS_U8(f->tb_height[0]);

    int boundary = a0 > 0 && !(a0 & 3);
    if (boundary && ((!f->lfase && (dir ? c->left_slice : c->upper_slice) && !(a0 & ctb_mask)) ||
                     (!f->lfate && (dir ? c->left_tile : c->upper_tile) && !(a0 & ctb_mask))))
        boundary = 0;
    if (boundary) {
        /* the neighbour's lists: those of the CTB the P side lies in (ff_vvc_get_ref_list, vvc_refs.c:66) */
        const int px0 = dir ? x0 - 1 : x0, py0 = dir ? y0 : y0 - 1;
        const int nslice = ((const int16_t *)(uintptr_t)f->slice_idx)[(py0 >> f->ctb_log2) * f->ctb_width + (px0 >> f->ctb_log2)];
        const int32_t *nrpl = ((dir ? c->left_slice : c->upper_slice) ? BS_I32(f->ref_poc) + nslice * 64 : c->rpl);
        for (int i = 0; i < along; i += 4) {
            const int qx = dir ? x0 : x0 + i, qy = dir ? y0 + i : y0;
            const int px = dir ? qx - 1 : qx, py = dir ? qy : qy - 1;
            const orc_mvfield *neigh = &c->mvf[(py >> 2) * mpw + (px >> 2)], *curr = &c->mvf[(qy >> 2) * mpw + (qx >> 2)];
            const int tq = BS_TU(f, qx, qy), tp = BS_TU(f, px, py);
            const int off_c = (dir ? cb_x : cb_y) - a0;
            int bs, len_p, len_q;
            if (BS_U8(f->pcmf[0])[tp] && BS_U8(f->pcmf[0])[tq])
                bs = 0;
            else if (curr->pred_flag == 0 || neigh->pred_flag == 0 || curr->ciip_flag || neigh->ciip_flag)
                bs = 2;
            else if (BS_U8(f->tu_coded_flag[0])[tq] || BS_U8(f->tu_coded_flag[0])[tp])
                bs = 1;
            else if (off_c && ((off_c % 8) || !has_sb))
                bs = 0;
            else
                bs = bs_boundary_strength(curr, neigh, c->rpl, nrpl);
            tab_bs[tq] = (uint8_t)bs;
            /* derive_max_filter_length_luma (:374-397) */
            {
                const int size_p = tb_size[tp], size_q = tb_size[tq];
                const int off_p = (py >> f->min_cb_log2) * f->min_cb_width + (px >> f->min_cb_log2);
                if (size_p <= 4 || size_q <= 4) {
                    len_p = len_q = 1;
                } else {
                    len_p = len_q = 3;
                    if (size_p >= 32) len_p = 7;
                    if (size_q >= 32) len_q = 7;
                }
                if (has_sb) len_q = orc_min(5, len_q);
                if (BS_U8(f->msf)[off_p] || BS_U8(f->iaf)[off_p]) len_p = orc_min(5, len_p);
            }
            tab_p[tq] = (uint8_t)len_p;
            tab_q[tq] = (uint8_t)len_q;
        }
    }
    if (sb_cu) {
        /* vvc_deblock_subblock_bs_vertical / _horizontal (:399-475): internal edges every 8 samples from the coding block origin */
        for (int i = 8 - ((a0 - (dir ? cb_x : cb_y)) % 8); i < across; i += 8)
            for (int j = 0; j < along; j += 4) {
                const int qx = dir ? x0 + i : x0 + j, qy = dir ? y0 + j : y0 + i;
                const int px = dir ? qx - 1 : qx, py = dir ? qy : qy - 1;
                const orc_mvfield *neigh = &c->mvf[(py >> 2) * mpw + (px >> 2)], *curr = &c->mvf[(qy >> 2) * mpw + (qx >> 2)];
                const int t = BS_TU(f, qx, qy);
                const int len = (i == 4 || i == across - 4) ? 1 : (i == 8 || i == across - 8) ? 2 : 3;
                tab_bs[t] = (uint8_t)bs_boundary_strength(curr, neigh, c->rpl, c->rpl);
                tab_p[t] = tab_q[t] = (uint8_t)len;
            }
    }
}

static void bs_chroma_tu(const bs_ctx *c, int dir, int x0, int y0, int width, int height)
{
    const orc_bs_frame *f = c->f;
    const int mpw = f->min_pu_width;
    const int a0 = dir ? x0 : y0, along = dir ? height : width;
    const int ctb_mask = (1 << f->ctb_log2) - 1;
    int boundary = a0 > 0 && !(a0 & ((8 << (dir ? f->hs : f->vs)) - 1));
    if (boundary && ((!f->lfase && (dir ? c->left_slice : c->upper_slice) && !(a0 & ctb_mask)) ||
                     (!f->lfate && (dir ? c->left_tile : c->upper_tile) && !(a0 & ctb_mask))))
        boundary = 0;
    if (!boundary)
        return;
    for (int i = 0; i < along; i += 2) {
        const int qx = dir ? x0 : x0 + i, qy = dir ? y0 + i : y0;
        const int px = dir ? qx - 1 : qx, py = dir ? qy : qy - 1;
        const orc_mvfield *neigh = &c->mvf[(py >> 2) * mpw + (px >> 2)], *curr = &c->mvf[(qy >> 2) * mpw + (qx >> 2)];
        const int tq = BS_TU(f, qx, qy), tp = BS_TU(f, px, py);
        const int pcmf = BS_U8(f->pcmf[1])[tp] && BS_U8(f->pcmf[1])[tq];
        for (int k = 1; k <= 2; k++) {
            const int cbf = BS_U8(f->tu_coded_flag[k])[tp] | BS_U8(f->tu_coded_flag[k])[tq] | BS_U8(f->tu_joint_cbcr)[tp] | BS_U8(f->tu_joint_cbcr)[tq];
            int bs = 0;
            if (pcmf)
                bs = 0;
            else if (curr->pred_flag == 0 || neigh->pred_flag == 0 || curr->ciip_flag || neigh->ciip_flag)
                bs = 2;
            else if (cbf)
                bs = 1;
            BS_U8(f->bs[dir][k])[tq] = (uint8_t)bs;
        }
    }
}

ORC_API void orc_deblock_bs_pass(const orc_bs_frame *f)
{
    const int ctb_size = 1 << f->ctb_log2;
    const int ctb_height = (f->height + ctb_size - 1) >> f->ctb_log2;
    const int16_t *slice = (const int16_t *)(uintptr_t)f->slice_idx;
    const int16_t *col_bd = (const int16_t *)(uintptr_t)f->ctb_to_col_bd, *row_bd = (const int16_t *)(uintptr_t)f->ctb_to_row_bd;
    const size_t n_tu = (size_t)f->min_tu_width * (f->height >> 2);
    for (int dir = 0; dir < 2; dir++) {
        for (int k = 0; k < (f->n_comp >= 3 ? 3 : 1); k++) memset(BS_U8(f->bs[dir][k]), 0, n_tu);
        memset(BS_U8(f->max_len_p[dir]), 0, n_tu);
        memset(BS_U8(f->max_len_q[dir]), 0, n_tu);
    }
    for (int ry = 0; ry < ctb_height; ry++)
        for (int rx = 0; rx < f->ctb_width; rx++) {
            const int rs = ry * f->ctb_width + rx;
            bs_ctx c = { f, (const orc_mvfield *)(uintptr_t)f->mvf, BS_I32(f->ref_poc) + slice[rs] * 64, 0, 0, 0, 0, rx, ry };
            c.left_tile   = rx > 0 && col_bd[rx] != col_bd[rx - 1];
            c.left_slice  = rx > 0 && slice[rs] != slice[rs - 1];
            c.upper_tile  = ry > 0 && row_bd[ry] != row_bd[ry - 1];
            c.upper_slice = ry > 0 && slice[rs] != slice[rs - f->ctb_width];
            const int x0 = rx << f->ctb_log2, y0 = ry << f->ctb_log2;
            const int x_end = orc_min(x0 + ctb_size, f->width) >> 2, y_end = orc_min(y0 + ctb_size, f->height) >> 2;
            for (int dir = 1; dir >= 0; dir--)
                for (int tree = 0; tree < (f->n_comp >= 3 ? 2 : 1); tree++) {
                    const int hs = tree ? f->hs : 0, vs = tree ? f->vs : 0;
                    for (int y = y0 >> 2; y < y_end; y++)
                        for (int x = x0 >> 2; x < x_end; x++) {
                            const int off = y * f->min_tu_width + x;
                            if ((BS_I32(f->tb_pos_x0[tree])[off] >> 2) != x || (BS_I32(f->tb_pos_y0[tree])[off] >> 2) != y)
                                continue;
                            const int w = BS_U8(f->tb_width[tree])[off] << hs, h = BS_U8(f->tb_height[tree])[off] << vs;
                            if (tree) bs_chroma_tu(&c, dir, x << 2, y << 2, w, h);
                            else      bs_luma_tu(&c, dir, x << 2, y << 2, w, h);
                        }
                }
        }
}
