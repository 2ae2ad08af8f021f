/*
 * oracle/orc_inter.c — CPU restatement of the inter-prediction DSP slots.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see orc_common.h).
 *
 * Follows, by reading:
 *   libavcodec/h26x/h2656_inter_template.c  (put/put_uni/put_uni_w x {pixels,h,v,hv}, luma :29-334, chroma :336-577)
 *   libavcodec/vvc/vvc_inter_template.c     (avg :25, w_avg :42, ciip :60, gpm :78, bdof/prof :101-317, dmvr :324-413)
 *   libavcodec/vvc/vvcdsp.c                 (pad_int16 :29, vvc_sad :49)
 */
#include "vvc_oracle.h"
#include "orc_common.h"

/* ------------------------------------------------------------------ separable DCTIF (8-tap luma / 4-tap chroma) */

/* Σ f[k] * px[o + (k - lead) * step]; lead = 3 (luma) or 1 (chroma): h2656_inter_template.c:87,:336 */
ORC_INLINE int fir_pixels(const uint8_t *src, ptrdiff_t o, ptrdiff_t step, const int8_t *f, int ntap, int wide)
{
    const int lead = ntap == 8 ? 3 : 1;
    int acc = 0;
    for (int k = 0; k < ntap; k++)
        acc += f[k] * orc_ld(src, o + (k - lead) * step, wide);
    return acc;
}

ORC_INLINE int fir_i16(const int16_t *t, ptrdiff_t o, ptrdiff_t step, const int8_t *f, int ntap)
{
    const int lead = ntap == 8 ? 3 : 1;
    int acc = 0;
    for (int k = 0; k < ntap; k++)
        acc += f[k] * t[o + (k - lead) * step];
    return acc;
}

enum { MC_PUT, MC_UNI, MC_UNI_W };

/*
 * All twelve (kind x frac) variants of one component share this body.  `val` is the 14-bit-scaled
 * intermediate; what happens to it afterwards depends on `kind`:
 *   put      : stored to int16 (row stride 128)                     — :29,:97,:112,:127
 *   put_uni  : clip_px((val + 2^(13-bd)) >> (14-bd))                — :154-245 (integer position = plain copy :44)
 *   put_uni_w: clip_px(((val*wx + 2^(shift-1)) >> shift) + ox*2^(bd-8)), shift = denom + 14 - bd — :60,:247-334
 */
ORC_INLINE void mc_body(const int bd, const int ntap, const int kind, const int vfrac, const int hfrac,
    int16_t *dst16, uint8_t *dst, ptrdiff_t dst_stride, const uint8_t *src, ptrdiff_t src_stride,
    int h, int denom, int wx, int ox_in, const int8_t *hf, const int8_t *vf, int w)
{
    const int wide = bd > 8;
    const int lead = ntap == 8 ? 3 : 1;
    const ptrdiff_t ss = src_stride >> wide;
    const int sh_uni = 14 - bd, off_uni = 1 << (sh_uni - 1);
    const int sh_w = denom + 14 - bd, off_w = 1 << (sh_w - 1);
    const int ox = ox_in * (1 << (bd - 8));
    int16_t tmp[(ORC_PB + 7) * ORC_PB];

    if (vfrac && hfrac) {
        /* horizontal pass over h + ntap - 1 rows into an int16 plane of row stride 128 (:135-141) */
        for (int y = 0; y < h + ntap - 1; y++)
            for (int x = 0; x < w; x++)
                tmp[y * ORC_PB + x] = (int16_t)(fir_pixels(src, (ptrdiff_t)(y - lead) * ss + x, 1, hf, ntap, wide) >> (bd - 8));
    }
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            const ptrdiff_t o = (ptrdiff_t)y * ss + x;
            int val;
            if (vfrac && hfrac)
                val = fir_i16(tmp, (ptrdiff_t)(y + lead) * ORC_PB + x, ORC_PB, vf, ntap) >> 6;
            else if (hfrac)
                val = fir_pixels(src, o, 1, hf, ntap, wide) >> (bd - 8);
            else if (vfrac)
                val = fir_pixels(src, o, ss, vf, ntap, wide) >> (bd - 8);
            else
                val = orc_ld(src, o, wide) << (14 - bd);
            if (kind == MC_PUT)
                dst16[y * ORC_PB + x] = (int16_t)val;
            else if (kind == MC_UNI)
                orc_st(dst + y * dst_stride, x, (vfrac || hfrac) ? orc_clip_px((val + off_uni) >> sh_uni, bd) : orc_ld(src, o, wide), wide);
            else
                orc_st(dst + y * dst_stride, x, orc_clip_px(((val * wx + off_w) >> sh_w) + ox, bd), wide);
        }
    }
}

#define MC_CALL(BD) mc_body(BD, chroma ? 4 : 8, kind, vfrac, hfrac, dst16, dst, dst_stride, src, src_stride, height, denom, wx, ox, hf, vf, width)
static void mc_dispatch(int bd, int chroma, int kind, int vfrac, int hfrac, int16_t *dst16, uint8_t *dst, ptrdiff_t dst_stride,
    const uint8_t *src, ptrdiff_t src_stride, int height, int denom, int wx, int ox, const int8_t *hf, const int8_t *vf, int width)
{
    ORC_BD_SWITCH(bd, MC_CALL(8), MC_CALL(10), MC_CALL(12));
}

ORC_API void orc_put(int bd, int chroma, int vfrac, int hfrac, int16_t *dst, const uint8_t *src, ptrdiff_t src_stride,
    int height, const int8_t *hf, const int8_t *vf, int width)
{
    mc_dispatch(bd, chroma, MC_PUT, vfrac, hfrac, dst, NULL, 0, src, src_stride, height, 0, 0, 0, hf, vf, width);
}

ORC_API void orc_put_uni(int bd, int chroma, int vfrac, int hfrac, uint8_t *dst, ptrdiff_t dst_stride,
    const uint8_t *src, ptrdiff_t src_stride, int height, const int8_t *hf, const int8_t *vf, int width)
{
    mc_dispatch(bd, chroma, MC_UNI, vfrac, hfrac, NULL, dst, dst_stride, src, src_stride, height, 0, 0, 0, hf, vf, width);
}

ORC_API void orc_put_uni_w(int bd, int chroma, int vfrac, int hfrac, uint8_t *dst, ptrdiff_t dst_stride,
    const uint8_t *src, ptrdiff_t src_stride, int height, int denom, int wx, int ox,
    const int8_t *hf, const int8_t *vf, int width)
{
    mc_dispatch(bd, chroma, MC_UNI_W, vfrac, hfrac, NULL, dst, dst_stride, src, src_stride, height, denom, wx, ox, hf, vf, width);
}

/* ------------------------------------------------------------------ bi-prediction blends */

/* vvc_inter_template.c:25 */
ORC_API void orc_avg(int bd, uint8_t *dst, ptrdiff_t dst_stride, const int16_t *src0, const int16_t *src1, int width, int height)
{
    const int wide = bd > 8, shift = orc_max(3, 15 - bd), off = 1 << (shift - 1);
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++)
            orc_st(dst + y * dst_stride, x, orc_clip_px((src0[y * ORC_PB + x] + src1[y * ORC_PB + x] + off) >> shift, bd), wide);
}

/* vvc_inter_template.c:42 */
ORC_API void orc_w_avg(int bd, uint8_t *dst, ptrdiff_t dst_stride, const int16_t *src0, const int16_t *src1, int width, int height,
    int denom, int w0, int w1, int o0, int o1)
{
    const int wide = bd > 8, shift = denom + orc_max(3, 15 - bd);
    const int off = (((o0 + o1) << (bd - 8)) + 1) << (shift - 1);
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++)
            orc_st(dst + y * dst_stride, x,
                   orc_clip_px((src0[y * ORC_PB + x] * w0 + src1[y * ORC_PB + x] * w1 + off) >> shift, bd), wide);
}

/* vvc_inter_template.c:60 — dst holds the intra prediction; no clip, result truncated to the pixel type */
ORC_API void orc_put_ciip(int bd, uint8_t *dst, ptrdiff_t dst_stride, int width, int height,
    const uint8_t *inter, ptrdiff_t inter_stride, int intra_weight)
{
    const int wide = bd > 8, inter_weight = 4 - intra_weight;
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            uint8_t *d = dst + y * dst_stride;
            orc_st(d, x, (orc_ld(d, x, wide) * intra_weight + orc_ld(inter + y * inter_stride, x, wide) * inter_weight + 2) >> 2, wide);
        }
}

/* vvc_inter_template.c:78 */
ORC_API void orc_put_gpm(int bd, uint8_t *dst, ptrdiff_t dst_stride, int width, int height,
    const int16_t *src0, const int16_t *src1, const uint8_t *weights, int step_x, int step_y)
{
    const int wide = bd > 8, shift = orc_max(5, 17 - bd), off = 1 << (shift - 1);
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            const int wgt = weights[(ptrdiff_t)y * step_y + (ptrdiff_t)x * step_x];
            orc_st(dst + y * dst_stride, x,
                   orc_clip_px((src0[y * ORC_PB + x] * wgt + src1[y * ORC_PB + x] * (8 - wgt) + off) >> shift, bd), wide);
        }
}

/* ------------------------------------------------------------------ BDOF / PROF */

/* vvc_inter_template.c:101 — one-sample ring of integer-position samples around the w x h int16 block */
ORC_API void orc_bdof_fetch_samples(int bd, int16_t *dst, const uint8_t *src, ptrdiff_t src_stride, int x_frac, int y_frac,
    int width, int height)
{
    const int wide = bd > 8, shift = 14 - bd;
    const ptrdiff_t ss = src_stride >> wide;
    const int x_off = (x_frac >> 3) - 1, y_off = (y_frac >> 3) - 1;
    for (int y = -1; y <= height; y++)
        for (int x = -1; x <= width; x++) {
            if (y >= 0 && y < height && x >= 0 && x < width)
                continue;                       /* interior untouched */
            /* ring position (x, y) reads the pixel at (x + 1 + x_off, y + 1 + y_off) */
            dst[y * ORC_PB + x] = (int16_t)(orc_ld(src, (ptrdiff_t)(y + 1 + y_off) * ss + (x + 1 + x_off), wide) << shift);
        }
}

/* vvc_inter_template.c:130 */
ORC_API void orc_fetch_samples(int bd, int16_t *dst, const uint8_t *src, ptrdiff_t src_stride, int x_frac, int y_frac)
{
    orc_bdof_fetch_samples(bd, dst, src, src_stride, x_frac, y_frac, 4, 4);
}

/* vvcdsp.c:29 — replicate the w x h block outward by one sample on every side (corners from the padded rows) */
static void pad_ring_i16(int16_t *blk, ptrdiff_t stride, int w, int h)
{
    for (int y = 0; y < h; y++) {
        blk[y * stride - 1] = blk[y * stride];
        blk[y * stride + w] = blk[y * stride + w - 1];
    }
    for (int x = -1; x <= w; x++) {
        blk[-stride + x] = blk[x];
        blk[h * stride + x] = blk[(h - 1) * stride + x];
    }
}

/* vvc_inter_template.c:135 — central differences of (sample >> 6); bd-independent */
ORC_API void orc_prof_grad_filter(int bd, int16_t *gradient_h, int16_t *gradient_v, ptrdiff_t gradient_stride,
    const int16_t *src, ptrdiff_t src_stride, int width, int height, int pad)
{
    (void)bd;
    int16_t *gh = gradient_h + pad * (1 + gradient_stride);
    int16_t *gv = gradient_v + pad * (1 + gradient_stride);
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            const int16_t *p = src + y * src_stride + x;
            gh[y * gradient_stride + x] = (int16_t)((p[1] >> 6) - (p[-1] >> 6));
            gv[y * gradient_stride + x] = (int16_t)((p[src_stride] >> 6) - (p[-src_stride] >> 6));
        }
    if (pad) {
        pad_ring_i16(gradient_h + 1 + gradient_stride, gradient_stride, width, height);
        pad_ring_i16(gradient_v + 1 + gradient_stride, gradient_stride, width, height);
    }
}

/* shared 4x4 PROF refinement: vvc_inter_template.c:160,181,210.  mode 0 -> int16, 1 -> uni, 2 -> uni_w */
static void prof_4x4(int bd, int mode, int16_t *dst16, uint8_t *dst, ptrdiff_t dst_stride, const int16_t *src,
    const int16_t *dmx, const int16_t *dmy, int denom, int wx, int ox_in)
{
    const int wide = bd > 8;
    const int limit = 1 << orc_max(13, bd + 1);
    const int sh_uni = 14 - bd, off_uni = 1 << (sh_uni - 1);
    const int sh_w = denom + orc_max(2, 14 - bd), off_w = 1 << (sh_w - 1);
    const int ox = ox_in * (1 << (bd - 8));
    int16_t gh[16], gv[16];
    orc_prof_grad_filter(bd, gh, gv, 4, src, ORC_PB, 4, 4, 0);
    for (int y = 0; y < 4; y++)
        for (int x = 0; x < 4; x++) {
            const int o = y * 4 + x;
            const int di = gh[o] * dmx[o] + gv[o] * dmy[o];
            const int val = src[y * ORC_PB + x] + orc_clip3(di, -limit, limit - 1);
            if (mode == 0)
                dst16[y * ORC_PB + x] = (int16_t)val;
            else if (mode == 1)
                orc_st(dst + y * dst_stride, x, orc_clip_px((val + off_uni) >> sh_uni, bd), wide);
            else
                orc_st(dst + y * dst_stride, x, orc_clip_px(((val * wx + off_w) >> sh_w) + ox, bd), wide);
        }
}

ORC_API void orc_apply_prof(int bd, int16_t *dst, const int16_t *src, const int16_t *diff_mv_x, const int16_t *diff_mv_y)
{
    prof_4x4(bd, 0, dst, NULL, 0, src, diff_mv_x, diff_mv_y, 0, 0, 0);
}

ORC_API void orc_apply_prof_uni(int bd, uint8_t *dst, ptrdiff_t dst_stride, const int16_t *src,
    const int16_t *diff_mv_x, const int16_t *diff_mv_y)
{
    prof_4x4(bd, 1, NULL, dst, dst_stride, src, diff_mv_x, diff_mv_y, 0, 0, 0);
}

ORC_API void orc_apply_prof_uni_w(int bd, uint8_t *dst, ptrdiff_t dst_stride, const int16_t *src,
    const int16_t *diff_mv_x, const int16_t *diff_mv_y, int denom, int wx, int ox)
{
    prof_4x4(bd, 2, NULL, dst, dst_stride, src, diff_mv_x, diff_mv_y, denom, wx, ox);
}

/*
 * vvc_inter_template.c:288 — BDOF on a block of at most 16x16.  Gradients live in an 18x18 plane whose (1,1)
 * is the block origin; src0/src1 are padded IN PLACE by one replicated ring (part of the slot's contract).
 */
#define BDOF_GS 18
ORC_API void orc_apply_bdof(int bd, uint8_t *dst, ptrdiff_t dst_stride, int16_t *src0, int16_t *src1, int block_w, int block_h)
{
    const int wide = bd > 8;
    const int sh = 15 - bd, off = 1 << (sh - 1);
    int16_t gh[2][BDOF_GS * BDOF_GS], gv[2][BDOF_GS * BDOF_GS];

    orc_prof_grad_filter(bd, gh[0], gv[0], BDOF_GS, src0, ORC_PB, block_w, block_h, 1);
    pad_ring_i16(src0, ORC_PB, block_w, block_h);
    orc_prof_grad_filter(bd, gh[1], gv[1], BDOF_GS, src1, ORC_PB, block_w, block_h, 1);
    pad_ring_i16(src1, ORC_PB, block_w, block_h);

    for (int by = 0; by < block_h; by += 4)
        for (int bx = 0; bx < block_w; bx += 4) {
            /* 6x6 window: samples (bx-1..bx+4, by-1..by+4); gradient plane index (by + j, bx + i) (:237-265) */
            int sgx2 = 0, sgy2 = 0, sgxgy = 0, sgxdi = 0, sgydi = 0;
            for (int j = 0; j < 6; j++)
                for (int i = 0; i < 6; i++) {
                    const int so = (by - 1 + j) * ORC_PB + (bx - 1 + i);
                    const int go = (by + j) * BDOF_GS + (bx + i);
                    const int diff = (src0[so] >> 4) - (src1[so] >> 4);
                    const int th = (gh[0][go] + gh[1][go]) >> 1;
                    const int tv = (gv[0][go] + gv[1][go]) >> 1;
                    sgx2 += orc_abs(th);
                    sgy2 += orc_abs(tv);
                    sgxgy += orc_sign(tv) * th;
                    sgxdi += -orc_sign(th) * diff;
                    sgydi += -orc_sign(tv) * diff;
                }
            const int vx = sgx2 > 0 ? orc_clip3((sgxdi * 4) >> orc_log2(sgx2), -15, 15) : 0;
            const int vy = sgy2 > 0 ? orc_clip3(((sgydi * 4) - ((vx * sgxgy) >> 1)) >> orc_log2(sgy2), -15, 15) : 0;
            for (int y = 0; y < 4; y++)
                for (int x = 0; x < 4; x++) {
                    const int so = (by + y) * ORC_PB + bx + x;
                    const int go = (by + 1 + y) * BDOF_GS + (bx + 1 + x);
                    const int corr = vx * (gh[0][go] - gh[1][go]) + vy * (gv[0][go] - gv[1][go]);
                    orc_st(dst + (by + y) * dst_stride, bx + x, orc_clip_px((src0[so] + off + src1[so] + corr) >> sh, bd), wide);
                }
        }
}

/* ------------------------------------------------------------------ DMVR */

/* vvcdsp.c:49 — SAD over every other row of two (w+4)x(h+4) bilinear planes displaced by ±(dx-2, dy-2) */
ORC_API int orc_sad(const int16_t *src0, const int16_t *src1, int dx, int dy, int block_w, int block_h)
{
    int sad = 0;
    dx -= 2;
    dy -= 2;
    src0 += (2 + dy) * ORC_PB + 2 + dx;
    src1 += (2 - dy) * ORC_PB + 2 - dx;
    for (int y = 0; y < block_h; y += 2)
        for (int x = 0; x < block_w; x++)
            sad += orc_abs(src0[y * ORC_PB + x] - src1[y * ORC_PB + x]);
    return sad;
}

/* ff_vvc_inter_luma_dmvr_filters (vvc_data.c:1906): bilinear {16 - f, f}, f = 0..15 */
ORC_INLINE int bil(int a, int b, int f) { return (16 - f) * a + f * b; }

/* vvc_inter_template.c:324-413 — bilinear fetch to 10-bit precision */
ORC_API void orc_dmvr(int bd, int vfrac, int hfrac, int16_t *dst, const uint8_t *src, ptrdiff_t src_stride, int height,
    intptr_t mx, intptr_t my, int width)
{
    const int wide = bd > 8;
    const ptrdiff_t ss = src_stride >> wide;
    const int sh1 = bd - 6, off1 = 1 << (sh1 - 1);
    int16_t tmp[(ORC_PB + 1) * ORC_PB];

    if (vfrac && hfrac)
        for (int y = 0; y < height + 1; y++)
            for (int x = 0; x < width; x++) {
                const ptrdiff_t o = (ptrdiff_t)y * ss + x;
                tmp[y * ORC_PB + x] = (int16_t)((bil(orc_ld(src, o, wide), orc_ld(src, o + 1, wide), (int)mx) + off1) >> sh1);
            }
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            const ptrdiff_t o = (ptrdiff_t)y * ss + x;
            int v;
            if (vfrac && hfrac)
                v = (bil(tmp[y * ORC_PB + x], tmp[(y + 1) * ORC_PB + x], (int)my) + 8) >> 4;
            else if (hfrac)
                v = (bil(orc_ld(src, o, wide), orc_ld(src, o + 1, wide), (int)mx) + off1) >> sh1;
            else if (vfrac)
                v = (bil(orc_ld(src, o, wide), orc_ld(src, o + ss, wide), (int)my) + off1) >> sh1;
            else if (bd > 10)
                v = (orc_ld(src, o, wide) + (1 << (bd - 11))) >> (bd - 10);
            else
                v = orc_ld(src, o, wide) << (10 - bd);
            dst[y * ORC_PB + x] = (int16_t)v;
        }
}


/* ------------------------------------------------------------------ callers: one regular bi-predicted sub-block
 *
 * What pred_regular_blk (vvc_inter.c:772-822) does per sub-block around the slots above: derive_sb_mv -> dmvr_mv_refine
 * (:685-748, parametric_mv_refine :642-681), then luma_mc_bi (:253-296) and chroma_mc_bi (:330-369) with their edge
 * emulation (:33-110; libavcodec/videodsp_template.c:28 replicates the nearest sample of the allowed rectangle, i.e. it
 * reads the plane at clamped coordinates).  Flattened: planes, position, motion and flags arrive in orc_bipred_job.
 */
extern const int8_t orc_tab_inter_luma_filters[3 * 16 * 8], orc_tab_inter_chroma_filters[3 * 32 * 4];

#define EMU_STRIDE (ORC_PB + 32)                 /* EDGE_EMU_BUFFER_STRIDE, vvc_ctu.h */

/* emulated_edge_mc restated: window of bw x bh samples whose top-left is plane sample (x0, y0), every coordinate clamped to
 * [cx0, cx1] x [cy0, cy1]; buf is pixel-typed with EMU_STRIDE samples per row */
static void emu_window(int wide, uint8_t *buf, const uint8_t *plane, ptrdiff_t stride_bytes, int x0, int y0, int bw, int bh,
                       int cx0, int cy0, int cx1, int cy1)
{
    const ptrdiff_t ss = stride_bytes >> wide;
    for (int y = 0; y < bh; y++)
        for (int x = 0; x < bw; x++) {
            const int sx = orc_clip3(x0 + x, cx0, cx1), sy = orc_clip3(y0 + y, cy0, cy1);
            orc_st(buf, (ptrdiff_t)y * EMU_STRIDE + x, orc_ld(plane, (ptrdiff_t)sy * ss + sx, wide), wide);
        }
}

/* vvc_inter.c:642-681 */
static int parametric_mv_refine(const int *sad, int stride)
{
    const int sad_minus = sad[-stride], sad_center = sad[0], sad_plus = sad[stride];
    int denom = ((sad_minus + sad_plus) - (sad_center << 1)) << 3;
    if (!denom)
        return 0;
    if (sad_minus == sad_center)
        return -8;
    if (sad_plus == sad_center)
        return 8;
    int num = (sad_minus - sad_plus) * (1 << 4), sign_num = 0, quotient = 0;
    if (num < 0) {
        num = -num;
        sign_num = 1;
    }
    for (int counter = 3; counter > 0; counter--) {
        quotient <<= 1;
        if (num >= denom) {
            num -= denom;
            quotient++;
        }
        denom >>= 1;
    }
    return sign_num ? -quotient : quotient;
}

/* lmcs.filter (vvc_filter_template.c:25) on the luma block a job has just predicted: predict_inter does it per coding unit after every
 * sub-block (vvc_inter.c:888-891), pred_regular_luma on the inter part of a CIIP block (:573-574); a per-sample map, so per block is the same */
static void lmcs_block(int bd, uint64_t lut, uint64_t dst, ptrdiff_t stride, int w, int h)
{
    if (lut)
        orc_lmcs_filter(bd, (uint8_t *)(uintptr_t)dst, stride, w, h, (const uint8_t *)(uintptr_t)lut);
}


ORC_API void orc_bipred_block(int bd, const orc_bipred_job *job)
{
    const int wide = bd > 8;
    const int w = job->w, h = job->h, chroma = job->chroma;
    const uint8_t *ref[2] = { (const uint8_t *)(uintptr_t)job->ref0, (const uint8_t *)(uintptr_t)job->ref1 };
    const ptrdiff_t rstride[2] = { job->ref0_stride, job->ref1_stride };
    orc_bipred_result *rec = (orc_bipred_result *)(uintptr_t)job->rec;
    int mv[4] = { job->mv[0], job->mv[1], job->mv[2], job->mv[3] };
    int bdof = !chroma && job->bdof && !(job->pred_flag == 1 || job->pred_flag == 2);
    static _Thread_local int16_t tmpbuf[2][(ORC_PB + 4) * ORC_PB];
    static _Thread_local uint8_t emu[2 * EMU_STRIDE * (ORC_PB + 8)];

    if (chroma && rec) {
        /* chroma follows the luma block's refined motion (vvc_inter.c:622-628: mv is the one derive_sb_mv refined) */
        for (int k = 0; k < 4; k++) mv[k] = rec->mv[k];
    }
    if (!chroma && job->dmvr && !(job->pred_flag == 1 || job->pred_flag == 2)) {
        /* dmvr_mv_refine, vvc_inter.c:685-748 */
        int sad[5][5], min_dx = 2, min_dy = 2, min_sad, searched = 0;
        for (int i = 0; i < 2; i++) {
            const int pred_w = w + 4, pred_h = h + 4;
            const int mx = mv[2 * i] & 15, my = mv[2 * i + 1] & 15;
            const int ox = job->x + (mv[2 * i] >> 4) - 2, oy = job->y + (mv[2 * i + 1] >> 4) - 2;
            emu_window(wide, emu, ref[i], rstride[i], ox, oy, pred_w + 1, pred_h + 1, 0, 0, job->pic_w - 1, job->pic_h - 1);
            orc_dmvr(bd, !!my, !!mx, tmpbuf[i], emu, (ptrdiff_t)EMU_STRIDE << wide, pred_h, mx, my, pred_w);
        }
        min_sad = orc_sad(tmpbuf[0], tmpbuf[1], 2, 2, w, h);
        min_sad -= min_sad >> 2;
        sad[2][2] = min_sad;
        if (min_sad >= w * h) {
            int dmv[2];
            searched = 1;
            for (int dy = 0; dy < 5; dy++)
                for (int dx = 0; dx < 5; dx++)
                    if (dx != 2 || dy != 2) {
                        sad[dy][dx] = orc_sad(tmpbuf[0], tmpbuf[1], dx, dy, w, h);
                        if (sad[dy][dx] < min_sad) {
                            min_sad = sad[dy][dx];
                            min_dx = dx;
                            min_dy = dy;
                        }
                    }
            dmv[0] = (min_dx - 2) * 16;
            dmv[1] = (min_dy - 2) * 16;
            if (min_dx != 0 && min_dx != 4 && min_dy != 0 && min_dy != 4) {
                dmv[0] += parametric_mv_refine(&sad[min_dy][min_dx], 1);
                dmv[1] += parametric_mv_refine(&sad[min_dy][min_dx], 5);
            }
            for (int i = 0; i < 2; i++) {
                mv[2 * i] = orc_clip3(mv[2 * i] + (1 - 2 * i) * dmv[0], -(1 << 17), (1 << 17) - 1);           /* ff_vvc_clip_mv */
                mv[2 * i + 1] = orc_clip3(mv[2 * i + 1] + (1 - 2 * i) * dmv[1], -(1 << 17), (1 << 17) - 1);
            }
        }
        if (min_sad < 2 * w * h)
            bdof = 0;
        if (rec) {
            rec->min_sad = min_sad;
            rec->searched = searched;
        }
    }
    if (!chroma && rec) {
        for (int k = 0; k < 4; k++) rec->mv[k] = mv[k];
        rec->bdof = bdof;
    }

    /* luma_mc_bi :253-296 / chroma_mc_bi :330-369; uni-prediction: luma_mc_uni :222-251 / chroma_mc_uni :298-328 */
    const int uni = job->pred_flag == 1 || job->pred_flag == 2;
    const int before = chroma ? 1 : 3, after = chroma ? 2 : 4, extra = before + after;
    const int shx = 4 + (chroma ? job->hs : 0), shy = 4 + (chroma ? job->vs : 0);
    int16_t *tmp[2] = { tmpbuf[0] + 2 * ORC_PB + 32, tmpbuf[1] + 2 * ORC_PB + 32 };      /* room for the BDOF ring */
    for (int i = 0; i < 2; i++) {
        if (uni && i != job->pred_flag - 1)
            continue;
        const int mvx = mv[2 * i], mvy = mv[2 * i + 1];
        const int mx = chroma ? (mvx & ((1 << shx) - 1)) << (1 - job->hs) : mvx & 15;
        const int my = chroma ? (mvy & ((1 << shy) - 1)) << (1 - job->vs) : mvy & 15;
        const int ox = job->x + (mvx >> shx), oy = job->y + (mvy >> shy);
        const int8_t *hf = chroma ? orc_tab_inter_chroma_filters + (job->hf_idx * 32 + mx) * 4 : orc_tab_inter_luma_filters + (job->hf_idx * 16 + mx) * 8;
        const int8_t *vf = chroma ? orc_tab_inter_chroma_filters + (job->vf_idx * 32 + my) * 4 : orc_tab_inter_luma_filters + (job->vf_idx * 16 + my) * 8;
        int cx0 = 0, cy0 = 0, cx1 = job->pic_w - 1, cy1 = job->pic_h - 1;
        if (job->dmvr && !uni) {
            /* emulated_edge_dmvr :61-88: the readable rectangle is the window of the UNREFINED block */
            const int x_sb = job->x + (job->mv[2 * i] >> shx), y_sb = job->y + (job->mv[2 * i + 1] >> shy);
            cx0 = orc_min(orc_max(x_sb - before, 0), job->pic_w - 1);
            cy0 = orc_min(orc_max(y_sb - before, 0), job->pic_h - 1);
            cx1 = cx0 + orc_max(orc_min(job->pic_w, x_sb + w + after) - cx0, 1) - 1;
            cy1 = cy0 + orc_max(orc_min(job->pic_h, y_sb + h + after) - cy0, 1) - 1;
        }
        uint8_t *buf = emu + (size_t)i * EMU_STRIDE * (ORC_PB + 8);
        emu_window(wide, buf, ref[i], rstride[i], ox - before, oy - before, w + extra, h + extra, cx0, cy0, cx1, cy1);
        const uint8_t *src = buf + (((ptrdiff_t)before * EMU_STRIDE + before) << wide);
        if (uni) {
            uint8_t *udst = (uint8_t *)(uintptr_t)job->dst;
            if (job->weight_flag)
                orc_put_uni_w(bd, chroma, !!my, !!mx, udst, job->dst_stride, src, (ptrdiff_t)EMU_STRIDE << wide, h, job->denom, job->w0, job->o0, hf, vf, w);
            else
                orc_put_uni(bd, chroma, !!my, !!mx, udst, job->dst_stride, src, (ptrdiff_t)EMU_STRIDE << wide, h, hf, vf, w);
            if (!chroma)
                lmcs_block(bd, job->lmcs_lut, job->dst, job->dst_stride, w, h);
            return;
        }
        orc_put(bd, chroma, !!my, !!mx, tmp[i], src, (ptrdiff_t)EMU_STRIDE << wide, h, hf, vf, w);
        if (bdof)
            orc_bdof_fetch_samples(bd, tmp[i], src, (ptrdiff_t)EMU_STRIDE << wide, mx, my, w, h);
    }
    uint8_t *dst = (uint8_t *)(uintptr_t)job->dst;
    if (bdof)
        orc_apply_bdof(bd, dst, job->dst_stride, tmp[0], tmp[1], w, h);
    else if (job->weight_flag)
        orc_w_avg(bd, dst, job->dst_stride, tmp[0], tmp[1], w, h, job->denom, job->w0, job->w1, job->o0, job->o1);
    else
        orc_avg(bd, dst, job->dst_stride, tmp[0], tmp[1], w, h);
    if (!chroma)
        lmcs_block(bd, job->lmcs_lut, job->dst, job->dst_stride, w, h);
}


/* ------------------------------------------------------------------ callers: one (tile of a) geometric-partition coding unit
 *
 * pred_gpm_blk (vvc_inter.c:466-527) for one component: luma_mc / chroma_mc of each part into an int16 plane (put[..] at the part's
 * motion, edge emulation to the picture), then put_gpm with the mask weights the job addresses.
 */
ORC_API void orc_gpm_block(int bd, const orc_gpm_job *g)
{
    const orc_bipred_job *job = &g->base;
    const int wide = bd > 8;
    const int w = job->w, h = job->h, chroma = job->chroma;
    const uint8_t *ref[2] = { (const uint8_t *)(uintptr_t)job->ref0, (const uint8_t *)(uintptr_t)job->ref1 };
    const ptrdiff_t rstride[2] = { job->ref0_stride, job->ref1_stride };
    static _Thread_local int16_t tmpbuf[2][(ORC_PB + 4) * ORC_PB];
    static _Thread_local uint8_t emu[EMU_STRIDE * (ORC_PB + 8) * 2];
    const int before = chroma ? 1 : 3, after = chroma ? 2 : 4, extra = before + after;
    const int shx = 4 + (chroma ? job->hs : 0), shy = 4 + (chroma ? job->vs : 0);
    for (int i = 0; i < 2; i++) {
        const int mvx = job->mv[2 * i], mvy = job->mv[2 * i + 1];
        const int mx = chroma ? (mvx & ((1 << shx) - 1)) << (1 - job->hs) : mvx & 15;
        const int my = chroma ? (mvy & ((1 << shy) - 1)) << (1 - job->vs) : mvy & 15;
        const int ox = job->x + (mvx >> shx), oy = job->y + (mvy >> shy);
        const int8_t *hf = chroma ? orc_tab_inter_chroma_filters + (job->hf_idx * 32 + mx) * 4 : orc_tab_inter_luma_filters + (job->hf_idx * 16 + mx) * 8;
        const int8_t *vf = chroma ? orc_tab_inter_chroma_filters + (job->vf_idx * 32 + my) * 4 : orc_tab_inter_luma_filters + (job->vf_idx * 16 + my) * 8;
        emu_window(wide, emu, ref[i], rstride[i], ox - before, oy - before, w + extra, h + extra, 0, 0, job->pic_w - 1, job->pic_h - 1);
        const uint8_t *src = emu + (((ptrdiff_t)before * EMU_STRIDE + before) << wide);
        orc_put(bd, chroma, !!my, !!mx, tmpbuf[i], src, (ptrdiff_t)EMU_STRIDE << wide, h, hf, vf, w);
    }
    orc_put_gpm(bd, (uint8_t *)(uintptr_t)job->dst, job->dst_stride, w, h, tmpbuf[0], tmpbuf[1], (const uint8_t *)(uintptr_t)g->weights, g->step_x, g->step_y);
    if (!chroma)
        lmcs_block(bd, job->lmcs_lut, job->dst, job->dst_stride, w, h);
}

/* ------------------------------------------------------------------ callers: one 4x4 luma sub-block of an affine CU
 *
 * luma_prof_uni (vvc_inter.c:369-406) and luma_prof_bi (:408-447), as pred_affine_blk (:864-897) calls them per sub-block:
 * the affine filter set (ff_vvc_inter_luma_filters[2]), edge emulation to the picture, and — where cb_prof_flag is set —
 * put into a temporary, fetch_samples, apply_prof*.  diff_mv = int16 [list][x | y][16] (pu->diff_mv_x / diff_mv_y).
 */
ORC_API void orc_affine_block(int bd, const orc_affine_job *job)
{
    const int wide = bd > 8;
    const uint8_t *ref[2] = { (const uint8_t *)(uintptr_t)job->ref0, (const uint8_t *)(uintptr_t)job->ref1 };
    const ptrdiff_t rstride[2] = { job->ref0_stride, job->ref1_stride };
    const int16_t *dmv = (const int16_t *)(uintptr_t)job->diff_mv;
    const int prof[2] = { job->prof0, job->prof1 };
    uint8_t *dst = (uint8_t *)(uintptr_t)job->dst;
    static _Thread_local int16_t tmpbuf[3][(8 + 4) * ORC_PB];
    static _Thread_local uint8_t emu[EMU_STRIDE * 16 * 2];
    int16_t *tmp[2] = { tmpbuf[0] + 2 * ORC_PB + 32, tmpbuf[1] + 2 * ORC_PB + 32 };
    int16_t *prof_tmp = tmpbuf[2] + 2 * ORC_PB + 32;                 /* lc->tmp(2) + PROF_TEMP_OFFSET: room for the ring */
    const int bi = job->pred_flag == 3;

    for (int i = 0; i < 2; i++) {
        if (!(job->pred_flag & (1 << i)))
            continue;
        const int mvx = job->mv[2 * i], mvy = job->mv[2 * i + 1];
        const int mx = mvx & 15, my = mvy & 15;
        const int ox = job->x + (mvx >> 4), oy = job->y + (mvy >> 4);
        const int8_t *hf = orc_tab_inter_luma_filters + (2 * 16 + mx) * 8, *vf = orc_tab_inter_luma_filters + (2 * 16 + my) * 8;
        emu_window(wide, emu, ref[i], rstride[i], ox - 3, oy - 3, 4 + 7, 4 + 7, 0, 0, job->pic_w - 1, job->pic_h - 1);
        const uint8_t *src = emu + (((ptrdiff_t)3 * EMU_STRIDE + 3) << wide);
        const ptrdiff_t ss = (ptrdiff_t)EMU_STRIDE << wide;
        const int16_t *dmx = dmv + i * 32, *dmy = dmx + 16;
        if (bi) {
            if (!prof[i]) {
                orc_put(bd, 0, !!my, !!mx, tmp[i], src, ss, 4, hf, vf, 4);
            } else {
                orc_put(bd, 0, !!my, !!mx, prof_tmp, src, ss, 4, hf, vf, 4);
                orc_fetch_samples(bd, prof_tmp, src, ss, mx, my);
                orc_apply_prof(bd, tmp[i], prof_tmp, dmx, dmy);
            }
            continue;
        }
        /* uni-prediction: weights in (w0, o0) */
        if (prof[i]) {
            orc_put(bd, 0, !!my, !!mx, prof_tmp, src, ss, 4, hf, vf, 4);
            orc_fetch_samples(bd, prof_tmp, src, ss, mx, my);
            if (!job->weight_flag)
                orc_apply_prof_uni(bd, dst, job->dst_stride, prof_tmp, dmx, dmy);
            else
                orc_apply_prof_uni_w(bd, dst, job->dst_stride, prof_tmp, dmx, dmy, job->denom, job->w0, job->o0);
        } else if (!job->weight_flag) {
            orc_put_uni(bd, 0, !!my, !!mx, dst, job->dst_stride, src, ss, 4, hf, vf, 4);
        } else {
            orc_put_uni_w(bd, 0, !!my, !!mx, dst, job->dst_stride, src, ss, 4, job->denom, job->w0, job->o0, hf, vf, 4);
        }
    }
    if (bi) {
        if (job->weight_flag)
            orc_w_avg(bd, dst, job->dst_stride, tmp[0], tmp[1], 4, 4, job->denom, job->w0, job->w1, job->o0, job->o1);
        else
            orc_avg(bd, dst, job->dst_stride, tmp[0], tmp[1], 4, 4);
    }
    lmcs_block(bd, job->lmcs_lut, job->dst, job->dst_stride, 4, 4);
}

/* ---- pred_regular_blk over a list of coding units (vvc_inter.c:783-813): per unit the sub-block walk, the sub-block's MvField
 * (ff_vvc_get_mvf), the reference pictures of its ref_idx, hpel_if_idx as the luma filter set (pred_regular_luma's hf_idx / vf_idx),
 * derive_weight / derive_weight_uni (:129-177), one job per component; sub-blocks larger than 16x16 (neither DMVR nor BDOF there) in
 * 16x16 tiles, which changes nothing for plain interpolation + averaging.  Then each job through orc_bipred_block, luma first. */
ORC_API void orc_inter_frame_build(const orc_inter_frame *f)
{
    static const int bcw_w_lut[5] = { 4, 5, 3, 10, -2 };        /* vvc_inter.c:29 */
    const orc_inter_pu *pus = (const orc_inter_pu *)(uintptr_t)f->pus;
    const orc_inter_slice *slices = (const orc_inter_slice *)(uintptr_t)f->slices;
    const orc_ref_pic *refs = (const orc_ref_pic *)(uintptr_t)f->refs;
    const orc_mv_field *mvf_tab = (const orc_mv_field *)(uintptr_t)f->mvf;
    orc_bipred_job *jl = (orc_bipred_job *)(uintptr_t)f->jobs_luma, *jc = (orc_bipred_job *)(uintptr_t)f->jobs_chroma;
    orc_bipred_result *rec = (orc_bipred_result *)(uintptr_t)f->records;
    for (int u = 0; u < f->n_pus; u++) {
        const orc_inter_pu *pu = pus + u;
        const orc_inter_slice *sl = slices + pu->slice;
        const int sbw = pu->cb_width / pu->num_sb_x, sbh = pu->cb_height / pu->num_sb_y;
        const int tw = sbw < 16 ? sbw : 16, th = sbh < 16 ? sbh : 16;
        uint32_t job = pu->first_job;
        for (int sby = 0; sby < pu->num_sb_y; sby++)
            for (int sbx = 0; sbx < pu->num_sb_x; sbx++) {
                const int sx = pu->x0 + sbx * sbw, sy = pu->y0 + sby * sbh;
                const orc_mv_field *mv = mvf_tab + (sy >> 2) * f->mvf_stride + (sx >> 2);
                const int bi = mv->pred_flag == 3;
                for (int ty = 0; ty < sbh; ty += th)
                    for (int tx = 0; tx < sbw; tx += tw, job++)
                        for (int c = 0; c < (f->chroma_format_idc ? 3 : 1); c++) {
                            const int hs = c ? f->hs : 0, vs = c ? f->vs : 0;
                            const int x = (sx + tx) >> hs, y = (sy + ty) >> vs;
                            orc_bipred_job j;
                            memset(&j, 0, sizeof(j));
                            j.dst = f->dst[c] + (uint64_t)y * f->dst_stride[c] + ((uint64_t)x << f->pixel_shift);
                            j.dst_stride = f->dst_stride[c];
                            for (int l = 0; l < 2; l++) {
                                if (!(mv->pred_flag & (1 << l)))
                                    continue;
                                const orc_ref_pic *rp = refs + l * 16 + mv->ref_idx[l];
                                if (l) { j.ref1 = rp->plane[c]; j.ref1_stride = rp->stride[c]; }
                                else   { j.ref0 = rp->plane[c]; j.ref0_stride = rp->stride[c]; }
                                j.mv[2 * l] = mv->mv[l][0];
                                j.mv[2 * l + 1] = mv->mv[l][1];
                            }
                            j.rec = (uint64_t)(uintptr_t)(rec + job);
                            j.x = (int16_t)x; j.y = (int16_t)y; j.w = (int16_t)(tw >> hs); j.h = (int16_t)(th >> vs);
                            j.pic_w = (int16_t)(f->width >> hs); j.pic_h = (int16_t)(f->height >> vs);
                            j.chroma = c > 0; j.hs = f->hs; j.vs = f->vs;
                            j.dmvr = bi && pu->dmvr_flag;
                            j.bdof = !c && bi && pu->bdof_flag;
                            j.hf_idx = j.vf_idx = c ? 0 : pu->hpel_if_idx;
                            j.pred_flag = mv->pred_flag;
                            /* predict_inter's lmcs.filter (vvc_inter.c:888-891): luma of the coding unit, not for CIIP */
                            j.lmcs_lut = (!c && sl->lmcs_used && !pu->ciip_flag) ? f->lmcs_fwd_lut : 0;
                            if (bi) {
                                const int weight_flag = sl->weighted_pred || (sl->weighted_bipred && !pu->dmvr_flag);
                                if ((weight_flag || mv->bcw_idx) && !(mv->bcw_idx && pu->ciip_flag)) {
                                    j.weight_flag = 1;
                                    if (mv->bcw_idx) {
                                        j.denom = 2; j.w1 = (int16_t)bcw_w_lut[mv->bcw_idx]; j.w0 = (int16_t)(8 - j.w1);
                                    } else {
                                        j.denom = sl->log2_denom[c > 0];
                                        j.w0 = sl->weight[0][c][mv->ref_idx[0]]; j.w1 = sl->weight[1][c][mv->ref_idx[1]];
                                        j.o0 = sl->offset[0][c][mv->ref_idx[0]]; j.o1 = sl->offset[1][c][mv->ref_idx[1]];
                                    }
                                }
                            } else if (sl->weighted_pred || sl->weighted_bipred) {
                                const int lx = mv->pred_flag - 1;
                                j.weight_flag = 1;
                                j.denom = sl->log2_denom[c > 0];
                                j.w0 = sl->weight[lx][c][mv->ref_idx[lx]];
                                j.o0 = sl->offset[lx][c][mv->ref_idx[lx]];
                            }
                            if (c == 0) jl[job] = j; else jc[2 * job + c - 1] = j;
                        }
            }
    }
}

ORC_API void orc_inter_frame_pass(int bd, const orc_inter_frame *f)
{
    orc_inter_frame_build(f);
    const orc_bipred_job *jl = (const orc_bipred_job *)(uintptr_t)f->jobs_luma, *jc = (const orc_bipred_job *)(uintptr_t)f->jobs_chroma;
    for (int i = 0; i < f->n_jobs; i++)
        orc_bipred_block(bd, jl + i);
    if (f->dmvr_mvf) {
        /* set_dmvr_info (vvc_inter.c:750-762) */
        const orc_mv_field *src = (const orc_mv_field *)(uintptr_t)f->mvf;
        orc_mv_field *dst = (orc_mv_field *)(uintptr_t)f->dmvr_mvf;
        const orc_bipred_result *rec = (const orc_bipred_result *)(uintptr_t)f->records;
        for (int i = 0; i < f->n_jobs; i++) {
            const orc_bipred_job *j = jl + i;
            if (!j->dmvr)
                continue;
            orc_mv_field m = src[(j->y >> 2) * f->mvf_stride + (j->x >> 2)];
            m.mv[0][0] = rec[i].mv[0]; m.mv[0][1] = rec[i].mv[1]; m.mv[1][0] = rec[i].mv[2]; m.mv[1][1] = rec[i].mv[3];
            for (int y = j->y; y < j->y + j->h; y += 4)
                for (int x = j->x; x < j->x + j->w; x += 4)
                    dst[(y >> 2) * f->mvf_stride + (x >> 2)] = m;
        }
    }
    if (f->chroma_format_idc)
        for (int i = 0; i < 2 * f->n_jobs; i++)
            orc_bipred_block(bd, jc + i);
}
