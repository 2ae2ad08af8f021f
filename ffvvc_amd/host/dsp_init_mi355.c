/*
 * dsp_init_mi355.c — installs the MI355X kernels into the decoder's DSP function-pointer table.
 *
 * This is the C host shim of the backend: the MI355X counterpart of libavcodec/x86/vvc/vvcdsp_init.c:294-361.
 * Every table slot gets a small trampoline that binds the bit depth (the reference binds it by instantiating its
 * templates three times, libavcodec/vvc/vvcdsp.c:216-226) and the table indices, and forwards to the C ABI of
 * libvvc_mi355.so (include/vvc_mi355.h).  Plain C11, no HIP types: it only needs the two headers.
 */
#include <string.h>

#include "vvc_mi355.h"
#include "vvc_mi355_dsp.h"

/* ------------------------------------------------------------------ trampolines, generated per bit depth */

#define PUT3(BD, C, V, H)                                                                                              \
    static void put_##BD##_##C##V##H(int16_t *dst, const uint8_t *src, ptrdiff_t ss, int height,                       \
                                     const int8_t *hf, const int8_t *vf, int width)                                    \
    { vvc355_put(BD, C, V, H, dst, src, ss, height, hf, vf, width); }                                                  \
    static void put_uni_##BD##_##C##V##H(uint8_t *dst, ptrdiff_t ds, const uint8_t *src, ptrdiff_t ss, int height,     \
                                         const int8_t *hf, const int8_t *vf, int width)                                \
    { vvc355_put_uni(BD, C, V, H, dst, ds, src, ss, height, hf, vf, width); }                                          \
    static void put_uni_w_##BD##_##C##V##H(uint8_t *dst, ptrdiff_t ds, const uint8_t *src, ptrdiff_t ss, int height,   \
                                           int denom, int wx, int ox, const int8_t *hf, const int8_t *vf, int width)   \
    { vvc355_put_uni_w(BD, C, V, H, dst, ds, src, ss, height, denom, wx, ox, hf, vf, width); }

#define DMVR1(BD, V, H)                                                                                                \
    static void dmvr_##BD##_##V##H(int16_t *dst, const uint8_t *src, ptrdiff_t ss, int height,                         \
                                   intptr_t mx, intptr_t my, int width)                                                \
    { vvc355_dmvr(BD, V, H, dst, src, ss, height, mx, my, width); }

#define LF1(BD, D)                                                                                                     \
    static int ladf_##BD##_##D(const uint8_t *pix, ptrdiff_t stride) { return vvc355_lf_ladf_level(BD, D, pix, stride); } \
    static void lf_luma_##BD##_##D(uint8_t *pix, ptrdiff_t stride, const int32_t *beta, const int32_t *tc,             \
        const uint8_t *no_p, const uint8_t *no_q, const uint8_t *lp, const uint8_t *lq, int hor_ctu_edge)              \
    { vvc355_lf_filter_luma(BD, D, pix, stride, beta, tc, no_p, no_q, lp, lq, hor_ctu_edge); }                         \
    static void lf_chroma_##BD##_##D(uint8_t *pix, ptrdiff_t stride, const int32_t *beta, const int32_t *tc,           \
        const uint8_t *no_p, const uint8_t *no_q, const uint8_t *lp, const uint8_t *lq, int shift)                     \
    { vvc355_lf_filter_chroma(BD, D, pix, stride, beta, tc, no_p, no_q, lp, lq, shift); }

#define PER_BD(BD)                                                                                                     \
    PUT3(BD, 0, 0, 0) PUT3(BD, 0, 0, 1) PUT3(BD, 0, 1, 0) PUT3(BD, 0, 1, 1)                                            \
    PUT3(BD, 1, 0, 0) PUT3(BD, 1, 0, 1) PUT3(BD, 1, 1, 0) PUT3(BD, 1, 1, 1)                                            \
    DMVR1(BD, 0, 0) DMVR1(BD, 0, 1) DMVR1(BD, 1, 0) DMVR1(BD, 1, 1)                                                    \
    LF1(BD, 0) LF1(BD, 1)                                                                                              \
    static void avg_##BD(uint8_t *d, ptrdiff_t ds, const int16_t *s0, const int16_t *s1, int w, int h)                 \
    { vvc355_avg(BD, d, ds, s0, s1, w, h); }                                                                           \
    static void w_avg_##BD(uint8_t *d, ptrdiff_t ds, const int16_t *s0, const int16_t *s1, int w, int h,               \
                           int denom, int w0, int w1, int o0, int o1)                                                  \
    { vvc355_w_avg(BD, d, ds, s0, s1, w, h, denom, w0, w1, o0, o1); }                                                  \
    static void put_ciip_##BD(uint8_t *d, ptrdiff_t ds, int w, int h, const uint8_t *inter, ptrdiff_t is, int iw)      \
    { vvc355_put_ciip(BD, d, ds, w, h, inter, is, iw); }                                                               \
    static void put_gpm_##BD(uint8_t *d, ptrdiff_t ds, int w, int h, const int16_t *s0, const int16_t *s1,             \
                             const uint8_t *weights, int sx, int sy)                                                   \
    { vvc355_put_gpm(BD, d, ds, w, h, s0, s1, weights, sx, sy); }                                                      \
    static void fetch_samples_##BD(int16_t *d, const uint8_t *s, ptrdiff_t ss, int xf, int yf)                         \
    { vvc355_fetch_samples(BD, d, s, ss, xf, yf); }                                                                    \
    static void bdof_fetch_samples_##BD(int16_t *d, const uint8_t *s, ptrdiff_t ss, int xf, int yf, int w, int h)      \
    { vvc355_bdof_fetch_samples(BD, d, s, ss, xf, yf, w, h); }                                                         \
    static void prof_grad_filter_##BD(int16_t *gh, int16_t *gv, ptrdiff_t gs, const int16_t *s, ptrdiff_t ss,          \
                                      int w, int h, int pad)                                                           \
    { vvc355_prof_grad_filter(BD, gh, gv, gs, s, ss, w, h, pad); }                                                     \
    static void apply_prof_##BD(int16_t *d, const int16_t *s, const int16_t *mx, const int16_t *my)                    \
    { vvc355_apply_prof(BD, d, s, mx, my); }                                                                           \
    static void apply_prof_uni_##BD(uint8_t *d, ptrdiff_t ds, const int16_t *s, const int16_t *mx, const int16_t *my)  \
    { vvc355_apply_prof_uni(BD, d, ds, s, mx, my); }                                                                   \
    static void apply_prof_uni_w_##BD(uint8_t *d, ptrdiff_t ds, const int16_t *s, const int16_t *mx,                   \
                                      const int16_t *my, int denom, int wx, int ox)                                    \
    { vvc355_apply_prof_uni_w(BD, d, ds, s, mx, my, denom, wx, ox); }                                                  \
    static void apply_bdof_##BD(uint8_t *d, ptrdiff_t ds, int16_t *s0, int16_t *s1, int w, int h)                      \
    { vvc355_apply_bdof(BD, d, ds, s0, s1, w, h); }                                                                    \
    static void pred_planar_##BD(uint8_t *s, const uint8_t *t, const uint8_t *l, int w, int h, ptrdiff_t st)           \
    { vvc355_pred_planar(BD, s, t, l, w, h, st); }                                                                     \
    static void pred_mip_##BD(uint8_t *s, const uint8_t *t, const uint8_t *l, int w, int h, ptrdiff_t st, int m, int tr) \
    { vvc355_pred_mip(BD, s, t, l, w, h, st, m, tr); }                                                                 \
    static void pred_dc_##BD(uint8_t *s, const uint8_t *t, const uint8_t *l, int w, int h, ptrdiff_t st)               \
    { vvc355_pred_dc(BD, s, t, l, w, h, st); }                                                                         \
    static void pred_v_##BD(uint8_t *s, const uint8_t *t, int w, int h, ptrdiff_t st) { vvc355_pred_v(BD, s, t, w, h, st); } \
    static void pred_h_##BD(uint8_t *s, const uint8_t *l, int w, int h, ptrdiff_t st) { vvc355_pred_h(BD, s, l, w, h, st); } \
    static void pred_angular_v_##BD(uint8_t *s, const uint8_t *t, const uint8_t *l, int w, int h, ptrdiff_t st,        \
                                    int c_idx, int mode, int ref_idx, int ff, int pdpc)                                \
    { vvc355_pred_angular_v(BD, s, t, l, w, h, st, c_idx, mode, ref_idx, ff, pdpc); }                                  \
    static void pred_angular_h_##BD(uint8_t *s, const uint8_t *t, const uint8_t *l, int w, int h, ptrdiff_t st,        \
                                    int c_idx, int mode, int ref_idx, int ff, int pdpc)                                \
    { vvc355_pred_angular_h(BD, s, t, l, w, h, st, c_idx, mode, ref_idx, ff, pdpc); }                                  \
    static void add_residual_##BD(uint8_t *d, const int *r, int w, int h, ptrdiff_t st)                                \
    { vvc355_add_residual(BD, d, r, w, h, st); }                                                                       \
    static void add_residual_joint_##BD(uint8_t *d, const int *r, int w, int h, ptrdiff_t st, int cs, int sh)          \
    { vvc355_add_residual_joint(BD, d, r, w, h, st, cs, sh); }                                                         \
    static void lmcs_filter_##BD(uint8_t *d, ptrdiff_t ds, int w, int h, const uint8_t *lut)                           \
    { vvc355_lmcs_filter(BD, d, ds, w, h, lut); }                                                                      \
    static void sao_band_##BD(uint8_t *d, const uint8_t *s, ptrdiff_t ds, ptrdiff_t ss, const int16_t *off,            \
                              int left_class, int w, int h)                                                            \
    { vvc355_sao_band_filter(BD, d, s, ds, ss, off, left_class, w, h); }                                               \
    static void sao_edge_##BD(uint8_t *d, const uint8_t *s, ptrdiff_t ds, const int16_t *off, int eo, int w, int h)    \
    { vvc355_sao_edge_filter(BD, d, s, ds, off, eo, w, h); }                                                           \
    static void alf_luma_##BD(uint8_t *d, ptrdiff_t ds, const uint8_t *s, ptrdiff_t ss, int w, int h,                  \
                              const int16_t *f, const int16_t *c, int vb)                                              \
    { vvc355_alf_filter_luma(BD, d, ds, s, ss, w, h, f, c, vb); }                                                      \
    static void alf_chroma_##BD(uint8_t *d, ptrdiff_t ds, const uint8_t *s, ptrdiff_t ss, int w, int h,                \
                                const int16_t *f, const int16_t *c, int vb)                                            \
    { vvc355_alf_filter_chroma(BD, d, ds, s, ss, w, h, f, c, vb); }                                                    \
    static void alf_cc_##BD(uint8_t *d, ptrdiff_t ds, const uint8_t *l, ptrdiff_t ls, int w, int h, int hs, int vs,    \
                            const int16_t *f, int vb)                                                                  \
    { vvc355_alf_filter_cc(BD, d, ds, l, ls, w, h, hs, vs, f, vb); }                                                   \
    static void alf_classify_##BD(int *ci, int *ti, const uint8_t *s, ptrdiff_t ss, int w, int h, int vb, int *g)      \
    { vvc355_alf_classify(BD, ci, ti, s, ss, w, h, vb, g); }                                                           \
    static void alf_recon_##BD(int16_t *co, int16_t *cl, const int *ci, const int *ti, int n, const int16_t *cs,       \
                               const uint8_t *cis, const uint8_t *c2f)                                                 \
    { vvc355_alf_recon_coeff_and_clip(BD, co, cl, ci, ti, n, cs, cis, c2f); }                                          \
    static void install_##BD(VVC355DSPContext *c)                                                                      \
    {                                                                                                                  \
        for (int w = 0; w < 7; w++) {                                                                                  \
            c->inter.put[0][w][0][0] = put_##BD##_000; c->inter.put[0][w][0][1] = put_##BD##_001;                      \
            c->inter.put[0][w][1][0] = put_##BD##_010; c->inter.put[0][w][1][1] = put_##BD##_011;                      \
            c->inter.put[1][w][0][0] = put_##BD##_100; c->inter.put[1][w][0][1] = put_##BD##_101;                      \
            c->inter.put[1][w][1][0] = put_##BD##_110; c->inter.put[1][w][1][1] = put_##BD##_111;                      \
            c->inter.put_uni[0][w][0][0] = put_uni_##BD##_000; c->inter.put_uni[0][w][0][1] = put_uni_##BD##_001;      \
            c->inter.put_uni[0][w][1][0] = put_uni_##BD##_010; c->inter.put_uni[0][w][1][1] = put_uni_##BD##_011;      \
            c->inter.put_uni[1][w][0][0] = put_uni_##BD##_100; c->inter.put_uni[1][w][0][1] = put_uni_##BD##_101;      \
            c->inter.put_uni[1][w][1][0] = put_uni_##BD##_110; c->inter.put_uni[1][w][1][1] = put_uni_##BD##_111;      \
            c->inter.put_uni_w[0][w][0][0] = put_uni_w_##BD##_000; c->inter.put_uni_w[0][w][0][1] = put_uni_w_##BD##_001; \
            c->inter.put_uni_w[0][w][1][0] = put_uni_w_##BD##_010; c->inter.put_uni_w[0][w][1][1] = put_uni_w_##BD##_011; \
            c->inter.put_uni_w[1][w][0][0] = put_uni_w_##BD##_100; c->inter.put_uni_w[1][w][0][1] = put_uni_w_##BD##_101; \
            c->inter.put_uni_w[1][w][1][0] = put_uni_w_##BD##_110; c->inter.put_uni_w[1][w][1][1] = put_uni_w_##BD##_111; \
        }                                                                                                              \
        c->inter.avg = avg_##BD; c->inter.w_avg = w_avg_##BD;                                                          \
        c->inter.put_ciip = put_ciip_##BD; c->inter.put_gpm = put_gpm_##BD;                                            \
        c->inter.fetch_samples = fetch_samples_##BD; c->inter.bdof_fetch_samples = bdof_fetch_samples_##BD;            \
        c->inter.prof_grad_filter = prof_grad_filter_##BD; c->inter.apply_prof = apply_prof_##BD;                      \
        c->inter.apply_prof_uni = apply_prof_uni_##BD; c->inter.apply_prof_uni_w = apply_prof_uni_w_##BD;              \
        c->inter.apply_bdof = apply_bdof_##BD;                                                                         \
        c->inter.dmvr[0][0] = dmvr_##BD##_00; c->inter.dmvr[0][1] = dmvr_##BD##_01;                                    \
        c->inter.dmvr[1][0] = dmvr_##BD##_10; c->inter.dmvr[1][1] = dmvr_##BD##_11;                                    \
        c->intra.pred_planar = pred_planar_##BD; c->intra.pred_mip = pred_mip_##BD; c->intra.pred_dc = pred_dc_##BD;   \
        c->intra.pred_v = pred_v_##BD; c->intra.pred_h = pred_h_##BD;                                                  \
        c->intra.pred_angular_v = pred_angular_v_##BD; c->intra.pred_angular_h = pred_angular_h_##BD;                  \
        c->itx.add_residual = add_residual_##BD; c->itx.add_residual_joint = add_residual_joint_##BD;                  \
        c->lmcs.filter = lmcs_filter_##BD;                                                                             \
        c->lf.ladf_level[0] = ladf_##BD##_0; c->lf.ladf_level[1] = ladf_##BD##_1;                                      \
        c->lf.filter_luma[0] = lf_luma_##BD##_0; c->lf.filter_luma[1] = lf_luma_##BD##_1;                              \
        c->lf.filter_chroma[0] = lf_chroma_##BD##_0; c->lf.filter_chroma[1] = lf_chroma_##BD##_1;                      \
        for (int i = 0; i < 9; i++) { c->sao.band_filter[i] = sao_band_##BD; c->sao.edge_filter[i] = sao_edge_##BD; }  \
        c->alf.filter[0] = alf_luma_##BD; c->alf.filter[1] = alf_chroma_##BD; c->alf.filter_cc = alf_cc_##BD;          \
        c->alf.classify = alf_classify_##BD; c->alf.recon_coeff_and_clip = alf_recon_##BD;                             \
    }

PER_BD(8)
PER_BD(10)
PER_BD(12)

/* ------------------------------------------------------------------ inverse transforms: one trampoline per table entry
 * (the slot signature carries neither the transform types nor the block size; the bit depth is an argument) */

#define ITX1(H, V, LW, LH)                                                                                             \
    static void itx_##H##V##_##LW##x##LH(int *coeffs, size_t nzw, size_t nzh, intptr_t range, intptr_t bd)             \
    { vvc355_itx(H, V, LW, LH, coeffs, nzw, nzh, range, bd); }
#define ITX_LH(H, V, LW) ITX1(H, V, LW, 0) ITX1(H, V, LW, 1) ITX1(H, V, LW, 2) ITX1(H, V, LW, 3) ITX1(H, V, LW, 4) ITX1(H, V, LW, 5) ITX1(H, V, LW, 6)
#define ITX_LW(H, V) ITX_LH(H, V, 0) ITX_LH(H, V, 1) ITX_LH(H, V, 2) ITX_LH(H, V, 3) ITX_LH(H, V, 4) ITX_LH(H, V, 5) ITX_LH(H, V, 6)
ITX_LW(0, 0) ITX_LW(0, 1) ITX_LW(0, 2) ITX_LW(1, 0) ITX_LW(1, 1) ITX_LW(1, 2) ITX_LW(2, 0) ITX_LW(2, 1) ITX_LW(2, 2)

#define ITX_ROW(H, V, LW) { itx_##H##V##_##LW##x0, itx_##H##V##_##LW##x1, itx_##H##V##_##LW##x2, itx_##H##V##_##LW##x3, \
                            itx_##H##V##_##LW##x4, itx_##H##V##_##LW##x5, itx_##H##V##_##LW##x6 }
#define ITX_TAB(H, V) { ITX_ROW(H, V, 0), ITX_ROW(H, V, 1), ITX_ROW(H, V, 2), ITX_ROW(H, V, 3), ITX_ROW(H, V, 4), ITX_ROW(H, V, 5), ITX_ROW(H, V, 6) }
static const vvc355_itx_fn itx_all[3][3][7][7] = {
    { ITX_TAB(0, 0), ITX_TAB(0, 1), ITX_TAB(0, 2) },
    { ITX_TAB(1, 0), ITX_TAB(1, 1), ITX_TAB(1, 2) },
    { ITX_TAB(2, 0), ITX_TAB(2, 1), ITX_TAB(2, 2) },
};

/* the combinations libavcodec/vvc/vvcdsp_template.c:142-159 installs (0 = DCT2, 1 = DST7, 2 = DCT8) */
static int itx_entry_exists(int trh, int trv, int lw, int lh)
{
    if (!lw && !lh)
        return 0;
    if (!lh)
        return trv == 0 && (lw == 4 || lw == 5 || (lw == 6 && trh == 0));
    if (!lw)
        return trh == 0 && (lh == 4 || lh == 5 || (lh == 6 && trv == 0));
    if (trh != 0 && (lw < 2 || lw > 5))
        return 0;
    if (trv != 0 && (lh < 2 || lh > 5))
        return 0;
    return 1;
}

static void pred_residual_joint_any(int *buf, int w, int h, int c_sign, int shift) { vvc355_pred_residual_joint(buf, w, h, c_sign, shift); }
static void transform_bdpcm_any(int *c, int w, int h, int vertical, int range) { vvc355_transform_bdpcm(c, w, h, vertical, range); }
static int  sad_any(const int16_t *s0, const int16_t *s1, int dx, int dy, int w, int h) { return vvc355_sad(s0, s1, dx, dy, w, h); }

void ff_vvc_dsp_init_mi355(VVC355DSPContext *c, int bit_depth)
{
    switch (bit_depth) {
    case 12: install_12(c); break;
    case 10: install_10(c); break;
    default: install_8(c);  break;          /* the reference maps every other depth to its 8-bit set (vvcdsp.c:249) */
    }
    c->inter.sad = sad_any;
    c->itx.pred_residual_joint = pred_residual_joint_any;
    c->itx.transform_bdpcm = transform_bdpcm_any;
    for (int h = 0; h < 3; h++)
        for (int v = 0; v < 3; v++)
            for (int lw = 0; lw < 7; lw++)
                for (int lh = 0; lh < 7; lh++)
                    if (itx_entry_exists(h, v, lw, lh))
                        c->itx.itx[h][v][lw][lh] = itx_all[h][v][lw][lh];
    /* untouched here: intra.intra_pred / intra_cclm_pred / lmcs_scale_chroma and sao.edge_restore take decoder structs
     * (VVCLocalContext, SAOParams); the in-tree shim flattens them into vvc355_*_job and calls vvc355_*_flat. */
}

int vvc355_dsp_count_slots(const VVC355DSPContext *c)
{
    void *const *p = (void *const *)c;
    int n = 0;
    for (size_t i = 0; i < sizeof(*c) / sizeof(void *); i++)
        n += p[i] != NULL;
    return n;
}

/* ------------------------------------------------------------------ table self-test (needs a GPU) */

static unsigned lcg(unsigned *s) { *s = *s * 1664525u + 1013904223u; return *s >> 8; }

int vvc355_dsp_table_selftest(int bd)
{
    VVC355DSPContext c;
    memset(&c, 0, sizeof(c));
    ff_vvc_dsp_init_mi355(&c, bd);
    const int px = bd > 8 ? 2 : 1, mask = (1 << bd) - 1;
    static uint8_t plane[80 * 96 * 2], d0[64 * 64 * 2], d1[64 * 64 * 2];
    static int16_t t0[64 * 128], t1[64 * 128], u0[64 * 128], u1[64 * 128];
    static const int8_t hf[8] = { -1, 4, -11, 40, 40, -11, 4, -1 }, vf[8] = { 0, 1, -3, 63, 4, -2, 1, 0 };
    unsigned seed = 12345u + (unsigned)bd;
    for (int i = 0; i < 80 * 96; i++) {
        const int v = (int)lcg(&seed) & mask;
        if (px == 2) ((uint16_t *)plane)[i] = (uint16_t)v; else plane[i] = (uint8_t)v;
    }
    const uint8_t *src = plane + (8 * 96 + 8) * px;
    /* luma hv put of a 32x16 block (table index log2(32) - 1 = 4), twice, then avg — through the table ... */
    c.inter.put[0][4][1][1](t0, src, 96 * px, 16, hf, vf, 32);
    c.inter.put[0][4][1][1](t1, src + 3 * px, 96 * px, 16, hf, vf, 32);
    c.inter.avg(d0, 64 * px, t0, t1, 32, 16);
    /* ... and directly through the C ABI */
    vvc355_put(bd, 0, 1, 1, u0, src, 96 * px, 16, hf, vf, 32);
    vvc355_put(bd, 0, 1, 1, u1, src + 3 * px, 96 * px, 16, hf, vf, 32);
    vvc355_avg(bd, d1, 64 * px, u0, u1, 32, 16);
    for (int y = 0; y < 16; y++)
        if (memcmp(d0 + y * 64 * px, d1 + y * 64 * px, 32 * px) || memcmp(t0 + y * 128, u0 + y * 128, 64))
            return 1;
    /* an inverse transform entry: DST7 x DCT8 16x8 */
    int ca[128], cb[128];
    for (int i = 0; i < 128; i++)
        ca[i] = cb[i] = (i % 16 < 5 && i / 16 < 3) ? (int)(lcg(&seed) & 0xffff) - 32768 : 0;
    if (!c.itx.itx[1][2][4][3] || c.itx.itx[2][0][1][1])
        return 2;
    c.itx.itx[1][2][4][3](ca, 5, 3, 15, bd);
    vvc355_itx(1, 2, 4, 3, cb, 5, 3, 15, bd);
    if (memcmp(ca, cb, sizeof(ca)))
        return 3;
    return 0;
}
