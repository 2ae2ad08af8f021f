/*
 * oracle/orc_itx.c — CPU restatement of the inverse-transform / residual DSP slots.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see orc_common.h).
 *
 * Follows, by reading:
 *   libavcodec/vvc/vvc_itx_1d.c   (DCT-2 butterflies :88-653 and their nz gating :64-67, matrix_mul :657, LFNST :708)
 *   libavcodec/vvc/vvcdsp.c       (scale_clip :67, scale :81, itx_2d :94, itx_1d :119, generator :140-195)
 *   libavcodec/vvc/vvcdsp_template.c (add_residual :32, joint :48,:65, transform_bdpcm :76, table fill :142-159)
 *
 * The reference evaluates DCT-2 with even/odd partial butterflies; all of its arithmetic is wrapping int32 adds and
 * multiplies, so each output equals the plain dot product of the (gated) inputs with the transform matrix column.
 * That dot product is what is written here; the matrix entry is looked up by its cosine angle index.
 */
#include "vvc_oracle.h"
#include "orc_common.h"

extern const int8_t orc_tab_dct2_cos[256];
extern const int8_t orc_tab_dst7_4[16], orc_tab_dst7_8[64], orc_tab_dst7_16[256], orc_tab_dst7_32[1024];
extern const int8_t orc_tab_dct8_4[16], orc_tab_dct8_8[64], orc_tab_dct8_16[256], orc_tab_dct8_32[1024];
extern const int8_t orc_tab_lfnst_8x8[4 * 2 * 16 * 48], orc_tab_lfnst_4x4[4 * 2 * 16 * 16];
extern const uint8_t orc_tab_lfnst_tr_set_index[95];

/* largest power of two <= k (k >= 1) */
static int pow2_floor(int k) { return 1 << orc_log2((unsigned)k); }

/* vvc_itx_1d.c:64-67: input k of a DCT-2 takes part iff it is one of the first two, or nz exceeds the power of two at
 * or below k (terms are gated in groups 2-3, 4-7, 8-15, 16-31); the 64-point transform never reads inputs 32..63 (:498) */
static int dct2_input_used(int n, int k, size_t nz)
{
    if (n == 64 && k >= 32)
        return 0;
    return k < 2 || nz > (size_t)pow2_floor(k);
}

ORC_API void orc_inv_tx_1d(int type, int n, int *c, ptrdiff_t stride, size_t nz)
{
    int in[64], out[64];
    if (n == 1)
        return;                                   /* the *_1 stubs, vvc_itx_1d.c:70-80 */
    if (type == ORC_DCT2) {
        const int step = 64 / n;
        for (int k = 0; k < n; k++)
            in[k] = dct2_input_used(n, k, nz) ? c[k * stride] : 0;
        for (int i = 0; i < n; i++) {
            unsigned acc = 0;
            for (int k = 0; k < n; k++)
                acc += (unsigned)in[k] * (unsigned)(int)orc_tab_dct2_cos[((2 * i + 1) * k * step) & 255];
            out[i] = (int)acc;
        }
    } else {
        const int8_t *m = type == ORC_DST7
            ? (n == 4 ? orc_tab_dst7_4 : n == 8 ? orc_tab_dst7_8 : n == 16 ? orc_tab_dst7_16 : orc_tab_dst7_32)
            : (n == 4 ? orc_tab_dct8_4 : n == 8 ? orc_tab_dct8_8 : n == 16 ? orc_tab_dct8_16 : orc_tab_dct8_32);
        if (nz > 16)
            abort();                              /* matrix_mul keeps 16 inputs (:659-660) */
        for (size_t j = 0; j < nz; j++)
            in[j] = c[j * stride];
        for (int i = 0; i < n; i++) {
            unsigned acc = 0;
            for (size_t j = 0; j < nz; j++)
                acc += (unsigned)in[j] * (unsigned)(int)m[j * n + i];
            out[i] = (int)acc;
        }
    }
    for (int i = 0; i < n; i++)
        c[i * stride] = out[i];
}

/* which (trh, trv, log2 w, log2 h) the reference installs, vvcdsp_template.c:142-159 */
static int itx_entry_exists(int trh, int trv, int lw, int lh)
{
    if (lw < 0 || lh < 0 || lw > 6 || lh > 6 || trh < 0 || trh > 2 || trv < 0 || trv > 2)
        return 0;
    if (lw == 0 && lh == 0)
        return 0;
    if (lh == 0)            /* w x 1 */
        return trv == ORC_DCT2 && (lw == 4 || lw == 5 || (lw == 6 && trh == ORC_DCT2));
    if (lw == 0)            /* 1 x h */
        return trh == ORC_DCT2 && (lh == 4 || lh == 5 || (lh == 6 && trv == ORC_DCT2));
    if (trh != ORC_DCT2 && (lw < 2 || lw > 5))
        return 0;
    if (trv != ORC_DCT2 && (lh < 2 || lh > 5))
        return 0;
    return 1;
}

ORC_API int orc_itx(int trh, int trv, int log2_w, int log2_h, int *coeffs, size_t nzw, size_t nzh,
    intptr_t log2_transform_range, intptr_t bd)
{
    if (!itx_entry_exists(trh, trv, log2_w, log2_h))
        return -1;
    const int w = 1 << log2_w, h = 1 << log2_h;
    const int range = (int)log2_transform_range;
    const int dc_only_ok = trh == ORC_DCT2 && trv == ORC_DCT2 && nzw == 1 && nzh == 1;

    if (w > 1 && h > 1) {                                            /* itx_2d, vvcdsp.c:94 */
        const int sh1 = 7, sh2 = 5 + range - (int)bd;
        if (w == h && dc_only_ok) {
            const int t = (coeffs[0] * 64 + (1 << (sh1 - 1))) >> sh1;
            const int dc = (t * 64 + (1 << (sh2 - 1))) >> sh2;
            for (int i = 0; i < w * h; i++)
                coeffs[i] = dc;
            return 0;
        }
        for (size_t x = 0; x < nzw; x++)                             /* columns: the vertical type, size h */
            orc_inv_tx_1d(trv, h, coeffs + x, w, nzh);
        for (int y = 0; y < h; y++)                                  /* scale_clip :67 */
            for (int x = 0; x < w; x++)
                coeffs[y * w + x] = (size_t)x < nzw ? orc_clip_intp2((coeffs[y * w + x] + (1 << (sh1 - 1))) >> sh1, range) : 0;
        for (int y = 0; y < h; y++)                                  /* rows: the horizontal type, size w */
            orc_inv_tx_1d(trh, w, coeffs + y * w, 1, nzw);
        for (int i = 0; i < w * h; i++)
            coeffs[i] = (coeffs[i] + (1 << (sh2 - 1))) >> sh2;
    } else {                                                         /* itx_1d, vvcdsp.c:119 */
        const int sh = 6 + range - (int)bd;
        if (dc_only_ok) {
            const int dc = (coeffs[0] * 64 + (1 << (sh - 1))) >> sh;
            for (int i = 0; i < w * h; i++)
                coeffs[i] = dc;
            return 0;
        }
        if (w > 1)
            orc_inv_tx_1d(trh, w, coeffs, 1, nzw);
        else
            orc_inv_tx_1d(trv, h, coeffs, 1, nzh);
        for (int i = 0; i < w * h; i++)
            coeffs[i] = (coeffs[i] + (1 << (sh - 1))) >> sh;
    }
    return 0;
}

/* vvc_itx_1d.c:708 */
ORC_API void orc_inv_lfnst_1d(int *v, const int *u, int no_zero_size, int n_tr_s, int pred_mode_intra, int lfnst_idx,
    int log2_transform_range)
{
    const int set = pred_mode_intra < 0 ? 1 : orc_tab_lfnst_tr_set_index[pred_mode_intra];
    const int8_t *m = n_tr_s > 16 ? orc_tab_lfnst_8x8 + (set * 2 + lfnst_idx - 1) * 16 * 48
                                  : orc_tab_lfnst_4x4 + (set * 2 + lfnst_idx - 1) * 16 * 16;
    for (int j = 0; j < n_tr_s; j++) {
        unsigned t = 0;
        for (int i = 0; i < no_zero_size; i++)
            t += (unsigned)u[i] * (unsigned)(int)m[i * n_tr_s + j];
        v[j] = orc_clip_intp2(((int)t + 64) >> 7, log2_transform_range);
    }
}

/* vvcdsp_template.c:32 */
ORC_API void orc_add_residual(int bd, uint8_t *dst, const int *res, int width, int height, ptrdiff_t stride)
{
    const int wide = bd > 8;
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++)
            orc_st(dst + y * stride, x, orc_clip_px(orc_ld(dst + y * stride, x, wide) + res[y * width + x], bd), wide);
}

/* vvcdsp_template.c:48 */
ORC_API void orc_add_residual_joint(int bd, uint8_t *dst, const int *res, int width, int height, ptrdiff_t stride, int c_sign, int shift)
{
    const int wide = bd > 8;
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            const int r = (res[y * width + x] * c_sign) >> shift;
            orc_st(dst + y * stride, x, orc_clip_px(orc_ld(dst + y * stride, x, wide) + r, bd), wide);
        }
}

/* vvcdsp_template.c:65 */
ORC_API void orc_pred_residual_joint(int *buf, int width, int height, int c_sign, int shift)
{
    for (int i = 0; i < width * height; i++)
        buf[i] = (buf[i] * c_sign) >> shift;
}

/* vvcdsp_template.c:76 — running sums down the columns (vertical) or along the rows, clipped at every step */
ORC_API void orc_transform_bdpcm(int *coeffs, int width, int height, int vertical, int log2_transform_range)
{
    if (vertical) {
        for (int y = 1; y < height; y++)
            for (int x = 0; x < width; x++)
                coeffs[y * width + x] = orc_clip_intp2(coeffs[y * width + x] + coeffs[(y - 1) * width + x], log2_transform_range);
    } else {
        for (int y = 0; y < height; y++)
            for (int x = 1; x < width; x++)
                coeffs[y * width + x] = orc_clip_intp2(coeffs[y * width + x] + coeffs[y * width + x - 1], log2_transform_range);
    }
}

/* vvc_intra.c:277-417 — scaling process for transform coefficients (dequant), flattened as in include/vvc_mi355.h:
 * derive_qp's bd_shift / rect_non_ts_flag (:297-309), derive_scale (:311-338), derive_scale_m's up-sampling and DC
 * override (:373-381), scale_coeff (:391-397), the loop of dequant (:410-417). */
ORC_API void orc_dequant(int *coeffs, int log2_w, int log2_h, int min_x, int min_y, int max_x, int max_y, int qp, int ts,
                         int dep_quant, int bit_depth, int log2_transform_range, const uint8_t *scale_matrix,
                         int log2_matrix_size, int dc)
{
    static const int level_scale[2][6] = { { 40, 45, 51, 57, 64, 72 }, { 57, 64, 72, 80, 90, 102 } };
    const int log_sum = log2_w + log2_h;
    const int rect = ts ? 0 : (log_sum & 1);
    const int bd_shift = ts ? 10 : bit_depth + rect + log_sum / 2 + 10 - log2_transform_range + dep_quant;
    const int bd_offset = (1 << bd_shift) >> 1;
    const int q = qp + ((dep_quant && !ts) ? 1 : 0);
    const int scale = level_scale[rect][q % 6] << (q / 6);
    int first = 1;
    for (int y = min_y; y <= max_y; y++) {
        for (int x = min_x; x <= max_x; x++) {
            int *c = coeffs + (y << log2_w) + x;
            int m = 16;
            if (scale_matrix) {
                const int off = y << log2_matrix_size >> log2_h << log2_matrix_size;
                m = scale_matrix[off + (x << log2_matrix_size >> log2_w)];
                if (first && dc >= 0 && !min_x && !min_y)
                    m = dc;
            }
            first = 0;
            if (*c)
                *c = orc_clip_intp2((int)((unsigned)*c * (unsigned)scale * (unsigned)m + (unsigned)bd_offset) >> bd_shift,
                                    log2_transform_range);
        }
    }
}
