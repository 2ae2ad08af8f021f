// Inverse transform (DCT-2 / DST-7 / DCT-8, 1-D and 2-D, up to 64x64), LFNST, residual add and BDPCM kernels for gfx950.
//
// Reference behaviour: libavcodec/vvc/vvcdsp.c:67-195 (scale_clip, scale, itx_2d, itx_1d, the generator),
// libavcodec/vvc/vvc_itx_1d.c:64-721 (nz gating of the DCT-2 butterflies, matrix_mul, ff_vvc_inv_lfnst_1d),
// libavcodec/vvc/vvcdsp_template.c:32-100 (add_residual, joint variants, transform_bdpcm) and :142-159 (which entries exist).
//
// The reference's partial butterflies are wrapping int32 arithmetic, i.e. exactly a dot product with the transform matrix
// over the inputs its nz gating keeps.  One workgroup owns one transform block: coefficients are staged in LDS, the column
// pass spreads (column, output row) pairs over the lanes, then the row pass does the same for (row, output column).
#include "common.hpp"
#include "runtime.hpp"
#include "../../include/vvc_mi355.h"

namespace vvc355 {

#define VVC355_TABLE(type, name, count) __device__ static const type d_tab_##name[count]
#include "tables.inc"
#undef VVC355_TABLE
// the reference's inline tables as compile-time constants (tables_small.inc): the arithmetic / packed forms the kernels use are proven
// equal to them here
#define VVC355_TABLE(type, name, count) static constexpr type c_##name[count]
#include "tables_small.inc"
#undef VVC355_TABLE
// levelScale[rect_non_ts][qp % 6] of the scaling process (level_scale, vvc_intra.c:329-336)
__host__ __device__ constexpr int level_scale_of(int rect, int rem)
{
    return rect ? (rem == 0 ? 57 : rem == 1 ? 64 : rem == 2 ? 72 : rem == 3 ? 80 : rem == 4 ? 90 : 102)
                : (rem == 0 ? 40 : rem == 1 ? 45 : rem == 2 ? 51 : rem == 3 ? 57 : rem == 4 ? 64 : 72);
}
// 6.5.2 up-right diagonal scan of a 4x4 block (ff_vvc_diag_scan_x / _y [2][2], vvc_data.c:27,152), one 4-bit field per scan position
static constexpr unsigned long long kDiag4X = 0x3323213210210100ull, kDiag4Y = 0x3231230123012010ull;
constexpr bool itx_small_tables_match()
{
    for (int r = 0; r < 2; r++)
        for (int q = 0; q < 6; q++)
            if (level_scale_of(r, q) != c_level_scale[r * 6 + q]) return false;
    for (int i = 0; i < 16; i++)
        if ((int)((kDiag4X >> (4 * i)) & 15) != c_diag_scan_4x4_x[i] || (int)((kDiag4Y >> (4 * i)) & 15) != c_diag_scan_4x4_y[i]) return false;
    return true;
}
static_assert(itx_small_tables_match(), "level scale / 4x4 diagonal scan differ from vvc_intra.c:329-336 / vvc_data.c:27,152");

enum { TX_DCT2 = 0, TX_DST7 = 1, TX_DCT8 = 2 };

__device__ __forceinline__ const int8_t *dxt_matrix(int type, int n)
{
    if (type == TX_DST7)
        return n == 4 ? d_tab_dst7_4 : n == 8 ? d_tab_dst7_8 : n == 16 ? d_tab_dst7_16 : d_tab_dst7_32;
    return n == 4 ? d_tab_dct8_4 : n == 8 ? d_tab_dct8_8 : n == 16 ? d_tab_dct8_16 : d_tab_dct8_32;
}

// Number of leading inputs a 1-D transform of size n reads for a given nz (vvc_itx_1d.c:64-67,:498,:659):
// DCT-2 gates inputs in groups {0,1},{2,3},{4..7},{8..15},{16..31} and never reads inputs >= 32 of the 64-point transform;
// DST-7 / DCT-8 read exactly nz (<= 16) inputs.
__device__ __forceinline__ int inputs_used(int type, int n, int nz)
{
    if (type != TX_DCT2)
        return nz;
    int used = 2;
    while (used < nz)
        used <<= 1;                  // k takes part iff k < 2 or nz > 2^floor(log2 k)  <=>  k < used
    used = min(used, n);
    return n == 64 ? min(used, 32) : used;
}

// output i of an n-point inverse transform of `cnt` inputs in[0], in[step], ...
__device__ __forceinline__ int inv_tx_out(int type, int n, int i, const int *in, int step, int cnt, const int8_t *cos_lds)
{
    unsigned acc = 0;
    if (type == TX_DCT2) {
        const int ang = (2 * i + 1) * (64 / n);
        for (int k = 0; k < cnt; k++)
            acc += (unsigned)in[k * step] * (unsigned)(int)cos_lds[(ang * k) & 255];
    } else {
        const int8_t *m = dxt_matrix(type, n) + i;
        for (int k = 0; k < cnt; k++)
            acc += (unsigned)in[k * step] * (unsigned)(int)m[k * n];
    }
    return (int)acc;
}

// Scaling process for transform coefficients (vvc_intra.c:277-417) as a per-coefficient function: derive_qp's shift and
// rectangular correction (:297-309), derive_scale (:311-338), derive_scale_m's up-sampling and DC override (:373-381),
// scale_coeff (:391-397).  Shared by dequant_kernel and by the itx kernels' fused load stage.
// derive_transform_type (vvc_intra.c:130-164): implicit / explicit MTS -> trh | trv << 4; flags = VVC355_TU_*
__host__ __device__ inline int derive_tr_type(int flags, int mts_idx, int lfnst_idx, int c_idx, int w, int h)
{
    const bool isp = flags & VVC355_TU_ISP, sbt = flags & VVC355_TU_SBT;
    if (c_idx || (isp && lfnst_idx))
        return 0;
    bool implicit = false;
    if (flags & VVC355_TU_MTS_ENABLED)
        implicit = isp || (sbt && (w > h ? w : h) <= 32) ||
                   (!(flags & VVC355_TU_EXPLICIT_MTS_INTRA) && (flags & VVC355_TU_INTRA) && !lfnst_idx && !(flags & VVC355_TU_MIP));
    if (implicit) {
        int trh, trv;
        if (sbt) {
            const bool hor = flags & VVC355_TU_SBT_HORIZONTAL, pos = flags & VVC355_TU_SBT_POS;
            trh = (hor || pos) ? 1 : 2;
            trv = (!hor || pos) ? 1 : 2;
        } else {
            trh = (w >= 4 && w <= 16) ? 1 : 0;
            trv = (h >= 4 && h <= 16) ? 1 : 0;
        }
        return trh | (trv << 4);
    }
    // mts_idx -> (trh, trv): { DCT2, DST7, DCT8, DST7, DCT8 } / { DCT2, DST7, DST7, DCT8, DCT8 }
    const int trh = mts_idx == 0 ? 0 : (mts_idx & 1) ? 1 : 2, trv = mts_idx == 0 ? 0 : mts_idx <= 2 ? 1 : 2;
    return trh | (trv << 4);
}
// a job that asks for it gets its transform types from the rule above instead of from its trh / trv fields
__device__ __forceinline__ void resolve_type(vvc355_itx_job &job)
{
    if (job.mts_flags & VVC355_ITX_DERIVE_TYPE) {
        const int t = derive_tr_type(job.tu_flags, job.mts_idx, job.lfnst_idx, job.c_idx, 1 << job.log2_w, 1 << job.log2_h);
        job.trh = (uint8_t)(t & 15);
        job.trv = (uint8_t)(t >> 4);
    }
}

struct Dequant {
    int on, scale, bd_shift, bd_offset, range, lw, lh, lm, dc;
    const uint8_t *sm;
    __device__ __forceinline__ void setup(int enable, int lw_, int lh_, int qp_in, int ts, int dep_quant, int bit_depth, int range_,
                                          const uint8_t *sm_, int lm_, int dc_)
    {
        on = enable; lw = lw_; lh = lh_; range = range_; sm = sm_; lm = lm_; dc = dc_;
        const int log_sum = lw + lh, rect = ts ? 0 : (log_sum & 1);
        bd_shift = ts ? 10 : bit_depth + rect + (log_sum / 2) + 10 - range + dep_quant;
        bd_offset = (1 << bd_shift) >> 1;
        const int qp = qp_in + (dep_quant && !ts ? 1 : 0), rem = qp % 6;
        scale = level_scale_of(rect, rem) << (qp / 6);
    }
    __device__ __forceinline__ int apply(int c, int x, int y) const
    {
        if (!on || !c)
            return c;
        int m = 16;
        if (sm) {
            m = gld<uint8_t>(sm + (((y << lm) >> lh) << lm) + ((x << lm) >> lw));
            if (dc >= 0 && x == 0 && y == 0)
                m = dc;
        }
        return clip_intp2((int)((unsigned)c * (unsigned)scale * (unsigned)m + (unsigned)bd_offset) >> bd_shift, range);
    }
    // the same for |c| < 2^15 (every level a conforming stream codes): c * m and (c * m) * scale are 24-bit multiplies, full
    // rate instead of two quarter-rate 32-bit multiplies; products wrap to 32 bits exactly like the expression above
    __device__ __forceinline__ int apply_small(int c, int x, int y) const
    {
        if (!on || !c)
            return c;
        int m = 16;
        if (sm) {
            m = gld<uint8_t>(sm + (((y << lm) >> lh) << lm) + ((x << lm) >> lw));
            if (dc >= 0 && x == 0 && y == 0)
                m = dc;
        }
        return clip_intp2((int)((unsigned)__mul24(__mul24(c, m), scale) + (unsigned)bd_offset) >> bd_shift, range);
    }
};

__device__ __forceinline__ Dequant itx_job_dequant(const vvc355_itx_job &job, int bd)
{
    Dequant dq;
    dq.setup(job.dq_flags & 1, job.log2_w, job.log2_h, job.dq_qp, 0, (job.dq_flags >> 1) & 1, bd, job.range,
             (const uint8_t *)job.scale_matrix, job.log2_matrix_size, job.dc);
    return dq;
}

#define ITX_SYNC()                                                                  \
    do {                                                                            \
        if (WAVE) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } \
        else __syncthreads();                                                       \
    } while (0)

// One transform block of any shape with w * h <= CAP, worked on by the NT lanes `tid` = 0..NT-1 of a group (NT <= 64: the
// group sits inside one wave and synchronises at wave level; NT = 256: the whole workgroup).  buf / tmp: CAP ints of LDS each.
template <int BD, int NT, int CAP>
__device__ __forceinline__ void itx_generic_block(const vvc355_itx_job &job, int *buf, int *tmp, const int8_t *cos_lds, int tid)
{
    constexpr bool WAVE = NT <= 64;                  // the group lives inside one wave
    const int w = 1 << job.log2_w, h = 1 << job.log2_h, n = w * h;
    const int nzw = job.nzw, nzh = job.nzh, range = job.range, bd = job.bd ? job.bd : BD;
    int *coeffs = (int *)job.coeffs;

    // I/O mapping: lane `tid` owns the PER consecutive elements starting at tid * PER (row-major), so coefficients move as
    // 16-byte vectors and pixels as 8/16-byte row segments.  The prediction samples the residual is added to are requested
    // together with the coefficients, long before they are needed (one memory round trip on the critical path, not two).
    constexpr int PER = CAP >= NT ? CAP / NT : 1;    // 1, 4, 4, 16
    using px_t = typename Px<BD>::type;
    uint8_t *dst = (uint8_t *)job.dst;
    const int e0 = tid * PER;
    const bool row_io = PER > 1 && w >= PER;         // the lane's elements sit in one row: vector pixel access
    px_t pred[PER];
    if (dst && e0 < n) {
        if (row_io) {
            const VVC355_GLOBAL px_t *prow = (const VVC355_GLOBAL px_t *)(dst + (ptrdiff_t)(e0 >> job.log2_w) * job.dst_stride) + (e0 & (w - 1));
#pragma unroll
            for (int q = 0; q < PER; q++)
                pred[q] = prow[q];                   // contiguous, aligned to PER samples: merged into wide loads
        } else {
#pragma unroll
            for (int q = 0; q < PER; q++) {
                const int o = e0 + q;
                pred[q] = o < n ? (px_t)ld_px<BD>(dst + (ptrdiff_t)(o >> job.log2_w) * job.dst_stride, o & (w - 1)) : (px_t)0;
            }
        }
    }
    const Dequant dq = itx_job_dequant(job, bd);
    if (PER < 4) {
#pragma unroll
        for (int q = 0; q < PER; q++)
            if (e0 + q < n)
                buf[e0 + q] = dq.apply(gld<int>(coeffs + e0 + q), (e0 + q) & (w - 1), (e0 + q) >> job.log2_w);
    } else {
#pragma unroll
        for (int c4 = 0; c4 < PER / 4; c4++) {
            const int e = e0 + c4 * 4;
            if (e < n) {                             // n is a multiple of 4 for every block of >= 4 coefficients
                int4 v = gld<int4>(coeffs + e);
                if (dq.on) {
                    const int y = e >> job.log2_w, x = e & (w - 1);      // w >= 4 here: the four share a row
                    v.x = dq.apply(v.x, x, y); v.y = dq.apply(v.y, x + 1, y); v.z = dq.apply(v.z, x + 2, y); v.w = dq.apply(v.w, x + 3, y);
                }
                *(int4 *)&buf[e] = v;
            }
        }
    }
    ITX_SYNC();

    const bool dc_only = job.trh == TX_DCT2 && job.trv == TX_DCT2 && nzw == 1 && nzh == 1;
    int sh_final;
    if (w > 1 && h > 1) {
        const int sh1 = 7;
        sh_final = 5 + range - bd;
        if (w == h && dc_only) {
            const int t = (buf[0] * 64 + (1 << (sh1 - 1))) >> sh1;
            const int dc = (t * 64 + (1 << (sh_final - 1))) >> sh_final;
            ITX_SYNC();
            for (int i = tid; i < n; i += NT)
                buf[i] = dc;
            sh_final = -1;
        } else if (w >= 4 && h >= 4) {
            // Register-tiled passes: every lane accumulates FOUR outputs that share the matrix entry, reading the four
            // inputs as one 16-byte LDS vector, i.e. one LDS vector + one table byte + four multiply-adds per four products.
            // Column pass: lane -> (row y, columns x0..x0+3); result stored transposed (tmp[x][y]) so that the row pass can
            // do the same with lane -> (column x, rows y0..y0+3).
            const int cnt = inputs_used(job.trv, h, nzh);
            const int cnt2 = inputs_used(job.trh, w, nzw);          // the row pass reads columns < cnt2 (zero beyond nzw)
            // 24-bit multiplies are exact when every input magnitude is below 2^23: always true for the clipped intermediates
            // of the row pass (range <= 20), checked here for the coefficients (the decoder's dequantiser clips them to range)
            bool small = true;
            for (int i = tid; i < n; i += NT)
                small &= (unsigned)(buf[i] + (1 << 23)) < (1u << 24);
            bool fast1;
            if (WAVE) {
                // group vote inside the wave: the NT lanes of this block occupy an aligned bit field of the ballot
                const unsigned long long gm = (NT == 64 ? ~0ull : ((1ull << (NT & 63)) - 1)) << ((threadIdx.x & 63) & ~(NT - 1));
                fast1 = (__ballot(small) & gm) == gm;
            } else {
                fast1 = (bool)__syncthreads_and(small);
            }
            const int gx = (cnt2 + 3) >> 2, lgh = job.log2_h;
            const int8_t *mv = job.trv == TX_DCT2 ? nullptr : dxt_matrix(job.trv, h);
            for (int g = tid; g < (gx << lgh); g += NT) {
                const int y = g & (h - 1), x0 = (g >> lgh) << 2;
                int acc[4] = { 0, 0, 0, 0 };
                if (x0 < nzw) {
                    const int ang = (2 * y + 1) * (64 >> lgh);
                    int a = 0;
                    for (int k = 0; k < cnt; k++) {
                        const int m = mv ? (int)mv[k * h + y] : (int)cos_lds[a];
                        a = (a + ang) & 255;
                        const int4 v = *(const int4 *)&buf[k * w + x0];
                        if (fast1) {
                            acc[0] += __mul24(m, v.x); acc[1] += __mul24(m, v.y); acc[2] += __mul24(m, v.z); acc[3] += __mul24(m, v.w);
                        } else {
                            acc[0] += m * v.x; acc[1] += m * v.y; acc[2] += m * v.z; acc[3] += m * v.w;
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int x = x0 + q;
                    if (x < w)
                        tmp[x * h + y] = x < nzw ? clip_intp2((acc[q] + (1 << (sh1 - 1))) >> sh1, range) : 0;
                }
            }
            ITX_SYNC();
            const int8_t *mh = job.trh == TX_DCT2 ? nullptr : dxt_matrix(job.trh, w);
            for (int g = tid; g < (n >> 2); g += NT) {
                const int x = g & (w - 1), y0 = (g >> job.log2_w) << 2;
                const int ang = (2 * x + 1) * (64 >> job.log2_w);
                int acc[4] = { 0, 0, 0, 0 }, a = 0;
                for (int k = 0; k < cnt2; k++) {
                    const int m = mh ? (int)mh[k * w + x] : (int)cos_lds[a];
                    a = (a + ang) & 255;
                    const int4 v = *(const int4 *)&tmp[k * h + y0];
                    acc[0] += __mul24(m, v.x); acc[1] += __mul24(m, v.y); acc[2] += __mul24(m, v.z); acc[3] += __mul24(m, v.w);
                }
                // the old contents of buf (the coefficients) are dead once every lane has left the column pass
#pragma unroll
                for (int q = 0; q < 4; q++)
                    buf[(y0 + q) * w + x] = acc[q];
            }
        } else {
            // column pass on columns < nzw (vertical type, size h), then scale_clip; other columns become zero
            const int cnt = inputs_used(job.trv, h, nzh);
            for (int o = tid; o < n; o += NT) {
                const int y = o >> job.log2_w, x = o & (w - 1);
                int v = 0;
                if (x < nzw) {
                    v = inv_tx_out(job.trv, h, y, buf + x, w, cnt, cos_lds);
                    v = clip_intp2((v + (1 << (sh1 - 1))) >> sh1, range);
                }
                tmp[o] = v;
            }
            ITX_SYNC();
            // row pass (horizontal type, size w) with nz = nzw
            const int cnt2 = inputs_used(job.trh, w, nzw);
            for (int o = tid; o < n; o += NT) {
                const int y = o >> job.log2_w, x = o & (w - 1);
                buf[o] = inv_tx_out(job.trh, w, x, tmp + y * w, 1, cnt2, cos_lds);
            }
        }
    } else {
        sh_final = 6 + range - bd;
        if (dc_only) {
            const int dc = (buf[0] * 64 + (1 << (sh_final - 1))) >> sh_final;
            ITX_SYNC();
            for (int i = tid; i < n; i += NT)
                buf[i] = dc;
            sh_final = -1;
        } else {
            const int type = w > 1 ? job.trh : job.trv, nz = w > 1 ? nzw : nzh;
            const int cnt = inputs_used(type, n, nz);
            for (int o = tid; o < n; o += NT)
                tmp[o] = inv_tx_out(type, n, o, buf, 1, cnt, cos_lds);
            ITX_SYNC();
            for (int o = tid; o < n; o += NT)
                buf[o] = tmp[o];
        }
    }
    ITX_SYNC();
    // final scale, then store residuals in place (slot semantics) and/or add them to the prediction
    if (e0 < n) {
        int r[PER];
#pragma unroll
        for (int q = 0; q < PER; q++) {
            const int o = e0 + q;
            r[q] = o < n ? (sh_final < 0 ? buf[o] : (buf[o] + (1 << (sh_final - 1))) >> sh_final) : 0;
        }
        if (job.store_coeffs) {
            if (PER < 4) {
#pragma unroll
                for (int q = 0; q < PER; q++)
                    if (e0 + q < n)
                        gst<int>(coeffs + e0 + q, r[q]);
            } else {
#pragma unroll
                for (int c4 = 0; c4 < PER / 4; c4++)
                    if (e0 + c4 * 4 < n)
                        gst<int4>(coeffs + e0 + c4 * 4, make_int4(r[c4 * 4], r[c4 * 4 + 1], r[c4 * 4 + 2], r[c4 * 4 + 3]));
            }
        }
        if (dst) {
            if (row_io) {
                VVC355_GLOBAL px_t *prow = (VVC355_GLOBAL px_t *)(dst + (ptrdiff_t)(e0 >> job.log2_w) * job.dst_stride) + (e0 & (w - 1));
                px_t outv[PER];
#pragma unroll
                for (int q = 0; q < PER; q++)
                    outv[q] = (px_t)clip_px<BD>((int)pred[q] + r[q]);
#pragma unroll
                for (int q = 0; q < PER; q++)
                    prow[q] = outv[q];
            } else {
#pragma unroll
                for (int q = 0; q < PER; q++) {
                    const int o = e0 + q;
                    if (o < n)
                        st_px<BD>(dst + (ptrdiff_t)(o >> job.log2_w) * job.dst_stride, o & (w - 1), clip_px<BD>((int)pred[q] + r[q]));
                }
            }
        }
    }
}

// NT lanes share one block of at most CAP coefficients (CAP / NT = 4 elements per lane, 16 for 64x64):
//   NT 4 / CAP 16 (4x4), NT 16 / CAP 64 (8x8), NT 64 / CAP 256 (16x16): sub-wave groups, wave-level synchronisation only;
//   NT 256 / CAP 1024 (32x32) and NT 256 / CAP 4096 (64x64): one workgroup per block.
template <int BD, int NT, int CAP>
__global__ __launch_bounds__(256) void itx_kernel(const vvc355_itx_job *__restrict__ jobs, int n_jobs)
{
    constexpr int TBS = 256 / NT;                    // blocks per workgroup
    __shared__ __attribute__((aligned(16))) int buf_all[TBS][CAP];
    __shared__ __attribute__((aligned(16))) int tmp_all[TBS][CAP];
    __shared__ int8_t cos_lds[256];
    cos_lds[threadIdx.x] = d_tab_dct2_cos[threadIdx.x];
    __syncthreads();
    const int sub = threadIdx.x / NT, tid = threadIdx.x % NT;
    const int ji = blockIdx.x * TBS + sub;
    if (ji >= n_jobs)
        return;                                      // whole groups leave together
    vvc355_itx_job job = jobs[ji];
    resolve_type(job);
    if (job.log2_w + job.log2_h > __builtin_ctz(CAP))
        return;                                      // larger than this launch's size class: contract violation, skipped
    itx_generic_block<BD, NT, CAP>(job, buf_all[sub], tmp_all[sub], cos_lds, tid);
}

// ------------------------------------------------------------------------------------------------ shape-specialised path
//
// All jobs of a launch share one shape W x H (both 4..64).  With log2_transform_range <= 15 the coefficients the reference
// reads and its clipped first-stage results are 16-bit, so both passes are packed 16-bit dot products (v_dot2_i32_i16, two
// multiply-adds per lane per instruction; the true sums stay far below 2^31, so the reference's wrapping int32 sums are
// reproduced exactly).  A block is cut into 4x4 tiles, one lane per tile in each pass; per 8 inputs a lane reads 4 matrix
// rows and 4 data rows as 16-byte LDS vectors and issues 64 dot products.  LDS images (int16, input index contiguous):
//   cT [x][k]  the coefficients the column pass reads, transposed; rows / columns the nz gating excludes are zero
//   tmp[y][k]  the clipped column-pass output, only the columns the row pass reads
//   tab[type][out][k]  the transform matrices of this shape (generated: itx16_<N> in tables.inc)
// Only the coefficients inside the nz window are fetched from HBM.  A workgroup in which any block is not eligible
// (other shape, range > 15, a coefficient beyond 16 bits) redoes all of its blocks with the generic code above.

typedef short itx_v2s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int dot2_i16(uint32_t a, uint32_t b, int acc)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(itx_v2s, a), __builtin_bit_cast(itx_v2s, b), acc, false);
}
__device__ __forceinline__ uint32_t pack_i16(int lo, int hi) { return ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16); }

template <int N> struct TxDim {
    static constexpr int KV = N < 32 ? N : 32;               // inputs an N-point inverse transform can read
    static constexpr int P = KV + (KV >= 16 ? 8 : 0);        // LDS row pitch (int16): 16-byte aligned rows, staggered banks
    static constexpr int KS = KV < 8 ? 4 : 8;                // inputs per LDS vector
    static constexpr int NTYPE = N <= 32 ? 3 : 1;            // DST-7 / DCT-8 exist up to 32 points
    __device__ static __forceinline__ const int16_t *table()
    {
        return N == 4 ? d_tab_itx16_4 : N == 8 ? d_tab_itx16_8 : N == 16 ? d_tab_itx16_16 : N == 32 ? d_tab_itx16_32 : d_tab_itx16_64;
    }
    // global -> LDS, re-pitched
    __device__ static __forceinline__ void stage(int16_t *lds)
    {
        const int16_t *src = table();
        constexpr int ND = NTYPE * N * KV / 2;
        for (int i = threadIdx.x; i < ND; i += 256) {
            const int row = i / (KV / 2), c2 = i % (KV / 2);
            *(uint32_t *)&lds[row * P + 2 * c2] = gld<uint32_t>(src + 2 * i);
        }
    }
};

template <int KS> __device__ __forceinline__ void lds_row(const int16_t *p, uint32_t (&d)[KS / 2])
{
    if constexpr (KS == 8) {
        const uint4 v = *(const uint4 *)p;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    } else {
        const uint2 v = *(const uint2 *)p;
        d[0] = v.x; d[1] = v.y;
    }
}

template <int BD, int LW, int LH>
__global__ __launch_bounds__(256) void itx_shape_kernel(const vvc355_itx_job *__restrict__ jobs, int n_jobs)
{
    using px_t = typename Px<BD>::type;
    constexpr int W = 1 << LW, H = 1 << LH, CAP = W * H;
    constexpr int NT = CAP / 16, TBS = 256 / NT;             // lanes per block (one per 4x4 tile), blocks per workgroup
    using DV = TxDim<H>;                                     // vertical transform: k runs over rows
    using DH = TxDim<W>;                                     // horizontal transform: k runs over columns
    constexpr int KVV = DV::KV, PV = DV::P, KSV = DV::KS;
    constexpr int KVH = DH::KV, PH = DH::P, KSH = DH::KS;
    constexpr bool WAVE = NT <= 64;
    constexpr int CT_SZ = KVH * PV, TMP_SZ = H * PH;
    constexpr int FAST_BYTES = TBS * (CT_SZ + TMP_SZ) * 2, GEN_BYTES = CAP * 8;
    __shared__ __attribute__((aligned(16))) char lds_raw[FAST_BYTES > GEN_BYTES ? FAST_BYTES : GEN_BYTES];
    __shared__ __attribute__((aligned(16))) int16_t tab_v[DV::NTYPE * H * PV];
    __shared__ __attribute__((aligned(16))) int16_t tab_h_own[W == H ? 8 : DH::NTYPE * W * PH];
    __shared__ int8_t cos_lds[256];
    const int16_t *tab_h = W == H ? tab_v : tab_h_own;

    DV::stage(tab_v);
    if (W != H)
        DH::stage(tab_h_own);
    cos_lds[threadIdx.x] = d_tab_dct2_cos[threadIdx.x];

    const int sub = threadIdx.x / NT, tid = threadIdx.x % NT;
    const int wg = xcd_chunked(blockIdx.x, gridDim.x);
    const int ji = wg * TBS + sub;
    const bool valid = ji < n_jobs;
    vvc355_itx_job job = jobs[valid ? ji : n_jobs - 1];
    resolve_type(job);
    const int nzw = job.nzw, nzh = job.nzh, range = job.range, bd = job.bd ? job.bd : BD;
    const int trh = job.trh, trv = job.trv;
    const int sh_final = 5 + range - bd;
    bool ok = job.log2_w == LW && job.log2_h == LH && range <= 15 && sh_final >= 1 && trh < DH::NTYPE && trv < DV::NTYPE;
    const bool dc_only = W == H && trh == TX_DCT2 && trv == TX_DCT2 && nzw == 1 && nzh == 1;
    const int cntv = dc_only ? 1 : inputs_used(trv, H, nzh);                // rows the column pass reads
    const int cnt2 = inputs_used(trh, W, nzw);                              // columns the row pass reads
    ok &= cntv <= KVV && cnt2 <= KVH;
    const int cnt2r = (cnt2 + KSH - 1) & ~(KSH - 1);

    // this lane's tile for I/O and for the row pass
    const int y0 = (tid / (W / 4)) * 4, x0 = (tid % (W / 4)) * 4;
    int *coeffs = (int *)job.coeffs;
    uint8_t *dst = (uint8_t *)job.dst;
    const bool act = valid && ok;

    // prediction samples first: they are needed last
    uint2 praw[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        praw[r] = make_uint2(0, 0);
        if (act && dst) {
            const uint8_t *p = dst + row_off(y0 + r, job.dst_stride) + x0 * (int)sizeof(px_t);
            if (BD > 8) praw[r] = gld<uint2>(p);
            else praw[r].x = gld<uint32_t>(p);
        }
    }
    int c[4][4];
    const Dequant dq = itx_job_dequant(job, bd);
    const bool need = act && y0 < cntv && x0 < nzw && x0 < KVH;
    unsigned mag = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        int4 v = make_int4(0, 0, 0, 0);
        if (need && y0 + r < cntv)
            v = gld<int4>(coeffs + (y0 + r) * W + x0);
        if (dq.on) {
            const unsigned lv = (unsigned)(v.x ^ (v.x >> 31)) | (unsigned)(v.y ^ (v.y >> 31)) | (unsigned)(v.z ^ (v.z >> 31)) | (unsigned)(v.w ^ (v.w >> 31));
            if ((lv >> 15) == 0) {
                v.x = dq.apply_small(v.x, x0, y0 + r); v.y = dq.apply_small(v.y, x0 + 1, y0 + r);
                v.z = dq.apply_small(v.z, x0 + 2, y0 + r); v.w = dq.apply_small(v.w, x0 + 3, y0 + r);
            } else {
                v.x = dq.apply(v.x, x0, y0 + r); v.y = dq.apply(v.y, x0 + 1, y0 + r);
                v.z = dq.apply(v.z, x0 + 2, y0 + r); v.w = dq.apply(v.w, x0 + 3, y0 + r);
            }
        }
        c[r][0] = x0 + 0 < nzw ? v.x : 0; c[r][1] = x0 + 1 < nzw ? v.y : 0;
        c[r][2] = x0 + 2 < nzw ? v.z : 0; c[r][3] = x0 + 3 < nzw ? v.w : 0;
#pragma unroll
        for (int q = 0; q < 4; q++)
            mag |= (unsigned)(c[r][q] ^ (c[r][q] >> 31));
    }
    ok &= (mag >> 15) == 0;
    if (!__syncthreads_and(!valid || ok)) {
        // some block of this workgroup needs the generic arithmetic: redo them all, one after the other, 256 lanes each
        int *gbuf = (int *)lds_raw, *gtmp = gbuf + CAP;
        for (int b = 0; b < TBS; b++) {
            const int jb = wg * TBS + b;
            if (jb >= n_jobs)
                break;
            vvc355_itx_job jg = jobs[jb];
            resolve_type(jg);
            if (jg.log2_w + jg.log2_h <= LW + LH)
                itx_generic_block<BD, 256, CAP>(jg, gbuf, gtmp, cos_lds, threadIdx.x);
            __syncthreads();
        }
        return;
    }
    if (WAVE && !valid)
        return;                                              // whole groups inside a wave; no workgroup barrier follows

    int16_t *cT = (int16_t *)lds_raw + sub * (CT_SZ + TMP_SZ), *tmp = cT + CT_SZ;
    if (y0 < KVV && x0 < KVH) {
#pragma unroll
        for (int q = 0; q < 4; q++)
            *(uint2 *)&cT[(x0 + q) * PV + y0] = make_uint2(pack_i16(c[0][q], c[1][q]), pack_i16(c[2][q], c[3][q]));
    }
    ITX_SYNC();

    // ---- column pass: tile (rows ya.., columns xa..) of tmp, only the columns the row pass reads
    {
        constexpr int YT = H / 4;
        const int xa = (tid / YT) * 4, ya = (tid % YT) * 4;
        if (xa < cnt2r) {
            int acc[4][4];
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int q = 0; q < 4; q++) acc[r][q] = 0;
            if (xa < nzw) {
                const int16_t *mrow = tab_v + (trv * H + ya) * PV;
                const int16_t *crow = cT + xa * PV;
                for (int k = 0; k < cntv; k += KSV) {
                    uint32_t m[4][KSV / 2], d[4][KSV / 2];
#pragma unroll
                    for (int r = 0; r < 4; r++) lds_row<KSV>(mrow + r * PV + k, m[r]);
#pragma unroll
                    for (int q = 0; q < 4; q++) lds_row<KSV>(crow + q * PV + k, d[q]);
#pragma unroll
                    for (int r = 0; r < 4; r++)
#pragma unroll
                        for (int q = 0; q < 4; q++)
#pragma unroll
                            for (int e = 0; e < KSV / 2; e++) acc[r][q] = dot2_i16(m[r][e], d[q][e], acc[r][q]);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
                int v[4];
#pragma unroll
                for (int q = 0; q < 4; q++) v[q] = clip_intp2((acc[r][q] + 64) >> 7, range);
                *(uint2 *)&tmp[(ya + r) * PH + xa] = make_uint2(pack_i16(v[0], v[1]), pack_i16(v[2], v[3]));
            }
        }
    }
    ITX_SYNC();

    // ---- row pass on this lane's I/O tile
    int acc[4][4];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int q = 0; q < 4; q++) acc[r][q] = 0;
    {
        const int16_t *mrow = tab_h + (trh * W + x0) * PH;
        const int16_t *trow = tmp + y0 * PH;
        for (int k = 0; k < cnt2; k += KSH) {
            uint32_t m[4][KSH / 2], d[4][KSH / 2];
#pragma unroll
            for (int q = 0; q < 4; q++) lds_row<KSH>(mrow + q * PH + k, m[q]);
#pragma unroll
            for (int r = 0; r < 4; r++) lds_row<KSH>(trow + r * PH + k, d[r]);
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int e = 0; e < KSH / 2; e++) acc[r][q] = dot2_i16(m[q][e], d[r][e], acc[r][q]);
        }
    }
    if (!valid)
        return;
    const int rnd = 1 << (sh_final - 1);
#pragma unroll
    for (int r = 0; r < 4; r++) {
        int res[4];
#pragma unroll
        for (int q = 0; q < 4; q++) res[q] = (acc[r][q] + rnd) >> sh_final;
        if (job.store_coeffs)
            gst<int4>(coeffs + (y0 + r) * W + x0, make_int4(res[0], res[1], res[2], res[3]));
        if (dst) {
            uint8_t *p = dst + row_off(y0 + r, job.dst_stride) + x0 * (int)sizeof(px_t);
            if (BD > 8) {
                const int o0 = clip_px<BD>((int)(praw[r].x & 0xffff) + res[0]), o1 = clip_px<BD>((int)(praw[r].x >> 16) + res[1]);
                const int o2 = clip_px<BD>((int)(praw[r].y & 0xffff) + res[2]), o3 = clip_px<BD>((int)(praw[r].y >> 16) + res[3]);
                gst<uint2>(p, make_uint2((uint32_t)o0 | ((uint32_t)o1 << 16), (uint32_t)o2 | ((uint32_t)o3 << 16)));
            } else {
                const uint32_t pr = praw[r].x;
                const int o0 = clip_px<BD>((int)(pr & 0xff) + res[0]), o1 = clip_px<BD>((int)((pr >> 8) & 0xff) + res[1]);
                const int o2 = clip_px<BD>((int)((pr >> 16) & 0xff) + res[2]), o3 = clip_px<BD>((int)(pr >> 24) + res[3]);
                gst<uint32_t>(p, (uint32_t)o0 | ((uint32_t)o1 << 8) | ((uint32_t)o2 << 16) | ((uint32_t)o3 << 24));
            }
        }
    }
}
#undef ITX_SYNC

// Scaling process for transform coefficients (vvc_intra.c:277-417): 16 lanes per transform block (most blocks are small and
// their non-zero windows smaller still), lanes over the scan rectangle.  levelScale / qp arithmetic per derive_qp (:277) and
// derive_scale (:311).
__global__ __launch_bounds__(256) void dequant_kernel(const vvc355_dequant_job *__restrict__ jobs, int n_jobs)
{
    const int ji = blockIdx.x * 16 + (threadIdx.x >> 4), tid = threadIdx.x & 15;
    if (ji >= n_jobs)
        return;
    const vvc355_dequant_job job = jobs[ji];
    const int lw = job.log2_w;
    Dequant dq;
    dq.setup(1, lw, job.log2_h, job.qp, job.ts, job.dep_quant, job.bit_depth, job.range, (const uint8_t *)job.scale_matrix,
             job.log2_matrix_size, job.dc);
    const int rw = job.max_x - job.min_x + 1, rh = job.max_y - job.min_y + 1;
    int *coeffs = (int *)job.coeffs;
    // lane -> column (tid mod rw') with rw' = rw rounded up to a power of two <= 16, so that no division is needed per element
    const int cw = rw >= 16 ? 16 : rw > 8 ? 16 : rw > 4 ? 8 : rw > 2 ? 4 : rw > 1 ? 2 : 1;    // columns per pass
    const int rows_per_pass = 16 / cw;
    for (int xb = 0; xb < rw; xb += 16) {
        const int xo = xb + (tid & (cw - 1));
        if (xo >= rw)
            continue;
        for (int yo = tid / cw; yo < rh; yo += rows_per_pass) {
            const int y = job.min_y + yo, x = job.min_x + xo;
            const int c = gld<int>(coeffs + (y << lw) + x);
            if (c)
                gst<int>(coeffs + (y << lw) + x, dq.apply(c, x, y));
        }
    }
}

// add_residual / add_residual_joint / pred_residual_joint (vvcdsp_template.c:32,48,65); job.src0 = int residuals,
// mode 0 add, 1 joint add (w0 = c_sign, denom = shift), 2 joint in place on the int buffer (dst unused)
template <int BD>
__global__ __launch_bounds__(256) void residual_kernel(const vvc355_blend_job *__restrict__ jobs)
{
    const vvc355_blend_job job = load_uniform(jobs + (blockIdx.y));
    int *res = (int *)job.src0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < job.w * job.h; i += gridDim.x * blockDim.x) {
        int r = res[i];
        if (job.mode)
            r = (r * job.w0) >> job.denom;
        if (job.mode == 2) {
            res[i] = r;
            continue;
        }
        const int y = i / job.w, x = i - y * job.w;
        uint8_t *row = (uint8_t *)job.dst + (ptrdiff_t)y * job.dst_stride;
        st_px<BD>(row, x, clip_px<BD>(ld_px<BD>(row, x) + r));
    }
}

// transform_bdpcm (:76): one lane per column (vertical) or row; job.dst = int coeffs, mode = vertical, denom = range
__global__ __launch_bounds__(128) void bdpcm_kernel(const vvc355_blend_job *__restrict__ jobs)
{
    const vvc355_blend_job job = load_uniform(jobs + (blockIdx.x));
    int *c = (int *)job.dst;
    const int w = job.w, h = job.h, range = job.denom, t = threadIdx.x;
    if (job.mode) {
        if (t < w)
            for (int y = 1; y < h; y++)
                c[y * w + t] = clip_intp2(c[y * w + t] + c[(y - 1) * w + t], range);
    } else {
        if (t < h)
            for (int x = 1; x < w; x++)
                c[t * w + x] = clip_intp2(c[t * w + x] + c[t * w + x - 1], range);
    }
}

// ff_vvc_inv_lfnst_1d (vvc_itx_1d.c:708): v[j] = clip((sum_i u[i] * M[i][j] + 64) >> 7); one lane per output
__global__ __launch_bounds__(64) void lfnst_kernel(int *v, const int *u, int nz, int n_tr_s, int set, int idx, int range)
{
    const int j = threadIdx.x;
    if (j >= n_tr_s)
        return;
    const int8_t *m = n_tr_s > 16 ? d_tab_lfnst_8x8 + (set * 2 + idx - 1) * 16 * 48 : d_tab_lfnst_4x4 + (set * 2 + idx - 1) * 16 * 16;
    unsigned t = 0;
    for (int i = 0; i < nz; i++)
        t += (unsigned)u[i] * (unsigned)(int)m[i * n_tr_s + j];
    v[j] = clip_intp2(((int)t + 64) >> 7, range);
}

// dequant (when asked) + ilfnst_transform (vvc_intra.c:65-127) of one transform block per wave, in place: the scaling process over
// the whole scan window first (what the reference's dequant leaves, :400-417), then the first 8 / 16 levels in 4x4 diagonal scan
// order through ff_vvc_inv_lfnst_1d, scattered into the top-left 4x4 or the 8x8 L-shape (48 outputs), transposed for modes > 34
__global__ __launch_bounds__(256) void lfnst_batch_kernel(const vvc355_lfnst_job *__restrict__ jobs, int n_jobs)
{
    __shared__ int u_all[4][16];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ji = blockIdx.x * 4 + wave;
    if (ji >= n_jobs)
        return;
    const vvc355_lfnst_job job = jobs[ji];
    int *coeffs = (int *)job.coeffs;
    const int lw = job.log2_w, w = 1 << lw, h = 1 << job.log2_h;
    if (job.dequant) {
        Dequant dq;
        dq.setup(1, lw, job.log2_h, job.qp, 0, job.dep_quant, job.bit_depth, job.range, (const uint8_t *)job.scale_matrix, job.log2_matrix_size, job.dc);
        const int rw = job.max_x + 1, n = rw * (job.max_y + 1);
        for (int i = lane; i < n; i += 64) {
            const int y = i / rw, x = i - y * rw;
            const int c = gld<int>(coeffs + (y << lw) + x);
            if (c)
                gst<int>(coeffs + (y << lw) + x, dq.apply(c, x, y));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the gather below reads what other lanes of this wave stored
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    const bool big = w >= 8 && h >= 8;
    const int n_out = big ? 48 : 16;
    const int nz = ((w == 8 && h == 8) || (w == 4 && h == 4)) ? 8 : 16;
    // 6.5.2 up-right diagonal scan of a 4x4 block (ff_vvc_diag_scan_x / _y [2][2]), packed one nibble per position
    const unsigned long long sx = kDiag4X, sy = kDiag4Y;
    if (lane < 16)
        u_all[wave][lane] = lane < nz ? gld<int>(coeffs + w * (int)((sy >> (4 * lane)) & 15) + (int)((sx >> (4 * lane)) & 15)) : 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane < n_out) {
        const int mode = job.pred_mode_intra;
        const int set = mode < 0 ? 1 : d_tab_lfnst_tr_set_index[mode];
        const int8_t *m = big ? d_tab_lfnst_8x8 + (set * 2 + job.lfnst_idx - 1) * 16 * 48 : d_tab_lfnst_4x4 + (set * 2 + job.lfnst_idx - 1) * 16 * 16;
        unsigned t = 0;
        for (int i = 0; i < nz; i++)
            t += (unsigned)u_all[wave][i] * (unsigned)(int)m[i * n_out + lane];
        const int v = clip_intp2(((int)t + 64) >> 7, job.range);
        int x, y;
        const int j = lane;
        if (mode > 34) {            // transposed placement (:86-110)
            if (!big)        { y = j & 3; x = j >> 2; }
            else if (j < 32) { y = j & 7; x = j >> 3; }
            else             { y = (j - 32) & 3; x = 4 + ((j - 32) >> 2); }
        } else {                    // row by row: 8 (4) values in rows 0..3, 4 in rows 4..7 (:111-120)
            if (!big)        { y = j >> 2; x = j & 3; }
            else if (j < 32) { y = j >> 3; x = j & 7; }
            else             { y = 4 + ((j - 32) >> 2); x = (j - 32) & 3; }
        }
        gst<int>(coeffs + y * w + x, v);
    }
}

static bool itx_entry_exists(int trh, int trv, int lw, int lh)
{
    if (lw < 0 || lh < 0 || lw > 6 || lh > 6 || trh < 0 || trh > 2 || trv < 0 || trv > 2 || (lw == 0 && lh == 0))
        return false;
    if (lh == 0) return trv == TX_DCT2 && (lw == 4 || lw == 5 || (lw == 6 && trh == TX_DCT2));
    if (lw == 0) return trh == TX_DCT2 && (lh == 4 || lh == 5 || (lh == 6 && trv == TX_DCT2));
    if (trh != TX_DCT2 && (lw < 2 || lw > 5)) return false;
    if (trv != TX_DCT2 && (lh < 2 || lh > 5)) return false;
    return true;
}

template <int BD, int LW>
static void launch_itx_shape(hipStream_t st, const vvc355_itx_job *jobs_dev, int n_jobs, int log2_h)
{
#define VVC355_ITX_SHAPE(LH)                                                                                          \
    case LH: {                                                                                                        \
        constexpr int TBS = 256 / ((1 << (LW + LH)) / 16);                                                            \
        hipLaunchKernelGGL((itx_shape_kernel<BD, LW, LH>), dim3((n_jobs + TBS - 1) / TBS), dim3(256), 0, st, jobs_dev, n_jobs); \
    } break;
    switch (log2_h) {
    VVC355_ITX_SHAPE(2) VVC355_ITX_SHAPE(3) VVC355_ITX_SHAPE(4) VVC355_ITX_SHAPE(5) VVC355_ITX_SHAPE(6)
    }
#undef VVC355_ITX_SHAPE
}


template <int BD>
static void launch_itx_shape_any(hipStream_t st, const vvc355_itx_job *jobs_dev, int n_jobs, int log2_w, int log2_h)
{
    switch (log2_w) {
    case 2: launch_itx_shape<BD, 2>(st, jobs_dev, n_jobs, log2_h); break;
    case 3: launch_itx_shape<BD, 3>(st, jobs_dev, n_jobs, log2_h); break;
    case 4: launch_itx_shape<BD, 4>(st, jobs_dev, n_jobs, log2_h); break;
    case 5: launch_itx_shape<BD, 5>(st, jobs_dev, n_jobs, log2_h); break;
    case 6: launch_itx_shape<BD, 6>(st, jobs_dev, n_jobs, log2_h); break;
    }
}

// one lane per transform block record: the 48-byte job the transform kernels consume (vvc355_itx_frame_build)
__global__ __launch_bounds__(256) void itx_build_kernel(const vvc355_itx_frame *__restrict__ fp)
{
    const vvc355_itx_frame f = load_uniform(fp);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= f.n_tus)
        return;
    const vvc355_itx_tu t = ((const vvc355_itx_tu *)f.tus)[i];
    vvc355_itx_job j = {};
    j.coeffs = f.coeffs + (uint64_t)t.coeff_off * 4;
    const int c = t.c_idx;
    const uint64_t plane = c == 0 ? f.plane[0] : c == 1 ? f.plane[1] : f.plane[2];
    const int stride = c == 0 ? f.stride[0] : c == 1 ? f.stride[1] : f.stride[2];
    const bool keep = (t.flags & 4) != 0;
    j.dst = keep ? 0 : plane + (uint64_t)t.y0 * stride + ((uint64_t)t.x0 << f.pixel_shift);
    j.dst_stride = stride;
    j.trh = t.tr & 15; j.trv = t.tr >> 4;
    j.log2_w = t.log2_w; j.log2_h = t.log2_h; j.nzw = t.nzw; j.nzh = t.nzh;
    j.range = f.range; j.bd = f.bd;
    j.store_coeffs = keep;
    j.dq_flags = (uint8_t)((t.flags & 1) | (t.flags & 2)); j.dq_qp = t.qp;
    j.log2_matrix_size = 1; j.dc = -1;
    j.mts_flags = (t.flags & 8) ? VVC355_ITX_DERIVE_TYPE : 0; j.tu_flags = f.tu_flags; j.c_idx = t.c_idx;
    ((vvc355_itx_job *)f.jobs)[i] = j;
    if (f.resid_jobs) {
        vvc355_lmcs_resid_job r = {};
        if (t.flags & 64) {
            r.dst = plane + (uint64_t)t.y0 * stride + ((uint64_t)t.x0 << f.pixel_shift);
            r.resid = j.coeffs; r.luma = f.plane[0];
            r.dst_stride = stride; r.luma_stride = f.stride[0];
            r.w = (int16_t)(1 << t.log2_w); r.h = (int16_t)(1 << t.log2_h);
            r.x_vpdu = (int16_t)((t.x0 << (c ? f.hs : 0)) & ~(f.size_y - 1)); r.y_vpdu = (int16_t)((t.y0 << (c ? f.vs : 0)) & ~(f.size_y - 1));
            r.pic_w = (int16_t)f.width; r.pic_h = (int16_t)f.height; r.size_y = f.size_y;
            r.avail_l = (t.flags >> 4) & 1; r.avail_t = (t.flags >> 5) & 1;
            r.joint = 8;
            if (f.scale_table) {          // the unit's entry of vvc355_lmcs_vpdu_scale_pass's table instead of a derivation per block
                const int ux = (f.width + f.size_y - 1) / f.size_y;
                r.luma = f.scale_table + (uint64_t)((r.y_vpdu / f.size_y) * ux + r.x_vpdu / f.size_y) * 2;
                r.joint = 8 | 16;
            }
        }
        ((vvc355_lmcs_resid_job *)f.resid_jobs)[i] = r;
    }
}

} // namespace vvc355

using namespace vvc355;

extern "C" {
extern const uint8_t vvc355_tab_lfnst_tr_set_index[95];

void vvc355_itx_batch(void *stream, int bd, const vvc355_itx_job *jobs_dev, int n_jobs, int max_log2_area)
{
    if (n_jobs <= 0) return;
    hipStream_t st = (hipStream_t)stream;
    VVC355_BD_DISPATCH(bd, {
        if (max_log2_area <= 4)       hipLaunchKernelGGL((itx_kernel<BD, 4, 16>), dim3((n_jobs + 63) / 64), dim3(256), 0, st, jobs_dev, n_jobs);
        else if (max_log2_area <= 6)  hipLaunchKernelGGL((itx_kernel<BD, 16, 64>), dim3((n_jobs + 15) / 16), dim3(256), 0, st, jobs_dev, n_jobs);
        else if (max_log2_area <= 8)  hipLaunchKernelGGL((itx_kernel<BD, 64, 256>), dim3((n_jobs + 3) / 4), dim3(256), 0, st, jobs_dev, n_jobs);
        else if (max_log2_area <= 10) hipLaunchKernelGGL((itx_kernel<BD, 256, 1024>), dim3(n_jobs), dim3(256), 0, st, jobs_dev, n_jobs);
        else                          hipLaunchKernelGGL((itx_kernel<BD, 256, 4096>), dim3(n_jobs), dim3(256), 0, st, jobs_dev, n_jobs);
    });
    HIP_CHECK(hipGetLastError());
}

void vvc355_itx_frame_build(void *stream, const vvc355_itx_frame *frame_dev, const vvc355_itx_frame *frame_host)
{
    if (frame_host->n_tus <= 0) return;
    hipLaunchKernelGGL(vvc355::itx_build_kernel, dim3((frame_host->n_tus + 255) / 256), dim3(256), 0, (hipStream_t)stream, frame_dev);
    HIP_CHECK(hipGetLastError());
}

void vvc355_itx_shape_batch(void *stream, int bd, const vvc355_itx_job *jobs_dev, int n_jobs, int log2_w, int log2_h)
{
    if (n_jobs <= 0) return;
    if (log2_w < 2 || log2_w > 6 || log2_h < 2 || log2_h > 6) {
        vvc355_itx_batch(stream, bd, jobs_dev, n_jobs, log2_w + log2_h);
        return;
    }
    hipStream_t st = (hipStream_t)stream;
    VVC355_BD_DISPATCH(bd, launch_itx_shape_any<BD>(st, jobs_dev, n_jobs, log2_w, log2_h));
    HIP_CHECK(hipGetLastError());
}

void vvc355_dequant_batch(void *stream, const vvc355_dequant_job *jobs_dev, int n_jobs)
{
    if (n_jobs <= 0) return;
    hipLaunchKernelGGL(dequant_kernel, dim3((n_jobs + 15) / 16), dim3(256), 0, (hipStream_t)stream, jobs_dev, n_jobs);
    HIP_CHECK(hipGetLastError());
}

void vvc355_dequant(int *coeffs, int log2_w, int log2_h, int min_x, int min_y, int max_x, int max_y, int qp, int ts,
                    int dep_quant, int bit_depth, int log2_transform_range, const uint8_t *scale_matrix, int log2_matrix_size, int dc)
{
    SlotCall call;
    vvc355_dequant_job job = {};
    job.coeffs = (uint64_t)call.linear(coeffs, (sizeof(int) << (log2_w + log2_h)), true, true);
    if (scale_matrix)
        job.scale_matrix = (uint64_t)call.linear(scale_matrix, (size_t)1 << (2 * log2_matrix_size), true, false);
    job.log2_w = (uint8_t)log2_w; job.log2_h = (uint8_t)log2_h;
    job.min_x = (uint8_t)min_x; job.min_y = (uint8_t)min_y; job.max_x = (uint8_t)max_x; job.max_y = (uint8_t)max_y;
    job.qp = (uint8_t)qp; job.ts = (uint8_t)ts; job.dep_quant = (uint8_t)dep_quant; job.bit_depth = (uint8_t)bit_depth;
    job.range = (uint8_t)log2_transform_range; job.log2_matrix_size = (uint8_t)log2_matrix_size; job.dc = (int16_t)dc;
    vvc355_dequant_batch(call.stream(), call.upload(&job, 1), 1);
}

int vvc355_itx(int trh, int trv, int log2_w, int log2_h, int *coeffs, size_t nzw, size_t nzh,
               intptr_t log2_transform_range, intptr_t bit_depth)
{
    if (!itx_entry_exists(trh, trv, log2_w, log2_h))
        return -1;
    const int n = 1 << (log2_w + log2_h);
    SlotCall call;
    vvc355_itx_job job = {};
    job.coeffs = (uint64_t)call.linear(coeffs, (size_t)n * sizeof(int), true, true);
    job.trh = (uint8_t)trh; job.trv = (uint8_t)trv; job.log2_w = (uint8_t)log2_w; job.log2_h = (uint8_t)log2_h;
    job.nzw = (uint8_t)nzw; job.nzh = (uint8_t)nzh; job.range = (uint8_t)log2_transform_range; job.bd = (uint8_t)bit_depth;
    job.store_coeffs = 1;
    const int kbd = (int)bit_depth == 8 || (int)bit_depth == 10 || (int)bit_depth == 12 ? (int)bit_depth : 10;
    if (log2_w >= 2 && log2_h >= 2)
        vvc355_itx_shape_batch(call.stream(), kbd, call.upload(&job, 1), 1, log2_w, log2_h);
    else
        vvc355_itx_batch(call.stream(), kbd, call.upload(&job, 1), 1, log2_w + log2_h);
    return 0;
}

void vvc355_inv_lfnst_1d(int *v, const int *u, int no_zero_size, int n_tr_s, int pred_mode_intra, int lfnst_idx,
                         int log2_transform_range)
{
    const int set = pred_mode_intra < 0 ? 1 : vvc355_tab_lfnst_tr_set_index[pred_mode_intra];
    SlotCall call;
    int *dv = (int *)call.linear(v, (size_t)n_tr_s * sizeof(int), false, true);
    const int *du = (const int *)call.linear(u, (size_t)no_zero_size * sizeof(int), true, false);
    hipLaunchKernelGGL(lfnst_kernel, dim3(1), dim3(64), 0, call.stream(), dv, du, no_zero_size, n_tr_s, set, lfnst_idx, log2_transform_range);
    HIP_CHECK(hipGetLastError());
}

void vvc355_lfnst_batch(void *stream, const vvc355_lfnst_job *jobs_dev, int n_jobs)
{
    if (n_jobs <= 0) return;
    hipLaunchKernelGGL(lfnst_batch_kernel, dim3((n_jobs + 3) / 4), dim3(256), 0, (hipStream_t)stream, jobs_dev, n_jobs);
    HIP_CHECK(hipGetLastError());
}

int vvc355_ilfnst_transform(int *coeffs, int w, int h, int pred_mode_intra, int lfnst_idx, int log2_transform_range)
{
    SlotCall call;
    vvc355_lfnst_job job = {};
    job.coeffs = (uint64_t)call.linear(coeffs, (size_t)w * h * sizeof(int), true, true);
    int lw = 0, lh = 0;
    while ((1 << lw) < w) lw++;
    while ((1 << lh) < h) lh++;
    job.log2_w = (uint8_t)lw; job.log2_h = (uint8_t)lh; job.range = (uint8_t)log2_transform_range;
    job.pred_mode_intra = (int8_t)pred_mode_intra; job.lfnst_idx = (uint8_t)lfnst_idx;
    vvc355_lfnst_batch(call.stream(), call.upload(&job, 1), 1);
    return (w >= 8 && h >= 8) ? 8 : 4;
}

int vvc355_derive_transform_type(int tu_flags, int mts_idx, int lfnst_idx, int c_idx, int w, int h)
{
    return derive_tr_type(tu_flags, mts_idx, lfnst_idx, c_idx, w, h);
}

static void slot_residual(int bd, int mode, uint8_t *dst, int *res, int width, int height, ptrdiff_t stride, int c_sign, int shift)
{
    if (width <= 0 || height <= 0) return;
    const int px = bd > 8 ? 2 : 1;
    SlotCall call;
    vvc355_blend_job job = {};
    if (mode != 2) {
        const Staged d = call.rect(dst, stride, 0, width * px, 0, height, true, true);
        job.dst = (uint64_t)d.dev; job.dst_stride = (int32_t)d.pitch;
    }
    job.src0 = (uint64_t)call.linear(res, (size_t)width * height * sizeof(int), true, mode == 2);
    job.w = (int16_t)width; job.h = (int16_t)height; job.mode = (int16_t)mode; job.w0 = (int16_t)c_sign; job.denom = (int16_t)shift;
    const vvc355_blend_job *jd = call.upload(&job, 1);
    const int gx = (width * height + 1023) / 1024;
    VVC355_BD_DISPATCH(bd, hipLaunchKernelGGL((residual_kernel<BD>), dim3(gx, 1), dim3(256), 0, call.stream(), jd));
    HIP_CHECK(hipGetLastError());
}

void vvc355_add_residual(int bd, uint8_t *dst, const int *res, int width, int height, ptrdiff_t stride)
{
    slot_residual(bd, 0, dst, const_cast<int *>(res), width, height, stride, 0, 0);
}

void vvc355_add_residual_joint(int bd, uint8_t *dst, const int *res, int width, int height, ptrdiff_t stride, int c_sign, int shift)
{
    slot_residual(bd, 1, dst, const_cast<int *>(res), width, height, stride, c_sign, shift);
}

void vvc355_pred_residual_joint(int *buf, int width, int height, int c_sign, int shift)
{
    slot_residual(10, 2, nullptr, buf, width, height, 0, c_sign, shift);
}

void vvc355_transform_bdpcm(int *coeffs, int width, int height, int vertical, int log2_transform_range)
{
    if (width <= 0 || height <= 0 || width > 128 || height > 128) return;
    SlotCall call;
    vvc355_blend_job job = {};
    job.dst = (uint64_t)call.linear(coeffs, (size_t)width * height * sizeof(int), true, true);
    job.w = (int16_t)width; job.h = (int16_t)height; job.mode = (int16_t)!!vertical; job.denom = (int16_t)log2_transform_range;
    hipLaunchKernelGGL(bdpcm_kernel, dim3(1), dim3(128), 0, call.stream(), call.upload(&job, 1));
    HIP_CHECK(hipGetLastError());
}

} // extern "C"
