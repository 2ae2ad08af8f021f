// Fused motion-compensated prediction for gfx950: for every block of at most 16x16 samples, interpolate from one or two
// reference pictures (8-tap luma / 4-tap chroma, any of the copy / h / v / hv cases) and blend (avg, w_avg) or round
// (put_uni, put_uni_w) straight to pixels — the 14-bit intermediates never go to HBM.
//
// This is the batched form of what the reference does per prediction block in luma_mc_bi / chroma_mc_bi / *_mc_uni
// (libavcodec/vvc/vvc_inter.c:222-460): put[..] x2 + avg / w_avg, or put_uni / put_uni_w.  Arithmetic follows
// libavcodec/h26x/h2656_inter_template.c:29-577 and libavcodec/vvc/vvc_inter_template.c:25-58 exactly (int16 narrowing of
// the horizontal pass and of the bi-prediction operands included).  Larger prediction blocks are cut into <= 16x16 tiles
// by the job builder; the interpolation is separable per output tile, so the result does not depend on the tiling.
//
// Mapping: one wave per block, four blocks per workgroup, no workgroup barrier.  The job descriptor is wave-uniform.
// Both reference windows are requested before anything waits on them, then go to LDS as uint16.  Each lane produces two
// horizontally adjacent outputs per step from aligned sample pairs with v_dot2c_i32_i16 (9 dot products for two 8-tap
// outputs, 5 for two 4-tap outputs) and writes the intermediate transposed, so that the vertical pass again reads aligned
// pairs.  The kernel is VALU-issue bound (rocprofv3: > 80 % of the per-SIMD VALU slots), hence the 4-tap specialisation.
#include "common.hpp"
#include "runtime.hpp"
#include "../../include/vvc_mi355.h"

namespace vvc355 {

typedef short v2s __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int dot2(uint32_t a, uint32_t b, int acc)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(v2s, a), __builtin_bit_cast(v2s, b), acc, false);
}
__device__ __forceinline__ uint32_t pack16(int lo, int hi) { return (uint32_t)(lo & 0xffff) | ((uint32_t)hi << 16); }
__device__ __forceinline__ int tap_of(uint32_t lo, uint32_t hi, int k)      // signed byte k of the 8-byte tap vector
{
    const uint32_t v = k < 4 ? lo : hi;
    return (int)(int8_t)(v >> ((k & 3) * 8));
}

static constexpr int kWinW = 24;      // LDS source window: up to 16 + 8 columns, even pitch
static constexpr int kWinH = 23;
static constexpr int kTmpP = 24;      // transposed intermediate: [column][row], up to 23 rows (+1 read-only slack), even pitch

// Two adjacent outputs of an NTAP-tap filter from aligned sample pairs d[m] = (p[2m], p[2m+1]):
// out0 = sum f[k] p[k], out1 = sum f[k] p[k+1].
template <int NTAP> struct Taps {
    static constexpr int NE = NTAP / 2, NO = NTAP / 2 + 1;
    uint32_t e[NE];     // (f0,f1) (f2,f3) ...
    uint32_t o[NO];     // (0,f0) (f1,f2) ... (f_last,0)
    __device__ __forceinline__ void set(uint32_t lo, uint32_t hi)
    {
        int f[NTAP];
#pragma unroll
        for (int k = 0; k < NTAP; k++) f[k] = tap_of(lo, hi, k);
#pragma unroll
        for (int m = 0; m < NE; m++) e[m] = pack16(f[2 * m], f[2 * m + 1]);
        o[0] = pack16(0, f[0]);
#pragma unroll
        for (int m = 1; m < NE; m++) o[m] = pack16(f[2 * m - 1], f[2 * m]);
        o[NE] = pack16(f[NTAP - 1], 0);
    }
    __device__ __forceinline__ void apply(const uint32_t (&d)[NO], int &out0, int &out1) const
    {
        int a = 0, b = 0;
#pragma unroll
        for (int m = 0; m < NE; m++) a = dot2(d[m], e[m], a);
#pragma unroll
        for (int m = 0; m < NO; m++) b = dot2(d[m], o[m], b);
        out0 = a; out1 = b;
    }
};

// Window staging is split in two so that the loads of BOTH references are in flight before anything waits on them.
// Window index (r, c) <-> picture sample (r - LEAD, c - LEAD) when that direction is filtered, else (r, c - LEAD):
// LEAD = 3 (8 taps) or 1 (4 taps).  Lane -> column c = lane & 31, rows (lane >> 5) + 2 * it.  Lanes / rows the reference
// would not read load the block's own origin sample instead (always valid) and are zeroed by the select.
template <int BD, int NTAP>
__device__ __forceinline__ void fetch_window(const uint8_t *src, int src_stride, int lw, int h, bool hfrac, bool vfrac,
                                             int lane, uint16_t (&v)[(16 + NTAP) / 2])
{
    using px_t = typename Px<BD>::type;
    constexpr int LEAD = NTAP == 8 ? 3 : 1, NIT = (16 + NTAP) / 2;
    const int w = 1 << lw;
    const int c = lane & 31, col = c - LEAD;
    const bool col_ok = hfrac ? (c < w + NTAP - 1) : (col >= 0 && col < w);
    const int r_hi = vfrac ? h + NTAP - 1 : h;                       // window rows the filter reads
    const px_t *base = (const px_t *)src;
    const px_t *p = (const px_t *)(src + (ptrdiff_t)((lane >> 5) - (vfrac ? LEAD : 0)) * src_stride) + col;
    const ptrdiff_t step = (ptrdiff_t)src_stride * 2 / (ptrdiff_t)sizeof(px_t);
#pragma unroll
    for (int it = 0; it < NIT; it++) {
        const int r = (lane >> 5) + 2 * it;
        const bool ok = col_ok && r < r_hi;
        const px_t *pp = ok ? p : base;
        const uint16_t s = (uint16_t)gld<px_t>(pp);
        v[it] = ok ? s : (uint16_t)0;
        p += step;
    }
}

template <int NTAP>
__device__ __forceinline__ void store_window(uint16_t *win, int lane, const uint16_t (&v)[(16 + NTAP) / 2])
{
    constexpr int NIT = (16 + NTAP) / 2;
    const int c = lane & 31;
    if (c < kWinW) {
        uint16_t *q = win + (lane >> 5) * kWinW + c;
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            if ((lane >> 5) + 2 * it < kWinH)
                q[it * 2 * kWinW] = v[it];
        }
    }
}

// One reference of one block, from its LDS window -> up to 4 intermediate values per lane (two row pairs of one column),
// 14-bit scaled ints.  Result layout: column x = lane & 15, row pair yp = (lane >> 4) + 4 * i (i = 0, 1), rows 2yp, 2yp + 1.
template <int BD, int NTAP>
__device__ __forceinline__ void interp_block(int lw, int h, bool hfrac, bool vfrac, uint32_t hf_lo, uint32_t hf_hi,
                                             uint32_t vf_lo, uint32_t vf_hi, const uint16_t *win, int16_t *tmpT, int lane, int (&val)[4])
{
    constexpr int LEAD = NTAP == 8 ? 3 : 1, NO = NTAP / 2 + 1;
    const int w = 1 << lw;
    const int sh = vfrac ? h + NTAP - 1 : h;

    // ---- horizontal pass -> tmpT[x][r] (int16), r over the sh window rows
    if (hfrac) {
        Taps<NTAP> t;
        t.set(hf_lo, hf_hi);
        const int lhw = lw - 1;                         // log2 of the number of output pairs per row
        const int n = sh << lhw;
        for (int i = lane; i < n; i += 64) {
            const int r = i >> lhw, xp = i & ((1 << lhw) - 1);                  // outputs x = 2*xp, 2*xp + 1
            const uint32_t *d = (const uint32_t *)(win + r * kWinW + 2 * xp);   // tap 0 of output x sits at window index x: aligned
            uint32_t dd[NO];
#pragma unroll
            for (int m = 0; m < NO; m++) dd[m] = d[m];
            int o0, o1;
            t.apply(dd, o0, o1);
            tmpT[(2 * xp) * kTmpP + r] = (int16_t)(o0 >> (BD - 8));
            tmpT[(2 * xp + 1) * kTmpP + r] = (int16_t)(o1 >> (BD - 8));
        }
    } else {
        const int n = sh << lw;
        for (int i = lane; i < n; i += 64) {
            const int r = i >> lw, x = i & (w - 1);
            tmpT[x * kTmpP + r] = (int16_t)win[r * kWinW + x + LEAD];           // raw samples
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- vertical pass: lane -> column x, two row pairs
    const int x = lane & 15;
    Taps<NTAP> t;
    t.set(vf_lo, vf_hi);
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int yp = (lane >> 4) + 4 * i;
        int o0 = 0, o1 = 0;
        if (x < w && 2 * yp < h) {
            if (vfrac) {
                const uint32_t *d = (const uint32_t *)(tmpT + x * kTmpP + 2 * yp);   // tap 0 of output y sits at window row y: aligned
                uint32_t dd[NO];
#pragma unroll
                for (int m = 0; m < NO; m++) dd[m] = d[m];
                t.apply(dd, o0, o1);
                // hv: second stage >> 6 on the int16 intermediates; v only: first stage on raw samples >> (bd - 8)
                const int sh2 = hfrac ? 6 : BD - 8;
                o0 >>= sh2; o1 >>= sh2;
            } else {
                const uint32_t d = *(const uint32_t *)(tmpT + x * kTmpP + 2 * yp);
                o0 = (int16_t)(d & 0xffff);
                o1 = (int16_t)(d >> 16);
                if (!hfrac) { o0 <<= 14 - BD; o1 <<= 14 - BD; }                      // integer position: sample << (14 - bd)
            }
        }
        val[2 * i] = o0;
        val[2 * i + 1] = o1;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// both references of one block: windows requested together, then interpolated one after the other
template <int BD, int NTAP>
__device__ __forceinline__ void predict_refs(const vvc355_pred_job *job, int lw, int h, int mode, int frac,
                                             uint16_t (*win)[kWinH * kWinW], int16_t *tmpT, int lane, int (&v0)[4], int (&v1)[4])
{
    const uint32_t *taps = (const uint32_t *)job->hf0;  // hf0, vf0, hf1, vf1: 8 dwords (4-tap filters use the low dword)
    {
        uint16_t r0[(16 + NTAP) / 2], r1[(16 + NTAP) / 2];
        fetch_window<BD, NTAP>((const uint8_t *)job->src0, job->src0_stride, lw, h, frac & 1, frac & 2, lane, r0);
        if (mode < 2)
            fetch_window<BD, NTAP>((const uint8_t *)job->src1, job->src1_stride, lw, h, frac & 4, frac & 8, lane, r1);
        store_window<NTAP>(win[0], lane, r0);
        if (mode < 2)
            store_window<NTAP>(win[1], lane, r1);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    interp_block<BD, NTAP>(lw, h, frac & 1, frac & 2, taps[0], taps[1], taps[2], taps[3], win[0], tmpT, lane, v0);
    if (mode < 2)
        interp_block<BD, NTAP>(lw, h, frac & 4, frac & 8, taps[4], taps[5], taps[6], taps[7], win[1], tmpT, lane, v1);
}

template <int BD>
__global__ __launch_bounds__(256) void pred_fused_kernel(const vvc355_pred_job *__restrict__ jobs, int n_jobs)
{
    __shared__ __attribute__((aligned(16))) uint16_t win_all[4][2][kWinH * kWinW];
    __shared__ __attribute__((aligned(16))) int16_t tmp_all[4][16 * kTmpP];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int ji = blockIdx.x * 4 + wave;
    if (ji >= n_jobs)
        return;
    const vvc355_pred_job *job = jobs + ji;              // wave-uniform address
    const int w = job->w, h = job->h, mode = job->mode, frac = job->frac;
    const int lw = 31 - __builtin_clz(w);
    int v0[4], v1[4] = { 0, 0, 0, 0 };
    if (job->chroma)
        predict_refs<BD, 4>(job, lw, h, mode, frac, win_all[wave], tmp_all[wave], lane, v0, v1);
    else
        predict_refs<BD, 8>(job, lw, h, mode, frac, win_all[wave], tmp_all[wave], lane, v0, v1);
    if (mode < 2) {
        // the reference carries bi-prediction operands in int16 planes (put[..] narrows on store)
#pragma unroll
        for (int i = 0; i < 4; i++) { v0[i] = (int16_t)v0[i]; v1[i] = (int16_t)v1[i]; }
    }

    const int denom = job->denom, w0 = job->w0, w1 = job->w1, o0 = job->o0, o1 = job->o1;
    int shift, off;
    if (mode == 0)      { shift = max(3, 15 - BD); off = 1 << (shift - 1); }                                        // avg
    else if (mode == 1) { shift = denom + max(3, 15 - BD); off = (((o0 + o1) << (BD - 8)) + 1) << (shift - 1); }    // w_avg
    else if (mode == 2) { shift = 14 - BD; off = 1 << (shift - 1); }                                                // put_uni
    else                { shift = denom + 14 - BD; off = 1 << (shift - 1); }                                        // put_uni_w
    const int x = lane & 15;
    uint8_t *dst = (uint8_t *)job->dst;
    const int dst_stride = job->dst_stride;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int y = 2 * ((lane >> 4) + 4 * (i >> 1)) + (i & 1);
        if (x >= w || y >= h)
            continue;
        int p;
        if (mode == 0)      p = (v0[i] + v1[i] + off) >> shift;
        else if (mode == 1) p = (v0[i] * w0 + v1[i] * w1 + off) >> shift;
        else if (mode == 2) p = (frac & 3) ? (v0[i] + off) >> shift : v0[i] >> (14 - BD);      // integer position = plain copy
        else                p = ((v0[i] * w0 + off) >> shift) + o0 * (1 << (BD - 8));
        st_px<BD>(dst + (ptrdiff_t)y * dst_stride, x, clip_px<BD>(p));
    }
}

} // namespace vvc355

extern "C" void vvc355_pred_fused_batch(void *stream, int bd, const vvc355_pred_job *jobs_dev, int n_jobs)
{
    using namespace vvc355;
    if (n_jobs <= 0) return;
    VVC355_BD_DISPATCH(bd, hipLaunchKernelGGL((pred_fused_kernel<BD>), dim3((n_jobs + 3) / 4), dim3(256), 0, (hipStream_t)stream, jobs_dev, n_jobs));
    HIP_CHECK(hipGetLastError());
}
