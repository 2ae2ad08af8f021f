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
// Mapping: one wave per block, four blocks per workgroup, no workgroup barrier.  The job descriptor is wave-uniform and is
// read with scalar loads.  The source window goes to LDS as uint16; each lane produces two horizontally adjacent outputs
// per step from aligned sample pairs with v_dot2c_i32_i16 (9 dot products for 16 taps) and writes the intermediate
// transposed, so that the vertical pass again reads aligned pairs.
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

// two adjacent outputs of an 8-tap filter from five aligned sample pairs d[0..4] = (p[0],p[1]) .. (p[8],p[9]):
// out0 = sum f[k] p[k], out1 = sum f[k] p[k+1]
struct Taps {
    uint32_t e[4];     // (f0,f1) (f2,f3) (f4,f5) (f6,f7)
    uint32_t o[5];     // (0,f0) (f1,f2) (f3,f4) (f5,f6) (f7,0)
    // lo/hi = the 8 tap bytes as stored in the job; chroma: 4 taps in `lo`, applied at positions -1..2 (indices 2..5)
    __device__ __forceinline__ void set(uint32_t lo, uint32_t hi, bool chroma)
    {
        int f[8];
#pragma unroll
        for (int k = 0; k < 8; k++)
            f[k] = chroma ? ((k >= 2 && k < 6) ? tap_of(lo, 0, k - 2) : 0) : tap_of(lo, hi, k);
        e[0] = pack16(f[0], f[1]); e[1] = pack16(f[2], f[3]); e[2] = pack16(f[4], f[5]); e[3] = pack16(f[6], f[7]);
        o[0] = pack16(0, f[0]); o[1] = pack16(f[1], f[2]); o[2] = pack16(f[3], f[4]); o[3] = pack16(f[5], f[6]); o[4] = pack16(f[7], 0);
    }
    __device__ __forceinline__ void apply(const uint32_t (&d)[5], int &out0, int &out1) const
    {
        out0 = dot2(d[3], e[3], dot2(d[2], e[2], dot2(d[1], e[1], dot2(d[0], e[0], 0))));
        out1 = dot2(d[4], o[4], dot2(d[3], o[3], dot2(d[2], o[2], dot2(d[1], o[1], dot2(d[0], o[0], 0)))));
    }
};

// One reference of one block -> up to 4 intermediate values per lane (two row pairs of one column), 14-bit scaled ints.
// Lane layout of the result: column x = lane & 15, row pair yp = (lane >> 4) + 4 * i  (i = 0, 1), rows 2*yp, 2*yp + 1.
// All arguments except `lane` are wave-uniform.
// Window staging is split in two so that the loads of BOTH references are in flight before anything waits on them:
// fetch_window issues 12 unconditional loads per lane (lane -> column c = lane & 31, rows (lane >> 5) + 2*it); lanes / rows the
// reference would not read load the block's own origin sample instead (always valid) and are zeroed by the select.
template <int BD>
__device__ __forceinline__ void fetch_window(const uint8_t *src, int src_stride, int lw, int h, bool chroma, bool hfrac, bool vfrac,
                                             int lane, uint16_t (&v)[12])
{
    using px_t = typename Px<BD>::type;
    const int w = 1 << lw;
    const int lead = chroma ? 1 : 3, trail = chroma ? 2 : 4;
    const int row0 = vfrac ? -3 : 0;
    const int c = lane & 31, col = c - 3;
    const bool col_ok = hfrac ? (col >= -lead && col < w + trail) : (col >= 0 && col < w);
    const int r_lo = vfrac ? 3 - lead : 0, r_hi = vfrac ? 3 + h + trail : h;         // window rows the filter reads
    const px_t *base = (const px_t *)src;
    const px_t *p = (const px_t *)(src + (ptrdiff_t)((lane >> 5) + row0) * src_stride) + col;
    const ptrdiff_t step = (ptrdiff_t)src_stride * 2 / (ptrdiff_t)sizeof(px_t);
#pragma unroll
    for (int it = 0; it < 12; it++) {
        const int r = (lane >> 5) + 2 * it;
        const bool ok = col_ok && r >= r_lo && r < r_hi;
        const px_t *pp = ok ? p : base;
        const uint16_t s = (uint16_t)*pp;
        v[it] = ok ? s : (uint16_t)0;
        p += step;
    }
}

__device__ __forceinline__ void store_window(uint16_t *win, int lane, const uint16_t (&v)[12])
{
    const int c = lane & 31;
    if (c < kWinW) {
        uint16_t *q = win + (lane >> 5) * kWinW + c;
#pragma unroll
        for (int it = 0; it < 12; it++) {
            if ((lane >> 5) + 2 * it < kWinH)
                q[it * 2 * kWinW] = v[it];
        }
    }
}

template <int BD>
__device__ __forceinline__ void interp_block(int lw, int h, bool chroma, bool hfrac, bool vfrac,
                                             uint32_t hf_lo, uint32_t hf_hi, uint32_t vf_lo, uint32_t vf_hi,
                                             const uint16_t *win, int16_t *tmpT, int lane, int (&val)[4])
{
    const int w = 1 << lw;
    const int sh = vfrac ? h + 7 : h;                  // window rows -3 .. h+3 (or 0 .. h-1); columns -3 .. w+4 (index = col + 3)

    // ---- horizontal pass -> tmpT[x][r] (int16), r over the sh window rows
    if (hfrac) {
        Taps t;
        t.set(hf_lo, hf_hi, chroma);
        const int lhw = lw - 1;                         // log2 of the number of output pairs per row
        const int n = sh << lhw;
        for (int i = lane; i < n; i += 64) {
            const int r = i >> lhw, xp = i & ((1 << lhw) - 1);                  // outputs x = 2*xp, 2*xp + 1
            const uint32_t *d = (const uint32_t *)(win + r * kWinW + 2 * xp);   // window index of column x-3 is x: even -> aligned
            const uint32_t dd[5] = { d[0], d[1], d[2], d[3], d[4] };
            int o0, o1;
            t.apply(dd, o0, o1);
            tmpT[(2 * xp) * kTmpP + r] = (int16_t)(o0 >> (BD - 8));
            tmpT[(2 * xp + 1) * kTmpP + r] = (int16_t)(o1 >> (BD - 8));
        }
    } else {
        const int n = sh << lw;
        for (int i = lane; i < n; i += 64) {
            const int r = i >> lw, x = i & (w - 1);
            tmpT[x * kTmpP + r] = (int16_t)win[r * kWinW + x + 3];              // raw samples
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- vertical pass: lane -> column x, two row pairs
    const int x = lane & 15;
    Taps t;
    t.set(vf_lo, vf_hi, chroma);
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int yp = (lane >> 4) + 4 * i;
        int o0 = 0, o1 = 0;
        if (x < w && 2 * yp < h) {
            if (vfrac) {
                const uint32_t *d = (const uint32_t *)(tmpT + x * kTmpP + 2 * yp);   // row y-3 is window row y: even -> aligned
                const uint32_t dd[5] = { d[0], d[1], d[2], d[3], d[4] };
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

template <int BD>
__global__ __launch_bounds__(256) void pred_fused_kernel(const vvc355_pred_job *__restrict__ jobs, int n_jobs)
{
    __shared__ __attribute__((aligned(16))) uint16_t win_all[4][2][kWinH * kWinW];
    __shared__ __attribute__((aligned(16))) int16_t tmp_all[4][16 * kTmpP];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int ji = blockIdx.x * 4 + wave;
    if (ji >= n_jobs)
        return;
    const vvc355_pred_job *job = jobs + ji;              // wave-uniform address: the fields below are scalar loads
    const uint32_t *taps = (const uint32_t *)job->hf0;  // hf0, vf0, hf1, vf1: 8 dwords
    const int w = job->w, h = job->h, mode = job->mode, frac = job->frac;
    const int lw = 31 - __builtin_clz(w);
    const bool chroma = job->chroma;
    int v0[4], v1[4] = { 0, 0, 0, 0 };
    {
        // both windows are requested before either is waited for
        uint16_t r0[12], r1[12];
        fetch_window<BD>((const uint8_t *)job->src0, job->src0_stride, lw, h, chroma, frac & 1, frac & 2, lane, r0);
        if (mode < 2)
            fetch_window<BD>((const uint8_t *)job->src1, job->src1_stride, lw, h, chroma, frac & 4, frac & 8, lane, r1);
        store_window(win_all[wave][0], lane, r0);
        if (mode < 2)
            store_window(win_all[wave][1], lane, r1);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    interp_block<BD>(lw, h, chroma, frac & 1, frac & 2, taps[0], taps[1], taps[2], taps[3], win_all[wave][0], tmp_all[wave], lane, v0);
    if (mode < 2) {
        interp_block<BD>(lw, h, chroma, frac & 4, frac & 8, taps[4], taps[5], taps[6], taps[7], win_all[wave][1], tmp_all[wave], lane, v1);
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
