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
#include <type_traits>
#include "common.hpp"
#include "runtime.hpp"
#include "../../include/vvc_mi355.h"

namespace vvc355 {

#define VVC355_TABLE(type, name, count) __device__ static const type d_tab_##name[count]
#include "tables.inc"
#undef VVC355_TABLE

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
// SPLIT: the block is two 8-wide blocks of two planes side by side (outputs 0..7 | 8..15); the second plane's window starts
// at window column 12 instead of 8, i.e. its outputs read 4 columns further right.
template <int BD, int NTAP, bool SPLIT = false>
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
            const uint32_t *d = (const uint32_t *)(win + r * kWinW + 2 * xp + (SPLIT && xp >= 4 ? 4 : 0));   // tap 0 of output x sits at window index x: aligned
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
            tmpT[x * kTmpP + r] = (int16_t)win[r * kWinW + x + LEAD + (SPLIT && x >= 8 ? 4 : 0)];           // raw samples
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
    const vvc355_pred_job job_copy = load_uniform(jobs + ji);           // wave-uniform address: scalar dword loads, fields unpacked on the SALU
    const vvc355_pred_job *job = &job_copy;
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
        gst_at<typename Px<BD>::type>(dst, (uint32_t)(__mul24(y, dst_stride) + x * (int)sizeof(typename Px<BD>::type)), (typename Px<BD>::type)clip_px<BD>(p));
    }
}

// ------------------------------------------------------------------------------------------------ regular bi-prediction
//
// One wave per sub-block of at most 16x16: what pred_regular_blk (vvc_inter.c:772-822) does around the slots, on device.
// Every read of a reference plane goes through clamped coordinates (edge emulation, vvc_inter.c:33-110).

struct ClampRect { int x0, y0, x1, y1; };               // inclusive, in samples of the component

// NIT row pairs of a window whose index (0, 0) is plane sample (wx0, wy0): lane -> column lane & 31, rows (lane >> 5) + 2 it
template <int BD, int NIT>
__device__ __forceinline__ void fetch_clamped(const uint8_t *plane, int stride, const ClampRect &rc, int wx0, int wy0, int lane,
                                              uint16_t (&v)[NIT])
{
    using px_t = typename Px<BD>::type;
    const int xa = clip3(wx0 + (lane & 31), rc.x0, rc.x1) * (int)sizeof(px_t);
#pragma unroll
    for (int it = 0; it < NIT; it++) {
        const int ya = clip3(wy0 + (lane >> 5) + 2 * it, rc.y0, rc.y1);
        v[it] = (uint16_t)gld_at<px_t>(plane, (uint32_t)(__mul24(ya, stride) + xa));      // clamped: never negative
    }
}

template <int NIT>
__device__ __forceinline__ void store_rows(uint16_t *win, int lane, const uint16_t (&v)[NIT])
{
    const int c = lane & 31;
    if (c < kWinW) {
        uint16_t *q = win + (lane >> 5) * kWinW + c;
#pragma unroll
        for (int it = 0; it < NIT; it++)
            if ((lane >> 5) + 2 * it < kWinH)
                q[it * 2 * kWinW] = v[it];
    }
}

// Unclamped form for windows that lie inside their readable rectangle (the common case): a lane moves 4 consecutive samples
// per step (one 8-byte load, one ds_write_b64), kWinW / 4 = 6 vectors per row: 3 steps cover a 23-row window, instead of 12
// one-sample steps.  fetch_vec4 only issues the loads (both references' loads go out before anything waits), put_vec4 stores.
template <int BD, int NV>
__device__ __forceinline__ void fetch_vec4(const uint8_t *plane, int stride, int wx0, int wy0, int nrows, int lane, uint2 (&v)[NV])
{
    using px_t = typename Px<BD>::type;
    const uint8_t *org = plane + row_off(wy0, stride) + wx0 * (int)sizeof(px_t);
#pragma unroll
    for (int it = 0; it < NV; it++) {
        const int id = lane + 64 * it, r = min(id / 6, nrows - 1), k = id - (id / 6) * 6;    // rows past the window re-read its last row
        const uint32_t p = (uint32_t)(__mul24(r, stride) + k * 4 * (int)sizeof(px_t));       // org is wave-uniform
        if (BD > 8) {
            v[it] = gld_at<uint2>(org, p);
        } else {
            const uint32_t q = gld_at<uint32_t>(org, p);
            v[it] = make_uint2(__builtin_amdgcn_perm(0, q, 0x0c010c00u), __builtin_amdgcn_perm(0, q, 0x0c030c02u));
        }
    }
}
template <int NV>
__device__ __forceinline__ void put_vec4(uint16_t *win, int nrows, int lane, const uint2 (&v)[NV])
{
#pragma unroll
    for (int it = 0; it < NV; it++) {
        const int id = lane + 64 * it, r = id / 6, k = id - r * 6;
        if (r < nrows)
            *(uint2 *)(win + r * kWinW + 4 * k) = v[it];
    }
}
__device__ __forceinline__ bool rect_holds(const ClampRect &rc, int wx0, int wy0, int nrows)
{
    return wx0 >= rc.x0 && wx0 + kWinW - 1 <= rc.x1 && wy0 >= rc.y0 && wy0 + nrows - 1 <= rc.y1;
}

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
        v += __shfl_xor(v, m, 64);
    return v;
}

static constexpr int kBilP = 20;        // DMVR bilinear plane pitch: (16 + 4) columns
static constexpr int kGs = 18;          // BDOF planes: block + one-sample ring

// Per-wave LDS, overlaid by phase: DMVR = win + bil, interpolation = win + tmpT, BDOF = smp + grad (each phase ends with a
// wave barrier before the next one writes).  5.5 KB per wave keeps 28 waves on a CU.
struct BipredLds {
    union {
        struct {
            uint16_t win[2][kWinH * kWinW];
            union {
                int16_t tmpT[16 * kTmpP];
                int16_t bil[2][2][kBilP * kBilP + 4];   // DMVR search planes [ref][natural | shifted by one sample]; the shifted
                                                         // copy keeps sample pairs 4-byte aligned for odd search offsets
            };
        };
        struct {
            int16_t smp[2][kGs * kGs];                   // the two predictions + fetched ring
            int16_t grad[5][kGs * kGs];                  // BDOF planes: D, TH, TV, GHD, GVD (see bdof_wave)
        };
    };
    int sad[28];
};

// vvc_inter.c:642-681
__device__ __forceinline__ int parametric_mv_refine(int sad_minus, int sad_center, int sad_plus)
{
    int denom = ((sad_minus + sad_plus) - (sad_center << 1)) << 3;
    if (!denom)
        return 0;
    if (sad_minus == sad_center)
        return -8;
    if (sad_plus == sad_center)
        return 8;
    int num = (sad_minus - sad_plus) * 16, quotient = 0;
    const bool neg = num < 0;
    if (neg)
        num = -num;
#pragma unroll
    for (int counter = 0; counter < 3; counter++) {
        quotient <<= 1;
        if (num >= denom) {
            num -= denom;
            quotient++;
        }
        denom >>= 1;
    }
    return neg ? -quotient : quotient;
}

// dmvr_mv_refine (vvc_inter.c:685-748): refines mv in place, may clear bdof
template <int BD>
__device__ __forceinline__ void dmvr_refine(const vvc355_bipred_job *job, BipredLds &L, int lane, int (&mv)[4], int &bdof,
                                            int &min_sad_out, int &searched)
{
    const int w = job->w, h = job->h, pw = w + 4, ph = h + 4;
    const ClampRect pic = { 0, 0, job->pic_w - 1, job->pic_h - 1 };
    {
        // (pw + 1) x (ph + 1) integer samples around each reference block, two rows back (emulated_edge_bilinear :90-110)
        const int ax = job->x + (mv[0] >> 4) - 2, ay = job->y + (mv[1] >> 4) - 2, bx = job->x + (mv[2] >> 4) - 2, by = job->y + (mv[3] >> 4) - 2;
        if (rect_holds(pic, ax, ay, ph + 1) && rect_holds(pic, bx, by, ph + 1)) {
            uint2 q0[2], q1[2];                              // 21 rows x 6 vectors = 126 <= 128
            fetch_vec4<BD, 2>((const uint8_t *)job->ref0, job->ref0_stride, ax, ay, ph + 1, lane, q0);
            fetch_vec4<BD, 2>((const uint8_t *)job->ref1, job->ref1_stride, bx, by, ph + 1, lane, q1);
            put_vec4<2>(L.win[0], ph + 1, lane, q0);
            put_vec4<2>(L.win[1], ph + 1, lane, q1);
        } else {
            uint16_t r0[11], r1[11];
            fetch_clamped<BD, 11>((const uint8_t *)job->ref0, job->ref0_stride, pic, ax, ay, lane, r0);
            fetch_clamped<BD, 11>((const uint8_t *)job->ref1, job->ref1_stride, pic, bx, by, lane, r1);
            store_rows<11>(L.win[0], lane, r0);
            store_rows<11>(L.win[1], lane, r1);
        }
        wave_sync();
    }
    // inter.dmvr[!!my][!!mx] (vvc_inter_template.c:324-413).  Lane -> (reference, pair of adjacent columns, segment of rows):
    // the horizontal stage of two outputs is two packed dot products on aligned sample pairs, and walking down the rows lets
    // the vertical stage reuse the previous row's horizontal result.
    const int sh1 = BD - 6, off1 = 1 << (sh1 - 1);
    {
        const int npair = pw >> 1;                                   // 6 or 10
        const int nseg = npair == 10 ? 3 : 5;                        // 2 * npair * nseg <= 64
        const int rps = (ph + nseg - 1) / nseg;                      // rows per segment
        const int per_ref = npair * nseg;
        const int i = lane >= per_ref, id = lane - i * per_ref;
        const int seg = npair == 10 ? (id >= 20 ? 2 : id >= 10 ? 1 : 0) : (id >= 24 ? 4 : id >= 18 ? 3 : id >= 12 ? 2 : id >= 6 ? 1 : 0);
        const int cp = id - seg * npair;
        if (lane < 2 * per_ref) {
            const int mx = mv[2 * i] & 15, my = mv[2 * i + 1] & 15;
            const uint32_t hc = pack16(16 - mx, mx);
            const uint16_t *win = L.win[i] + 2 * cp;
            const int r0 = seg * rps, r1 = min(r0 + rps, ph);
            if constexpr (BD <= 10) {
                // Up to 10 bits the four cases of inter.dmvr[!!my][!!mx] are one formula: with a zero fraction the two-tap filter is
                // 16 x sample, and (16 s + off1) >> sh1 = s << (10 - bd), ((16 t + 8) >> 4) = t exactly — so the general
                // horizontal-then-vertical form reproduces the copy, the h-only and the v-only variants bit for bit, without the
                // divergent branches (the two references of a wave have different fractions).  At 12 bits the v-only variant rounds
                // once where the general form would round twice: the case analysis stays (below).
                // ... and both columns of the lane's pair go through packed 16-bit arithmetic: every intermediate is at most
                // 16 x (2^bd - 1) + 8 < 2^16
                typedef unsigned short pku16 __attribute__((ext_vector_type(2)));
                auto PK = [](uint32_t v) { return __builtin_bit_cast(pku16, v); };
                auto SP = [](int v) { return pku16{ (unsigned short)v, (unsigned short)v }; };
                const pku16 MX = SP(mx), MX16 = SP(16 - mx), MY = SP(my), MY16 = SP(16 - my), OFF1 = SP(off1), EIGHT = SP(8);
                auto hstage = [&](int r) -> pku16 {
                    const uint32_t p0 = *(const uint32_t *)(win + r * kWinW), p1 = *(const uint32_t *)(win + r * kWinW + 2);
                    return (PK(p0) * MX16 + PK(__builtin_amdgcn_alignbit(p1, p0, 16)) * MX + OFF1) >> SP(sh1);
                };
                pku16 a = hstage(r0);
                volatile int16_t *sh_copy = L.bil[i][1];
                for (int r = r0; r < r1; r++) {
                    const pku16 b = hstage(r + 1);
                    const pku16 v = (a * MY16 + b * MY + EIGHT) >> SP(4);
                    const int e = r * kBilP + 2 * cp;
                    *(uint32_t *)&L.bil[i][0][e] = __builtin_bit_cast(uint32_t, v);
                    // shifted copy: element j holds natural element j + 1 (slot -1 of row 0 lands in the 4 spare elements).  Two
                    // 16-bit stores on purpose (volatile): merged into one 32-bit store at a 2-byte aligned address they are slow
                    sh_copy[e + 3] = (int16_t)v.x;
                    sh_copy[e + 4] = (int16_t)v.y;
                    a = b;
                }
            } else {
            // horizontal stage of row r for the two columns
            auto hstage = [&](int r, int &t0, int &t1) {
                const uint32_t p0 = *(const uint32_t *)(win + r * kWinW), p1 = *(const uint32_t *)(win + r * kWinW + 2);
                if (mx) {
                    t0 = (dot2(p0, hc, 0) + off1) >> sh1;
                    t1 = (dot2(__builtin_amdgcn_alignbit(p1, p0, 16), hc, 0) + off1) >> sh1;     // <= 1024: the int16 store of the reference changes nothing
                } else {
                    t0 = p0 & 0xffff; t1 = p0 >> 16;
                }
            };
            int a0, a1;
            hstage(r0, a0, a1);
            for (int r = r0; r < r1; r++) {
                int v0, v1, b0 = 0, b1 = 0;
                if (my)
                    hstage(r + 1, b0, b1);
                if (mx && my)      { v0 = ((16 - my) * a0 + my * b0 + 8) >> 4;         v1 = ((16 - my) * a1 + my * b1 + 8) >> 4; }
                else if (mx)       { v0 = a0;                                           v1 = a1; }
                else if (my)       { v0 = ((16 - my) * a0 + my * b0 + off1) >> sh1;     v1 = ((16 - my) * a1 + my * b1 + off1) >> sh1; }
                else               { v0 = (a0 + (1 << (BD - 11))) >> (BD - 10);         v1 = (a1 + (1 << (BD - 11))) >> (BD - 10); }
                const int e = r * kBilP + 2 * cp;
                *(uint32_t *)&L.bil[i][0][e] = pack16(v0, v1);
                L.bil[i][1][e + 3] = (int16_t)v0;
                L.bil[i][1][e + 4] = (int16_t)v1;
                if (my) { a0 = b0; a1 = b1; }
                else if (r + 1 < r1) hstage(r + 1, a0, a1);
            }
            }
        }
    }
    wave_sync();
    // inter.sad (vvcdsp.c:49): every other row.  plane(i, par) + n addresses natural element n of reference i through the copy
    // in which an offset of parity `par` is 4-byte aligned.
    auto plane = [&](int i, int par) { return par ? L.bil[i][1] + 4 - 1 : L.bil[i][0]; };
    int min_sad;
    {
        // centre cost over the whole wave: lane -> (pair of columns lane & 7, row pair lane >> 3)
        const int xp = lane & 7, r = lane >> 3;
        int acc = 0;
        if (2 * xp < w && 2 * r < h) {
            const uint32_t a = *(const uint32_t *)(L.bil[0][0] + (2 + 2 * r) * kBilP + 2 + 2 * xp);
            const uint32_t b = *(const uint32_t *)(L.bil[1][0] + (2 + 2 * r) * kBilP + 2 + 2 * xp);
            acc = __builtin_amdgcn_sad_u16(a, b, 0);
        }
        min_sad = __builtin_amdgcn_readfirstlane(wave_sum(acc));
    }
    min_sad -= min_sad >> 2;
    int min_dx = 2, min_dy = 2;
    searched = 0;
    if (min_sad >= w * h) {
        searched = 1;
        // the 25 costs without wave-wide reductions: lanes 2k and 2k + 1 own offset k and sum alternate row pairs
        {
            const int k = lane >> 1, half = lane & 1;
            uint32_t acc = 0;
            if (k < 25) {
                const int dy = k / 5, dx = k - dy * 5;
                const int16_t *a = plane(0, dx & 1) + dy * kBilP + dx;
                const int16_t *b = plane(1, dx & 1) + (4 - dy) * kBilP + (4 - dx);
                for (int r = half; 2 * r < h; r += 2) {
                    const uint32_t *ar = (const uint32_t *)(a + 2 * r * kBilP), *br = (const uint32_t *)(b + 2 * r * kBilP);
                    uint32_t va[8], vb[8];
#pragma unroll
                    for (int x = 0; x < 8; x++) { va[x] = ar[x]; vb[x] = br[x]; }      // 16 samples; both planes are 20 wide, so in range for w = 8 too
                    // (w is 8 or 16: a wave-uniform branch instead of eight per-lane selects)
#pragma unroll
                    for (int x = 0; x < 4; x++) acc = __builtin_amdgcn_sad_u16(va[x], vb[x], acc);
                    if (w > 8) {
#pragma unroll
                        for (int x = 4; x < 8; x++) acc = __builtin_amdgcn_sad_u16(va[x], vb[x], acc);
                    }
                }
            }
            acc += __shfl_xor(acc, 1, 64);
            if (k < 25 && !half)
                L.sad[k] = k == 12 ? min_sad : (int)acc;
        }
        wave_sync();
        // 8.5.3.4 array entry selection: the centre wins ties, then the earliest offset in scan order (dy outer, dx inner) =
        // the minimum of (cost, priority) pairs
        {
            uint32_t key = 0xffffffffu;
            if (lane < 25)
                key = ((uint32_t)L.sad[lane] << 5) | (uint32_t)(lane == 12 ? 0 : lane + 1);
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1)
                key = min(key, (uint32_t)__shfl_xor((int)key, m, 64));
            key = (uint32_t)__builtin_amdgcn_readfirstlane((int)key);     // the same in every lane: the selection below is scalar work
            min_sad = (int)(key >> 5);
            const int kk = key & 31, k = kk ? kk - 1 : 12;
            min_dy = k / 5;
            min_dx = k - min_dy * 5;
        }
        wave_sync();
        int dmv0 = (min_dx - 2) * 16, dmv1 = (min_dy - 2) * 16;
        if (min_dx != 0 && min_dx != 4 && min_dy != 0 && min_dy != 4) {
            const int k = min_dy * 5 + min_dx;
            const int sc = __builtin_amdgcn_readfirstlane(L.sad[k]);
            dmv0 += parametric_mv_refine(__builtin_amdgcn_readfirstlane(L.sad[k - 1]), sc, __builtin_amdgcn_readfirstlane(L.sad[k + 1]));
            dmv1 += parametric_mv_refine(__builtin_amdgcn_readfirstlane(L.sad[k - 5]), sc, __builtin_amdgcn_readfirstlane(L.sad[k + 5]));
        }
        mv[0] = clip3(mv[0] + dmv0, -(1 << 17), (1 << 17) - 1);            // ff_vvc_clip_mv
        mv[1] = clip3(mv[1] + dmv1, -(1 << 17), (1 << 17) - 1);
        mv[2] = clip3(mv[2] - dmv0, -(1 << 17), (1 << 17) - 1);
        mv[3] = clip3(mv[3] - dmv1, -(1 << 17), (1 << 17) - 1);
    }
    if (min_sad < 2 * w * h)
        bdof = 0;
    min_sad_out = min_sad;
    wave_sync();                                                            // bil / sad are dead from here on
}

// apply_bdof (vvc_inter_template.c:288) for one wave: interior = the two 14-bit predictions, ring = bdof_fetch_samples (:101)
template <int BD>
__device__ __forceinline__ void bdof_wave(const vvc355_bipred_job *job, BipredLds &L, int lane, int w, int h,
                                          const int (&v0)[4], const int (&v1)[4], const int (&ox)[2], const int (&oy)[2],
                                          const int (&fx)[2], const int (&fy)[2], const ClampRect (&rc)[2])
{
    using px_t = typename Px<BD>::type;
    int16_t *smp0 = L.smp[0], *smp1 = L.smp[1];
    {
        const int x = lane & 15;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int y = 2 * ((lane >> 4) + 4 * (i >> 1)) + (i & 1);
            if (x < w && y < h) {
                smp0[(y + 1) * kGs + x + 1] = (int16_t)v0[i];
                smp1[(y + 1) * kGs + x + 1] = (int16_t)v1[i];
            }
        }
    }
    // ring position (x, y) reads the integer sample at (x + (x_frac >> 3), y + (y_frac >> 3)) of the block
    const int n = 2 * (w + 2) + 2 * h;
    for (int e = lane; e < 2 * n; e += 64) {
        const int p = e >= n, i = e - p * n;
        int x, y;
        if (i < w + 2)            { y = -1; x = i - 1; }
        else if (i < 2 * (w + 2)) { y = h;  x = i - (w + 2) - 1; }
        else                      { const int k = i - 2 * (w + 2); y = k >> 1; x = (k & 1) ? w : -1; }
        const uint8_t *plane = (const uint8_t *)(p ? job->ref1 : job->ref0);
        const int stride = p ? job->ref1_stride : job->ref0_stride;
        const int xa = clip3(ox[p] + x + (fx[p] >> 3), rc[p].x0, rc[p].x1), ya = clip3(oy[p] + y + (fy[p] >> 3), rc[p].y0, rc[p].y1);
        const int s = gld<px_t>(plane + row_off(ya, stride) + xa * (int)sizeof(px_t));
        (p ? smp1 : smp0)[(y + 1) * kGs + x + 1] = (int16_t)(s << (14 - BD));
    }
    wave_sync();
    // What the sub-block sums and the output read are combinations of the two references' planes, so those are what is kept:
    //   D = (s0 >> 4) - (s1 >> 4), TH = (gh0 + gh1) >> 1, TV = (gv0 + gv1) >> 1, GHD = gh0 - gh1, GVD = gv0 - gv1
    // with gh / gv the gradients of prof_grad_filter (:135).  The reference replicates the rings of s and of every gradient plane
    // (pad_int16, vvcdsp.c:29); replicating D, TH, TV is the same thing, and GHD / GVD / s are only read inside the block.
    int16_t *pD = L.grad[0], *pTH = L.grad[1], *pTV = L.grad[2];
    // The lane keeps the samples it interpolated (v0 / v1: column lane & 15, rows 2g, 2g + 1, 2g + 8, 2g + 9): their own values,
    // GHD and GVD never go through LDS — only what other lanes read (D, TH, TV for the window sums) is stored.
    int ghd[4], gvd[4];
    {
        const int x = (lane & 15) + 1;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int y = 2 * ((lane >> 4) + 4 * (i >> 1)) + (i & 1) + 1;
            ghd[i] = gvd[i] = 0;
            if (x <= w && y <= h) {
                const int o = y * kGs + x;
                // rows come in pairs (2g, 2g + 1): one vertical neighbour is the lane's own other sample of the pair
                const int up0 = (i & 1) ? v0[i - 1] : (int)smp0[o - kGs], dn0 = (i & 1) ? (int)smp0[o + kGs] : v0[i + 1];
                const int up1 = (i & 1) ? v1[i - 1] : (int)smp1[o - kGs], dn1 = (i & 1) ? (int)smp1[o + kGs] : v1[i + 1];
                const int gh0 = (smp0[o + 1] >> 6) - (smp0[o - 1] >> 6), gv0 = (dn0 >> 6) - (up0 >> 6);
                const int gh1 = (smp1[o + 1] >> 6) - (smp1[o - 1] >> 6), gv1 = (dn1 >> 6) - (up1 >> 6);
                // the reference stores the gradients as int16 (no narrowing happens: |g| <= 2^9)
                pD[o] = (int16_t)((v0[i] >> 4) - (v1[i] >> 4));
                pTH[o] = (int16_t)((gh0 + gh1) >> 1);
                pTV[o] = (int16_t)((gv0 + gv1) >> 1);
                ghd[i] = gh0 - gh1;
                gvd[i] = gv0 - gv1;
            }
        }
    }
    wave_sync();
    // replicate rings: left / right columns first, then whole top / bottom rows
    if (lane < h) {
        const int o = (lane + 1) * kGs;
        pD[o] = pD[o + 1]; pD[o + w + 1] = pD[o + w];
        pTH[o] = pTH[o + 1]; pTH[o + w + 1] = pTH[o + w];
        pTV[o] = pTV[o + 1]; pTV[o + w + 1] = pTV[o + w];
    }
    wave_sync();
    if (lane < w + 2) {
        const int t = lane, b = (h + 1) * kGs + lane;
        pD[t] = pD[t + kGs]; pD[b] = pD[b - kGs];
        pTH[t] = pTH[t + kGs]; pTH[b] = pTH[b - kGs];
        pTV[t] = pTV[t + kGs]; pTV[b] = pTV[b - kGs];
    }
    wave_sync();
    // four lanes per 4x4 sub-block, all (at most 16) sub-blocks at once (derive_bdof_vx_vy :237, apply_bdof_min_block :267):
    // a lane sums one 3x3 quarter of the 6x6 window, two quad exchanges finish the sums, then it writes one row of the sub-block
    const int sbw = w >> 2, nsb = sbw * (h >> 2);
    const int sb = lane >> 2, q = lane & 3;
    if (sb < nsb) {
        const int by = (sb / sbw) * 4, bx = (sb % sbw) * 4;
        int sgx2 = 0, sgy2 = 0, sgxgy = 0, sgxdi = 0, sgydi = 0;
        {
            const int o0 = (by + 3 * (q >> 1)) * kGs + bx + 3 * (q & 1);
#pragma unroll
            for (int j = 0; j < 3; j++)
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    const int o = o0 + j * kGs + i;
                    const int diff = pD[o], th = pTH[o], tv = pTV[o];
                    sgx2 += abs(th);
                    sgy2 += abs(tv);
                    sgxgy += sign_of(tv) * th;
                    sgxdi += -sign_of(th) * diff;
                    sgydi += -sign_of(tv) * diff;
                }
        }
#pragma unroll
        for (int m = 2; m >= 1; m >>= 1) {
            sgx2 += __shfl_xor(sgx2, m, 4);
            sgy2 += __shfl_xor(sgy2, m, 4);
            sgxgy += __shfl_xor(sgxgy, m, 4);
            sgxdi += __shfl_xor(sgxdi, m, 4);
            sgydi += __shfl_xor(sgydi, m, 4);
        }
        const int vx = sgx2 > 0 ? clip3((sgxdi * 4) >> ilog2(sgx2), -15, 15) : 0;
        const int vy = sgy2 > 0 ? clip3(((sgydi * 4) - ((vx * sgxgy) >> 1)) >> ilog2(sgy2), -15, 15) : 0;
        if (q == 0)
            L.sad[sb] = (vx & 0xffff) | (vy << 16);       // the DMVR cost array is dead by now: (vx, vy) of sub-block sb
    }
    wave_sync();
    // apply_bdof_min_block (:267) on the lane's own four samples
    {
        const int sh = 15 - BD, off = 1 << (sh - 1);
        const int x = lane & 15;
        uint8_t *dst0 = (uint8_t *)job->dst;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int y = 2 * ((lane >> 4) + 4 * (i >> 1)) + (i & 1);
            if (x >= w || y >= h)
                continue;
            const int vv = L.sad[(y >> 2) * sbw + (x >> 2)];
            const int vx = (int16_t)vv, vy = vv >> 16;
            const int p = (v0[i] + off + v1[i] + vx * ghd[i] + vy * gvd[i]) >> sh;
            gst_at<px_t>(dst0, (uint32_t)(__mul24(y, job->dst_stride) + x * (int)sizeof(px_t)), (px_t)lmcs_fwd<BD>((const uint8_t *)job->lmcs_lut, clip_px<BD>(p)));
        }
    }
}

// both references at motion mv: windows through clamped coordinates, then the separable interpolation of interp_block
// both = false: uni-prediction — only the first reference (the caller put the list in use at index 0 of every array and in pa / sa)
template <int BD, int NTAP>
__device__ __forceinline__ void predict_clamped(const vvc355_bipred_job *job, BipredLds &L, int lane, int lw, int h,
                                                const int (&ox)[2], const int (&oy)[2], const int (&fx)[2], const int (&fy)[2],
                                                const ClampRect (&rc)[2], int (&v0)[4], int (&v1)[4],
                                                const uint8_t *pa, int sa, const uint8_t *pb, int sb, bool both)
{
    constexpr int LEAD = NTAP == 8 ? 3 : 1, NIT = (16 + NTAP) / 2;
    {
        const int ax = ox[0] - LEAD, ay = oy[0] - (fy[0] ? LEAD : 0), an = fy[0] ? h + NTAP - 1 : h;
        const int bx = ox[1] - LEAD, by = oy[1] - (fy[1] ? LEAD : 0), bn = fy[1] ? h + NTAP - 1 : h;
        if (rect_holds(rc[0], ax, ay, an) && (!both || rect_holds(rc[1], bx, by, bn))) {
            constexpr int NV = NTAP == 8 ? 3 : 2;            // 23 x 6 = 138 <= 192 (luma), 19 x 6 = 114 <= 128 (chroma, h <= 16)
            uint2 q0[NV], q1[NV];
            fetch_vec4<BD, NV>(pa, sa, ax, ay, an, lane, q0);
            if (both)
                fetch_vec4<BD, NV>(pb, sb, bx, by, bn, lane, q1);
            put_vec4<NV>(L.win[0], an, lane, q0);
            if (both)
                put_vec4<NV>(L.win[1], bn, lane, q1);
        } else {
            uint16_t r0[NIT], r1[NIT];
            fetch_clamped<BD, NIT>(pa, sa, rc[0], ax, ay, lane, r0);
            if (both)
                fetch_clamped<BD, NIT>(pb, sb, rc[1], bx, by, lane, r1);
            store_rows<NIT>(L.win[0], lane, r0);
            if (both)
                store_rows<NIT>(L.win[1], lane, r1);
        }
        wave_sync();
    }
    uint32_t t[2][4];       // hf lo/hi, vf lo/hi per reference
#pragma unroll
    for (int i = 0; i < 2; i++) {
        if (NTAP == 8) {
            const uint2 hf = gld<uint2>(d_tab_inter_luma_filters + (job->hf_idx * 16 + fx[i]) * 8);
            const uint2 vf = gld<uint2>(d_tab_inter_luma_filters + (job->vf_idx * 16 + fy[i]) * 8);
            t[i][0] = hf.x; t[i][1] = hf.y; t[i][2] = vf.x; t[i][3] = vf.y;
        } else {
            t[i][0] = gld<uint32_t>(d_tab_inter_chroma_filters + (job->hf_idx * 32 + fx[i]) * 4); t[i][1] = 0;
            t[i][2] = gld<uint32_t>(d_tab_inter_chroma_filters + (job->vf_idx * 32 + fy[i]) * 4); t[i][3] = 0;
        }
    }
    interp_block<BD, NTAP>(lw, h, fx[0] != 0, fy[0] != 0, t[0][0], t[0][1], t[0][2], t[0][3], L.win[0], L.tmpT, lane, v0);
    if (both)
        interp_block<BD, NTAP>(lw, h, fx[1] != 0, fy[1] != 0, t[1][0], t[1][1], t[1][2], t[1][3], L.win[1], L.tmpT, lane, v1);
    else {
#pragma unroll
        for (int i = 0; i < 4; i++) v1[i] = 0;
    }
}

// the window / intermediate part of BipredLds only: what a launch of chroma jobs needs (more waves per CU)
struct BipredLdsLight {
    uint16_t win[2][kWinH * kWinW];
    int16_t tmpT[16 * kTmpP];
};

#include "mc_tools.hpp"

// what one wave of a luma launch needs: the general path's planes or the tools path's (same footprint, 5.6 KB: 28 waves per CU)
union BipredLdsAll {
    BipredLds gen;
    ToolsLds<16, 16> tools;
};

// gpm != nullptr: the two predictions are the two parts of a geometric-partition coding unit (pred_gpm_blk, vvc_inter.c:466-527:
// luma_mc / chroma_mc per part, then inter.put_gpm with the per-sample weights of the partition's mask)
template <int BD, bool TOOLS>
__device__ __forceinline__ void bipred_one(const vvc355_bipred_job *job, BipredLds &L, int lane, const vvc355_gpm_job *gpm = nullptr)
{
    if (!TOOLS && !job->chroma)
        return;                                          // contract: a chroma-only launch holds chroma jobs
    const int uni = job->pred_flag == 1 || job->pred_flag == 2;        // luma_mc_uni / chroma_mc_uni: one list, no DMVR / BDOF
    const int w = job->w, h = job->h, chroma = job->chroma, dmvr = job->dmvr && !uni;
    if constexpr (TOOLS) {
        // bi-predicted luma sub-blocks with DMVR and / or BDOF (8 or 16 on a side, the only shapes those tools run on): mc_tools.hpp
        if (!chroma && !uni && (job->dmvr || job->bdof) && (w == 8 || w == 16) && (h == 8 || h == 16)) {
            if (w == 16 && h == 16)     bipred_tools<BD, 16, 16>(job, *(ToolsLds<16, 16> *)&L, lane);
            else if (w == 16)           bipred_tools<BD, 16, 8>(job, *(ToolsLds<16, 8> *)&L, lane);
            else if (h == 16)           bipred_tools<BD, 8, 16>(job, *(ToolsLds<8, 16> *)&L, lane);
            else                        bipred_tools<BD, 8, 8>(job, *(ToolsLds<8, 8> *)&L, lane);
            return;
        }
    }
    const int lw = 31 - __builtin_clz(w);
    vvc355_bipred_result *rec = (vvc355_bipred_result *)job->rec;
    int mv[4] = { job->mv[0], job->mv[1], job->mv[2], job->mv[3] };
    int bdof = TOOLS && !chroma && job->bdof && !uni;
    if (chroma && rec) {
#pragma unroll
        for (int k = 0; k < 4; k++) mv[k] = gld<int>(&rec->mv[k]);
    }
    if (TOOLS && !chroma) {
        int min_sad = 0, searched = 0;
        if (dmvr) {
            dmvr_refine<BD>(job, L, lane, mv, bdof, min_sad, searched);
            // every lane holds the same refined motion: say so, and everything derived from it (positions, fractions, readable
            // rectangles, filter taps, window base addresses) is scalar work instead of 64 identical lanes of vector work
#pragma unroll
            for (int k = 0; k < 4; k++) mv[k] = __builtin_amdgcn_readfirstlane(mv[k]);
            bdof = __builtin_amdgcn_readfirstlane(bdof);
            min_sad = __builtin_amdgcn_readfirstlane(min_sad);
        }
        if (rec && lane == 0) {
#pragma unroll
            for (int k = 0; k < 4; k++) gst<int>(&rec->mv[k], mv[k]);
            gst<int>(&rec->bdof, bdof);
            gst<int>(&rec->min_sad, min_sad);
            gst<int>(&rec->searched, searched);
        }
    }
    // integer positions, fractions, readable rectangles (luma_mc_bi :262-283 / chroma_mc_bi :344-362, emulated_edge* :33-88)
    const int before = chroma ? 1 : 3, after = chroma ? 2 : 4;
    const int shx = 4 + (chroma ? job->hs : 0), shy = 4 + (chroma ? job->vs : 0);
    const uint8_t *lut = chroma ? nullptr : (const uint8_t *)job->lmcs_lut;      // luma of an LMCS slice is stored through the forward map
    int ox[2], oy[2], fx[2], fy[2];
    ClampRect rc[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int mvx = mv[2 * i], mvy = mv[2 * i + 1];
        fx[i] = chroma ? (mvx & ((1 << shx) - 1)) << (1 - job->hs) : mvx & 15;
        fy[i] = chroma ? (mvy & ((1 << shy) - 1)) << (1 - job->vs) : mvy & 15;
        ox[i] = job->x + (mvx >> shx);
        oy[i] = job->y + (mvy >> shy);
        rc[i] = ClampRect{ 0, 0, job->pic_w - 1, job->pic_h - 1 };
        if (dmvr) {
            const int x_sb = job->x + (job->mv[2 * i] >> shx), y_sb = job->y + (job->mv[2 * i + 1] >> shy);
            rc[i].x0 = min(max(x_sb - before, 0), job->pic_w - 1);
            rc[i].y0 = min(max(y_sb - before, 0), job->pic_h - 1);
            rc[i].x1 = rc[i].x0 + max(min((int)job->pic_w, x_sb + w + after) - rc[i].x0, 1) - 1;
            rc[i].y1 = rc[i].y0 + max(min((int)job->pic_h, y_sb + h + after) - rc[i].y0, 1) - 1;
        }
    }
    int v0[4], v1[4];
    const uint8_t *pa = (const uint8_t *)job->ref0, *pb = (const uint8_t *)job->ref1;
    int sa = job->ref0_stride, sb = job->ref1_stride;
    if (uni && job->pred_flag == 2) {                     // list 1 only: it takes the first slot
        pa = pb; sa = sb;
        ox[0] = ox[1]; oy[0] = oy[1]; fx[0] = fx[1]; fy[0] = fy[1]; rc[0] = rc[1];
    }
    if (chroma)
        predict_clamped<BD, 4>(job, L, lane, lw, h, ox, oy, fx, fy, rc, v0, v1, pa, sa, pb, sb, !uni);
    else
        predict_clamped<BD, 8>(job, L, lane, lw, h, ox, oy, fx, fy, rc, v0, v1, pa, sa, pb, sb, !uni);
    if (uni) {
        // put_uni / put_uni_w (h2656_inter_template.c:44-81): rounding to pixels, weights from derive_weight_uni in (denom, w0, o0)
        const int wfu = job->weight_flag, sh = (wfu ? job->denom : 0) + 14 - BD, rnd = 1 << (sh - 1);
        const int xu = lane & 15;
        uint8_t *dstu = (uint8_t *)job->dst;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int y = 2 * ((lane >> 4) + 4 * (i >> 1)) + (i & 1);
            if (xu >= w || y >= h)
                continue;
            const int p = wfu ? ((v0[i] * job->w0 + rnd) >> sh) + job->o0 * (1 << (BD - 8)) : (v0[i] + rnd) >> sh;
            gst_at<typename Px<BD>::type>(dstu, (uint32_t)(__mul24(y, job->dst_stride) + xu * (int)sizeof(typename Px<BD>::type)), (typename Px<BD>::type)lmcs_fwd<BD>(lut, clip_px<BD>(p)));
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) { v0[i] = (int16_t)v0[i]; v1[i] = (int16_t)v1[i]; }     // put[..] stores int16
    if (gpm) {
        // put_gpm (vvc_inter_template.c:78): (s0 w + s1 (8 - w) + offset) >> max(5, 17 - bd), w = weights[y step_y + x step_x]
        constexpr int gsh = BD <= 12 ? 17 - BD : 5, goff = 1 << (gsh - 1);
        const uint8_t *wt = (const uint8_t *)gpm->weights;
        const int xg = lane & 15;
        uint8_t *dstg = (uint8_t *)job->dst;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int y = 2 * ((lane >> 4) + 4 * (i >> 1)) + (i & 1);
            if (xg >= w || y >= h)
                continue;
            const int wg = gld<uint8_t>(wt + y * gpm->step_y + xg * gpm->step_x);
            const int p = (v0[i] * wg + v1[i] * (8 - wg) + goff) >> gsh;
            gst_at<typename Px<BD>::type>(dstg, (uint32_t)(__mul24(y, job->dst_stride) + xg * (int)sizeof(typename Px<BD>::type)), (typename Px<BD>::type)lmcs_fwd<BD>(lut, clip_px<BD>(p)));
        }
        return;
    }
    if (TOOLS && bdof) {
        bdof_wave<BD>(job, L, lane, w, h, v0, v1, ox, oy, fx, fy, rc);
        return;
    }
    int shift, off;
    const int wf = job->weight_flag, w0 = job->w0, w1 = job->w1;
    if (!wf) { shift = max(3, 15 - BD); off = 1 << (shift - 1); }                                                   // avg
    else     { shift = job->denom + max(3, 15 - BD); off = (((job->o0 + job->o1) << (BD - 8)) + 1) << (shift - 1); } // w_avg
    const int x = lane & 15;
    uint8_t *dst = (uint8_t *)job->dst;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int y = 2 * ((lane >> 4) + 4 * (i >> 1)) + (i & 1);
        if (x >= w || y >= h)
            continue;
        const int p = wf ? (v0[i] * w0 + v1[i] * w1 + off) >> shift : (v0[i] + v1[i] + off) >> shift;
        gst_at<typename Px<BD>::type>(dst, (uint32_t)(__mul24(y, job->dst_stride) + x * (int)sizeof(typename Px<BD>::type)), (typename Px<BD>::type)lmcs_fwd<BD>(lut, clip_px<BD>(p)));
    }
}

template <int BD, bool TOOLS>
__global__ __launch_bounds__(256) void bipred_kernel(const vvc355_bipred_job *__restrict__ jobs, int n_jobs)
{
    __shared__ __attribute__((aligned(16))) typename std::conditional<TOOLS, BipredLdsAll, BipredLdsLight>::type lds_all[4];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int ji = xcd_chunked(blockIdx.x, gridDim.x) * 4 + wave;
    if (ji >= n_jobs)
        return;
    // The descriptor is copied dword-wise at a wave-uniform address (scalar loads, issued once); reading its byte / short
    // fields through the pointer would be a vector load with a full memory round trip at every point of use.
    const vvc355_bipred_job job_copy = load_uniform(jobs + ji);
    bipred_one<BD, TOOLS>(&job_copy, *(BipredLds *)&lds_all[wave], lane);      // without TOOLS only win / tmpT are touched
}

// Geometric-partition blocks: one wave per (<= 16x16 tile of a) part pair.  The job's base is a bi-prediction job without tools
// whose two references are the two parts' reference pictures.
template <int BD>
__global__ __launch_bounds__(256) void gpm_kernel(const vvc355_gpm_job *__restrict__ jobs, int n_jobs)
{
    __shared__ __attribute__((aligned(16))) BipredLdsLight lds_all[4];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int ji = xcd_chunked(blockIdx.x, gridDim.x) * 4 + wave;
    if (ji >= n_jobs)
        return;
    const vvc355_gpm_job job = load_uniform(jobs + ji);
    vvc355_bipred_job base = job.base;
    base.pred_flag = 3; base.dmvr = 0; base.bdof = 0; base.weight_flag = 0; base.rec = 0;
    if (base.chroma)
        bipred_one<BD, false>(&base, *(BipredLds *)&lds_all[wave], lane, &job);
    else
        bipred_one<BD, true>(&base, *(BipredLds *)&lds_all[wave], lane, &job);
}

// Chroma launch: one wave per PAIR of consecutive jobs.  When the two are the Cb and Cr blocks of one sub-block (same
// geometry and motion, width <= 8) they are predicted together as one 16-wide block whose halves come from two planes —
// an 8x8 block alone leaves half of the lanes of the vertical pass and of the window fetch idle.  Any other pair is done one
// job after the other.
template <int BD>
__global__ __launch_bounds__(256) void bipred_chroma_pair_kernel(const vvc355_bipred_job *__restrict__ jobs, int n_jobs)
{
    __shared__ __attribute__((aligned(16))) BipredLdsLight lds_all[4];
    using px_t = typename Px<BD>::type;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int ia = 2 * (xcd_chunked(blockIdx.x, gridDim.x) * 4 + wave);
    if (ia >= n_jobs)
        return;
    const bool has_b = ia + 1 < n_jobs;
    const vvc355_bipred_job ja = load_uniform(jobs + ia), jb = load_uniform(jobs + (has_b ? ia + 1 : ia));
    BipredLds &L = *(BipredLds *)&lds_all[wave];
    const bool bi_a = !(ja.pred_flag == 1 || ja.pred_flag == 2), bi_b = !(jb.pred_flag == 1 || jb.pred_flag == 2);
    bool pair = has_b && bi_a && bi_b && ja.chroma && jb.chroma && ja.w <= 8 && ja.w == jb.w && ja.h == jb.h && ja.x == jb.x && ja.y == jb.y &&
                ja.rec == jb.rec && ja.hs == jb.hs && ja.vs == jb.vs && ja.dmvr == jb.dmvr && ja.hf_idx == jb.hf_idx &&
                ja.vf_idx == jb.vf_idx && ja.pic_w == jb.pic_w && ja.pic_h == jb.pic_h;
#pragma unroll
    for (int k = 0; k < 4; k++) pair = pair && ja.mv[k] == jb.mv[k];
    if (!pair) {
        bipred_one<BD, false>(&ja, L, lane);
        if (has_b) {
            wave_sync();
            bipred_one<BD, false>(&jb, L, lane);
        }
        return;
    }
    const vvc355_bipred_job *job = &ja;
    const int w = job->w, h = job->h, dmvr = job->dmvr;
    const vvc355_bipred_result *rec = (const vvc355_bipred_result *)job->rec;
    int mv[4] = { job->mv[0], job->mv[1], job->mv[2], job->mv[3] };
    if (rec) {
#pragma unroll
        for (int k = 0; k < 4; k++) mv[k] = rec->mv[k];                  // written by the luma launch: plain (scalar) loads
    }
    const int shx = 4 + job->hs, shy = 4 + job->vs;
    int ox[2], oy[2], fx[2], fy[2];
    ClampRect rc[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int mvx = mv[2 * i], mvy = mv[2 * i + 1];
        fx[i] = (mvx & ((1 << shx) - 1)) << (1 - job->hs);
        fy[i] = (mvy & ((1 << shy) - 1)) << (1 - job->vs);
        ox[i] = job->x + (mvx >> shx);
        oy[i] = job->y + (mvy >> shy);
        rc[i] = ClampRect{ 0, 0, job->pic_w - 1, job->pic_h - 1 };
        if (dmvr) {
            const int x_sb = job->x + (job->mv[2 * i] >> shx), y_sb = job->y + (job->mv[2 * i + 1] >> shy);
            rc[i].x0 = min(max(x_sb - 1, 0), job->pic_w - 1);
            rc[i].y0 = min(max(y_sb - 1, 0), job->pic_h - 1);
            rc[i].x1 = rc[i].x0 + max(min((int)job->pic_w, x_sb + w + 2) - rc[i].x0, 1) - 1;
            rc[i].y1 = rc[i].y0 + max(min((int)job->pic_h, y_sb + h + 2) - rc[i].y0, 1) - 1;
        }
    }
    // windows: columns 0..11 from the first plane, 12..23 from the second, both starting one sample left of the block
    {
        const int c = lane & 31, second = c >= 12, cc = c - (second ? 12 : 0);
        uint16_t r[2][10];
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const uint8_t *pa = (const uint8_t *)(i ? ja.ref1 : ja.ref0), *pb = (const uint8_t *)(i ? jb.ref1 : jb.ref0);
            const int sa = i ? ja.ref1_stride : ja.ref0_stride, sb = i ? jb.ref1_stride : jb.ref0_stride;
            const uint8_t *plane = second ? pb : pa;
            const int stride = second ? sb : sa;
            const int xa = clip3(ox[i] - 1 + cc, rc[i].x0, rc[i].x1);
            const uint8_t *col = plane + xa * (int)sizeof(px_t);
            const int wy0 = oy[i] - (fy[i] ? 1 : 0);
#pragma unroll
            for (int it = 0; it < 10; it++) {
                const int ya = clip3(wy0 + (lane >> 5) + 2 * it, rc[i].y0, rc[i].y1);
                r[i][it] = (uint16_t)gld<px_t>(col + row_off(ya, stride));
            }
        }
        store_rows<10>(L.win[0], lane, r[0]);
        store_rows<10>(L.win[1], lane, r[1]);
        wave_sync();
    }
    uint32_t t[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        t[i][0] = gld<uint32_t>(d_tab_inter_chroma_filters + (job->hf_idx * 32 + fx[i]) * 4);
        t[i][1] = gld<uint32_t>(d_tab_inter_chroma_filters + (job->vf_idx * 32 + fy[i]) * 4);
    }
    int v0[4], v1[4];
    interp_block<BD, 4, true>(4, h, fx[0] != 0, fy[0] != 0, t[0][0], 0, t[0][1], 0, L.win[0], L.tmpT, lane, v0);
    interp_block<BD, 4, true>(4, h, fx[1] != 0, fy[1] != 0, t[1][0], 0, t[1][1], 0, L.win[1], L.tmpT, lane, v1);
#pragma unroll
    for (int i = 0; i < 4; i++) { v0[i] = (int16_t)v0[i]; v1[i] = (int16_t)v1[i]; }     // put[..] stores int16
    // lanes 0..7 of a row write the first plane, 8..15 the second, each with its own weights
    const int x = lane & 15, second = x >= 8, xo = x - (second ? 8 : 0);
    const int wf = second ? jb.weight_flag : ja.weight_flag, w0 = second ? jb.w0 : ja.w0, w1 = second ? jb.w1 : ja.w1;
    const int denom = second ? jb.denom : ja.denom, osum = second ? jb.o0 + jb.o1 : ja.o0 + ja.o1;
    const int shift = wf ? denom + max(3, 15 - BD) : max(3, 15 - BD);
    const int off = wf ? ((osum << (BD - 8)) + 1) << (shift - 1) : 1 << (shift - 1);
    uint8_t *dst = (uint8_t *)(second ? jb.dst : ja.dst);
    const int dst_stride = second ? jb.dst_stride : ja.dst_stride;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int y = 2 * ((lane >> 4) + 4 * (i >> 1)) + (i & 1);
        if (xo >= w || y >= h)
            continue;
        const int p = wf ? (v0[i] * w0 + v1[i] * w1 + off) >> shift : (v0[i] + v1[i] + off) >> shift;
        st_px<BD>(dst + row_off(y, dst_stride), xo, clip_px<BD>(p));
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

extern "C" void vvc355_bipred_batch(void *stream, int bd, const vvc355_bipred_job *jobs_dev, int n_jobs)
{
    using namespace vvc355;
    if (n_jobs <= 0) return;
    VVC355_BD_DISPATCH(bd, hipLaunchKernelGGL((bipred_kernel<BD, true>), dim3((n_jobs + 3) / 4), dim3(256), 0, (hipStream_t)stream, jobs_dev, n_jobs));
    HIP_CHECK(hipGetLastError());
}

extern "C" void vvc355_gpm_batch(void *stream, int bd, const vvc355_gpm_job *jobs_dev, int n_jobs)
{
    using namespace vvc355;
    if (n_jobs <= 0) return;
    VVC355_BD_DISPATCH(bd, hipLaunchKernelGGL((gpm_kernel<BD>), dim3((n_jobs + 3) / 4), dim3(256), 0, (hipStream_t)stream, jobs_dev, n_jobs));
    HIP_CHECK(hipGetLastError());
}

extern "C" void vvc355_bipred_chroma_batch(void *stream, int bd, const vvc355_bipred_job *jobs_dev, int n_jobs)
{
    using namespace vvc355;
    if (n_jobs <= 0) return;
    const int n_pairs = (n_jobs + 1) / 2;
    VVC355_BD_DISPATCH(bd, hipLaunchKernelGGL((bipred_chroma_pair_kernel<BD>), dim3((n_pairs + 3) / 4), dim3(256), 0, (hipStream_t)stream, jobs_dev, n_jobs));
    HIP_CHECK(hipGetLastError());
}

