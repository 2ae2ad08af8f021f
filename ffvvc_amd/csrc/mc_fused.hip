// Fused motion-compensated prediction for gfx950: for every block of at most 16x16 samples, interpolate from one or two
// reference pictures (8-tap luma / 4-tap chroma, any of the copy / h / v / hv cases) and blend (avg, w_avg) or round
// (put_uni, put_uni_w) straight to pixels — the 14-bit intermediates never go to HBM.
//
// This is the batched form of what the reference does per prediction block in luma_mc_bi / chroma_mc_bi / *_mc_uni
// (libavcodec/vvc/vvc_inter.c:222-460): put[..] x2 + avg / w_avg, or put_uni / put_uni_w.  Arithmetic follows
// libavcodec/h26x/h2656_inter_template.c:29-577 and libavcodec/vvc/vvc_inter_template.c:25-58 exactly (int16 narrowing of
// the horizontal pass included).  Larger prediction blocks are cut into <= 16x16 tiles by the job builder; the
// interpolation is separable per output tile, so the result does not depend on the tiling.
//
// Mapping: one wave per block, four blocks per workgroup, no workgroup barrier.  The source window goes to LDS as uint16;
// each lane produces two horizontally adjacent outputs per step from aligned sample pairs with v_dot2c_i32_i16 (9 dot
// products for 16 taps), writes the intermediate transposed so that the vertical pass again reads aligned pairs.
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

static constexpr int kWinW = 24;      // LDS source window: up to 16 + 7 columns, even pitch
static constexpr int kWinH = 23;
static constexpr int kTmpP = 24;      // transposed intermediate: [column][row], up to 23 rows, even pitch

// two adjacent outputs of an 8-tap filter from five aligned sample pairs d[0..4] = (p[0],p[1]) .. (p[8],p[9]):
// out0 = sum f[k] p[k], out1 = sum f[k] p[k+1]
struct Taps {
    uint32_t e[4];     // (f0,f1) (f2,f3) (f4,f5) (f6,f7)
    uint32_t o[5];     // (0,f0) (f1,f2) (f3,f4) (f5,f6) (f7,0)
    __device__ __forceinline__ void set(const int8_t *f)
    {
        e[0] = pack16(f[0], f[1]); e[1] = pack16(f[2], f[3]); e[2] = pack16(f[4], f[5]); e[3] = pack16(f[6], f[7]);
        o[0] = pack16(0, f[0]); o[1] = pack16(f[1], f[2]); o[2] = pack16(f[3], f[4]); o[3] = pack16(f[5], f[6]); o[4] = pack16(f[7], 0);
    }
    __device__ __forceinline__ void apply(const uint32_t *d, int &out0, int &out1) const
    {
        out0 = dot2(d[3], e[3], dot2(d[2], e[2], dot2(d[1], e[1], dot2(d[0], e[0], 0))));
        out1 = dot2(d[4], o[4], dot2(d[3], o[3], dot2(d[2], o[2], dot2(d[1], o[1], dot2(d[0], o[0], 0)))));
    }
};

// One reference of one block -> up to 4 intermediate values per lane (two row pairs of one column), 14-bit scaled ints.
// Lane layout of the result: column x = lane & 15, row pair yp = (lane >> 4) + 4 * i  (i = 0, 1), rows 2*yp, 2*yp + 1.
template <int BD>
__device__ __forceinline__ void interp_block(const uint8_t *src, int src_stride, int w, int h, bool chroma, bool hfrac, bool vfrac,
                                             const int8_t *hf, const int8_t *vf, uint16_t *win, int16_t *tmpT, int lane, int (&val)[4])
{
    using px_t = typename Px<BD>::type;
    // 4-tap chroma filters are applied as 8-tap filters with the taps at positions -1..2 (indices 2..5)
    int8_t fh[8], fv[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        fh[k] = chroma ? ((k >= 2 && k < 6) ? hf[k - 2] : 0) : hf[k];
        fv[k] = chroma ? ((k >= 2 && k < 6) ? vf[k - 2] : 0) : vf[k];
    }
    const int lead = chroma ? 1 : 3, trail = chroma ? 2 : 4;
    const int sw = w + 8, sh = vfrac ? h + 7 : h;            // window columns -3 .. w+4 (index = col + 3), rows -3 .. h+3
    const int row0 = vfrac ? -3 : 0;
    // stage the window; columns / rows the reference would not touch are left as zeros (they only meet zero taps)
    for (int r = lane >> 5; r < sh; r += 2) {
        const int c = lane & 31;
        if (c < sw) {
            const int col = c - 3, row = r + row0;
            const bool need = (hfrac ? (col >= -lead && col < w + trail) : (col >= 0 && col < w)) &&
                              (vfrac ? (row >= -lead && row < h + trail) : true);
            int v = 0;
            if (need)
                v = ((const px_t *)(src + (ptrdiff_t)row * src_stride))[col];
            win[r * kWinW + c] = (uint16_t)v;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // horizontal pass -> tmpT[x][r] (int16), r over the sh window rows
    const int half_w = w >> 1;
    if (hfrac) {
        Taps t;
        t.set(fh);
        for (int i = lane; i < sh * half_w; i += 64) {
            const int r = i / half_w, xp = i - r * half_w;           // outputs x = 2*xp, 2*xp + 1
            const uint32_t *d = (const uint32_t *)(win + r * kWinW + 2 * xp);   // window index of column x-3 is x: even -> aligned
            const uint32_t dd[5] = { d[0], d[1], d[2], d[3], d[4] };
            int o0, o1;
            t.apply(dd, o0, o1);
            tmpT[(2 * xp) * kTmpP + r] = (int16_t)(o0 >> (BD - 8));
            tmpT[(2 * xp + 1) * kTmpP + r] = (int16_t)(o1 >> (BD - 8));
        }
    } else {
        for (int i = lane; i < sh * w; i += 64) {
            const int r = i / w, x = i - r * w;
            tmpT[x * kTmpP + r] = (int16_t)win[r * kWinW + x + 3];       // raw samples
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // vertical pass: lane -> column x, row pairs
    const int x = lane & 15;
    Taps t;
    t.set(fv);
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int yp = (lane >> 4) + 4 * i;
        int o0 = 0, o1 = 0;
        if (x < w && 2 * yp < h) {
            if (vfrac) {
                const uint32_t *d = (const uint32_t *)(tmpT + x * kTmpP + 2 * yp);   // rows 2yp-3.. are window rows 2yp..: aligned
                const uint32_t dd[5] = { d[0], d[1], d[2], d[3], d[4] };
                t.apply(dd, o0, o1);
                // hv: second stage >> 6 on the int16 intermediates; v only: first stage on raw samples >> (bd - 8)
                const int sh2 = hfrac ? 6 : BD - 8;
                o0 >>= sh2; o1 >>= sh2;
            } else {
                o0 = tmpT[x * kTmpP + 2 * yp];
                o1 = tmpT[x * kTmpP + 2 * yp + 1];
                if (!hfrac) { o0 <<= 14 - BD; o1 <<= 14 - BD; }          // integer position: sample << (14 - bd)
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
    __shared__ __attribute__((aligned(16))) uint16_t win_all[4][kWinH * kWinW];
    __shared__ __attribute__((aligned(16))) int16_t tmp_all[4][16 * kTmpP];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ji = blockIdx.x * 4 + wave;
    if (ji >= n_jobs)
        return;
    const vvc355_pred_job job = jobs[ji];
    const int w = job.w, h = job.h, mode = job.mode;
    const bool chroma = job.chroma;
    int v0[4], v1[4] = { 0, 0, 0, 0 };
    interp_block<BD>((const uint8_t *)job.src0, job.src0_stride, w, h, chroma, job.frac & 1, job.frac & 2,
                     job.hf0, job.vf0, win_all[wave], tmp_all[wave], lane, v0);
    if (mode < 2) {
        interp_block<BD>((const uint8_t *)job.src1, job.src1_stride, w, h, chroma, job.frac & 4, job.frac & 8,
                         job.hf1, job.vf1, win_all[wave], tmp_all[wave], lane, v1);
        // the reference carries bi-prediction operands in int16 planes (put[..] narrows on store)
#pragma unroll
        for (int i = 0; i < 4; i++) { v0[i] = (int16_t)v0[i]; v1[i] = (int16_t)v1[i]; }
    }

    int shift, off;
    if (mode == 0)      { shift = max(3, 15 - BD); off = 1 << (shift - 1); }                                            // avg
    else if (mode == 1) { shift = job.denom + max(3, 15 - BD); off = (((job.o0 + job.o1) << (BD - 8)) + 1) << (shift - 1); }   // w_avg
    else if (mode == 2) { shift = 14 - BD; off = 1 << (shift - 1); }                                                    // put_uni
    else                { shift = job.denom + 14 - BD; off = 1 << (shift - 1); }                                        // put_uni_w
    const int x = lane & 15;
    uint8_t *dst = (uint8_t *)job.dst;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int y = 2 * ((lane >> 4) + 4 * (i >> 1)) + (i & 1);
        if (x >= w || y >= h)
            continue;
        int p;
        if (mode == 0)      p = (v0[i] + v1[i] + off) >> shift;
        else if (mode == 1) p = (v0[i] * job.w0 + v1[i] * job.w1 + off) >> shift;
        else if (mode == 2) p = (job.frac & 3) ? (v0[i] + off) >> shift : v0[i] >> (14 - BD);      // integer position = plain copy
        else                p = ((v0[i] * job.w0 + off) >> shift) + job.o0 * (1 << (BD - 8));
        st_px<BD>(dst + (ptrdiff_t)y * job.dst_stride, x, clip_px<BD>(p));
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
