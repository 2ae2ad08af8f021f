// Inter-prediction kernels for gfx950: separable 8-tap (luma) / 4-tap (chroma) DCTIF in the put / put_uni /
// put_uni_w flavours, bi-prediction blends (avg, w_avg, CIIP, GPM), BDOF, PROF, the DMVR bilinear fetch and SAD.
//
// Reference behaviour: libavcodec/h26x/h2656_inter_template.c:29-577 (put*), libavcodec/vvc/vvc_inter_template.c:25-413
// (avg :25, w_avg :42, put_ciip :60, put_gpm :78, bdof_fetch_samples :101, prof_grad_filter :135, apply_prof* :160-235,
// derive_bdof_vx_vy :237, apply_bdof :288, dmvr* :324-413), libavcodec/vvc/vvcdsp.c:29 (pad_int16), :49 (vvc_sad).
//
// Data layout: reference pictures are planar pixel arrays in HBM (byte strides); 14-bit intermediates are int16 planes with
// the reference's fixed row stride of 128 elements in slot mode, or any stride the batched caller chooses.
#include "common.hpp"
#include "runtime.hpp"
#include "../../include/vvc_mi355.h"

namespace vvc355 {

static constexpr int kMcTile = 32;                 // output tile edge handled by one workgroup
static constexpr int kMcSrcW = kMcTile + 8;        // + 7 apron, padded to even

__device__ __forceinline__ int fir8(const int8_t *f, int a0, int a1, int a2, int a3, int a4, int a5, int a6, int a7)
{
    return f[0] * a0 + f[1] * a1 + f[2] * a2 + f[3] * a3 + f[4] * a4 + f[5] * a5 + f[6] * a6 + f[7] * a7;
}

// NTAP-tap filter over LDS samples p[0], p[step], ...
template <int NTAP, typename T>
__device__ __forceinline__ int fir_lds(const int8_t *f, const T *p, int step)
{
    int acc = 0;
#pragma unroll
    for (int k = 0; k < NTAP; k++)
        acc += f[k] * (int)p[k * step];
    return acc;
}

// ------------------------------------------------------------------------------------------------ put / put_uni / put_uni_w

template <int BD, int NTAP>
__device__ __forceinline__ void mc_tile(const vvc355_mc_job &job, int x0, int y0, int tw, int th,
                                        uint16_t (*s)[kMcSrcW], int16_t (*t)[kMcTile])
{
    using px_t = typename Px<BD>::type;
    constexpr int LEAD = NTAP == 8 ? 3 : 1;
    const int hfrac = job.hfrac, vfrac = job.vfrac;
    const int lh = hfrac ? LEAD : 0, lv = vfrac ? LEAD : 0;
    const int sw = tw + (hfrac ? NTAP - 1 : 0), sh = th + (vfrac ? NTAP - 1 : 0);
    const uint8_t *src = (const uint8_t *)job.src;

    // stage the source window: rows y0-lv .., columns x0-lh ..
    for (int i = threadIdx.x; i < sw * sh; i += blockDim.x) {
        const int r = i / sw, c = i - r * sw;
        s[r][c] = ((const px_t *)(src + (ptrdiff_t)(y0 + r - lv) * job.src_stride))[x0 + c - lh];
    }
    __syncthreads();
    int8_t hf[8], vf[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { hf[k] = job.hf[k]; vf[k] = job.vf[k]; }

    if (hfrac && vfrac) {
        // horizontal pass into the int16 plane (narrowing store like the reference's tmp_array, :135-141)
        for (int i = threadIdx.x; i < tw * sh; i += blockDim.x) {
            const int r = i / tw, c = i - r * tw;
            t[r][c] = (int16_t)(fir_lds<NTAP>(hf, &s[r][c], 1) >> (BD - 8));
        }
        __syncthreads();
    }

    const int kind = job.kind;
    const int sh_uni = 14 - BD, off_uni = 1 << (sh_uni - 1);
    const int sh_w = job.denom + 14 - BD, off_w = 1 << (sh_w - 1);
    const int ox = job.ox * (1 << (BD - 8)), wx = job.wx;
    uint8_t *dst = (uint8_t *)job.dst;
    for (int i = threadIdx.x; i < tw * th; i += blockDim.x) {
        const int r = i / tw, c = i - r * tw;
        int val;
        if (hfrac && vfrac)
            val = fir_lds<NTAP>(vf, &t[r][c], kMcTile) >> 6;
        else if (hfrac)
            val = fir_lds<NTAP>(hf, &s[r][c], 1) >> (BD - 8);
        else if (vfrac)
            val = fir_lds<NTAP>(vf, &s[r][c], kMcSrcW) >> (BD - 8);
        else
            val = s[r][c] << (14 - BD);
        uint8_t *drow = dst + (ptrdiff_t)(y0 + r) * job.dst_stride;
        if (kind == 0)
            ((int16_t *)drow)[x0 + c] = (int16_t)val;
        else if (kind == 1)
            st_px<BD>(drow, x0 + c, (hfrac || vfrac) ? clip_px<BD>((val + off_uni) >> sh_uni) : (int)s[r][c]);
        else
            st_px<BD>(drow, x0 + c, clip_px<BD>(((val * wx + off_w) >> sh_w) + ox));
    }
}

// grid: (tiles_max, n_jobs); a job of w x h owns ceil(w/32) * ceil(h/32) tiles, surplus workgroups exit at once
template <int BD>
__global__ __launch_bounds__(256) void mc_kernel(const vvc355_mc_job *__restrict__ jobs)
{
    __shared__ uint16_t s[kMcTile + 7][kMcSrcW];
    __shared__ int16_t t[kMcTile + 7][kMcTile];
    const vvc355_mc_job job = load_uniform(jobs + (blockIdx.y));
    const int tiles_x = (job.w + kMcTile - 1) / kMcTile, tiles_y = (job.h + kMcTile - 1) / kMcTile;
    if ((int)blockIdx.x >= tiles_x * tiles_y)
        return;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int x0 = tx * kMcTile, y0 = ty * kMcTile;
    const int tw = min(kMcTile, job.w - x0), th = min(kMcTile, job.h - y0);
    if (job.chroma)
        mc_tile<BD, 4>(job, x0, y0, tw, th, s, t);
    else
        mc_tile<BD, 8>(job, x0, y0, tw, th, s, t);
}

// ------------------------------------------------------------------------------------------------ blends

// mode 0 avg, 1 w_avg, 2 put_ciip (src0 = inter pixels), 3 put_gpm (aux = weights)
template <int BD>
__global__ __launch_bounds__(256) void blend_kernel(const vvc355_blend_job *__restrict__ jobs)
{
    const vvc355_blend_job job = load_uniform(jobs + (blockIdx.y));
    const int w = job.w, h = job.h, mode = job.mode;
    uint8_t *dst = (uint8_t *)job.dst;
    const int s0 = job.src0_stride, s1 = job.src1_stride;
    int shift, off;
    if (mode == 0)      { shift = max(3, 15 - BD); off = 1 << (shift - 1); }
    else if (mode == 1) { shift = job.denom + max(3, 15 - BD); off = (((job.o0 + job.o1) << (BD - 8)) + 1) << (shift - 1); }
    else                { shift = max(5, 17 - BD); off = 1 << (shift - 1); }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < w * h; i += gridDim.x * blockDim.x) {
        const int y = i / w, x = i - y * w;
        uint8_t *drow = dst + (ptrdiff_t)y * job.dst_stride;
        if (mode == 2) {
            const int iw = job.w0;          // intra weight
            const int inter = ld_px<BD>((const uint8_t *)job.src0 + (ptrdiff_t)y * s0, x);
            st_px<BD>(drow, x, (ld_px<BD>(drow, x) * iw + inter * (4 - iw) + 2) >> 2);
            continue;
        }
        const int a = ((const int16_t *)((const uint8_t *)job.src0 + (ptrdiff_t)y * s0))[x];
        const int b = ((const int16_t *)((const uint8_t *)job.src1 + (ptrdiff_t)y * s1))[x];
        int v;
        if (mode == 0)      v = (a + b + off) >> shift;
        else if (mode == 1) v = (a * job.w0 + b * job.w1 + off) >> shift;
        else {
            const int wgt = ((const uint8_t *)job.aux)[(ptrdiff_t)y * job.step_y + (ptrdiff_t)x * job.step_x];
            v = (a * wgt + b * (8 - wgt) + off) >> shift;
        }
        st_px<BD>(drow, x, clip_px<BD>(v));
    }
}

// ------------------------------------------------------------------------------------------------ BDOF

// One workgroup per block (<= 16x16).  job.src0/src1: int16 planes (element stride job.src0_stride/2) whose one-sample ring
// was filled by bdof_fetch_samples; both are padded in place exactly like the reference does (vvc_inter_template.c:297-302).
template <int BD>
__global__ __launch_bounds__(256) void bdof_kernel(const vvc355_blend_job *__restrict__ jobs)
{
    constexpr int GS = 18;
    __shared__ int16_t smp[2][GS][GS];        // samples incl. ring, (1,1) = block origin
    __shared__ int16_t gh[2][GS][GS], gv[2][GS][GS];
    const vvc355_blend_job job = load_uniform(jobs + (blockIdx.x));
    const int w = job.w, h = job.h, tid = threadIdx.x;
    int16_t *src[2] = { (int16_t *)job.src0, (int16_t *)job.src1 };
    const int stride[2] = { job.src0_stride >> 1, job.src1_stride >> 1 };

    for (int i = tid; i < 2 * (h + 2) * (w + 2); i += blockDim.x) {
        const int p = i / ((h + 2) * (w + 2)), rem = i - p * (h + 2) * (w + 2);
        const int y = rem / (w + 2), x = rem - y * (w + 2);
        smp[p][y][x] = src[p][(ptrdiff_t)(y - 1) * stride[p] + (x - 1)];
    }
    __syncthreads();
    // gradients of the interior from the fetched ring (prof_grad_filter, :135)
    for (int i = tid; i < 2 * h * w; i += blockDim.x) {
        const int p = i / (h * w), rem = i - p * h * w;
        const int y = rem / w + 1, x = rem % w + 1;
        gh[p][y][x] = (int16_t)((smp[p][y][x + 1] >> 6) - (smp[p][y][x - 1] >> 6));
        gv[p][y][x] = (int16_t)((smp[p][y + 1][x] >> 6) - (smp[p][y - 1][x] >> 6));
    }
    __syncthreads();
    // replicate rings: left/right columns first, then full top/bottom rows (pad_int16, vvcdsp.c:29)
    for (int i = tid; i < 2 * h; i += blockDim.x) {
        const int p = i / h, y = i - p * h + 1;
        gh[p][y][0] = gh[p][y][1]; gh[p][y][w + 1] = gh[p][y][w];
        gv[p][y][0] = gv[p][y][1]; gv[p][y][w + 1] = gv[p][y][w];
        smp[p][y][0] = smp[p][y][1]; smp[p][y][w + 1] = smp[p][y][w];
    }
    __syncthreads();
    for (int i = tid; i < 2 * (w + 2); i += blockDim.x) {
        const int p = i / (w + 2), x = i - p * (w + 2);
        gh[p][0][x] = gh[p][1][x]; gh[p][h + 1][x] = gh[p][h][x];
        gv[p][0][x] = gv[p][1][x]; gv[p][h + 1][x] = gv[p][h][x];
        smp[p][0][x] = smp[p][1][x]; smp[p][h + 1][x] = smp[p][h][x];
    }
    __syncthreads();
    // the padding of src0/src1 is visible to the caller: write the ring back
    for (int i = tid; i < 2 * (h + 2) * (w + 2); i += blockDim.x) {
        const int p = i / ((h + 2) * (w + 2)), rem = i - p * (h + 2) * (w + 2);
        const int y = rem / (w + 2), x = rem - y * (w + 2);
        if (y == 0 || y == h + 1 || x == 0 || x == w + 1)
            src[p][(ptrdiff_t)(y - 1) * stride[p] + (x - 1)] = smp[p][y][x];
    }

    // thread -> pixel (py, px); the 16 lanes of a 4x4 sub-block sit in one wave (lane groups of 16)
    const int sb = tid >> 4, l = tid & 15;
    const int sbw = w >> 2, nsb = sbw * (h >> 2);
    if (sb >= nsb)
        return;                      // whole 16-lane groups leave together; shuffles below stay inside a group
    const int by = (sb / sbw) * 4, bx = (sb % sbw) * 4;
    int sgx2 = 0, sgy2 = 0, sgxgy = 0, sgxdi = 0, sgydi = 0;
    for (int e = l; e < 36; e += 16) {
        const int j = e / 6, i = e - j * 6;
        const int y = by + j, x = bx + i;                 // padded-plane coordinates of window element (j, i)
        const int diff = (smp[0][y][x] >> 4) - (smp[1][y][x] >> 4);
        const int th = (gh[0][y][x] + gh[1][y][x]) >> 1;
        const int tv = (gv[0][y][x] + gv[1][y][x]) >> 1;
        sgx2 += abs(th);
        sgy2 += abs(tv);
        sgxgy += sign_of(tv) * th;
        sgxdi += -sign_of(th) * diff;
        sgydi += -sign_of(tv) * diff;
    }
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) {
        sgx2 += __shfl_xor(sgx2, m, 16);
        sgy2 += __shfl_xor(sgy2, m, 16);
        sgxgy += __shfl_xor(sgxgy, m, 16);
        sgxdi += __shfl_xor(sgxdi, m, 16);
        sgydi += __shfl_xor(sgydi, m, 16);
    }
    const int vx = sgx2 > 0 ? clip3((sgxdi * 4) >> ilog2(sgx2), -15, 15) : 0;
    const int vy = sgy2 > 0 ? clip3(((sgydi * 4) - ((vx * sgxgy) >> 1)) >> ilog2(sgy2), -15, 15) : 0;
    const int py = by + (l >> 2), px = bx + (l & 3);
    const int y = py + 1, x = px + 1;
    const int sh = 15 - BD, off = 1 << (sh - 1);
    const int corr = vx * (gh[0][y][x] - gh[1][y][x]) + vy * (gv[0][y][x] - gv[1][y][x]);
    st_px<BD>((uint8_t *)job.dst + (ptrdiff_t)py * job.dst_stride, px, clip_px<BD>((smp[0][y][x] + off + smp[1][y][x] + corr) >> sh));
}

// ------------------------------------------------------------------------------------------------ PROF

// prof_grad_filter slot (:135) on an arbitrary w x h int16 block (pad = 0/1)
__global__ void prof_grad_kernel(int16_t *gh, int16_t *gv, int gstride, const int16_t *src, int sstride, int w, int h, int pad)
{
    // single workgroup; interior first, then the replicated ring
    int16_t *gh0 = gh + pad * (1 + gstride), *gv0 = gv + pad * (1 + gstride);
    for (int i = threadIdx.x; i < w * h; i += blockDim.x) {
        const int y = i / w, x = i - y * w;
        const int16_t *p = src + (ptrdiff_t)y * sstride + x;
        gh0[y * gstride + x] = (int16_t)((p[1] >> 6) - (p[-1] >> 6));
        gv0[y * gstride + x] = (int16_t)((p[sstride] >> 6) - (p[-sstride] >> 6));
    }
    if (!pad)
        return;
    __syncthreads();
    for (int i = threadIdx.x; i < h; i += blockDim.x) {
        gh0[i * gstride - 1] = gh0[i * gstride]; gh0[i * gstride + w] = gh0[i * gstride + w - 1];
        gv0[i * gstride - 1] = gv0[i * gstride]; gv0[i * gstride + w] = gv0[i * gstride + w - 1];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < w + 2; i += blockDim.x) {
        const int x = i - 1;
        gh0[-gstride + x] = gh0[x]; gh0[h * gstride + x] = gh0[(h - 1) * gstride + x];
        gv0[-gstride + x] = gv0[x]; gv0[h * gstride + x] = gv0[(h - 1) * gstride + x];
    }
}

// apply_prof / apply_prof_uni / apply_prof_uni_w (:160,:181,:210): 16 lanes per 4x4 job.
// job.src0 = int16 prediction (ring readable), job.src1 = diff_mv_x[16], job.aux = diff_mv_y[16]; mode 0 int16 / 1 uni / 2 uni_w
template <int BD>
__global__ __launch_bounds__(256) void prof_kernel(const vvc355_blend_job *__restrict__ jobs, int n_jobs)
{
    const int j = blockIdx.x * 16 + (threadIdx.x >> 4), l = threadIdx.x & 15;
    if (j >= n_jobs)
        return;
    const vvc355_blend_job job = jobs[j];
    const int y = l >> 2, x = l & 3;
    const int ss = job.src0_stride >> 1;
    const int16_t *p = (const int16_t *)job.src0 + (ptrdiff_t)y * ss + x;
    const int g_h = (int16_t)((p[1] >> 6) - (p[-1] >> 6));
    const int g_v = (int16_t)((p[ss] >> 6) - (p[-ss] >> 6));
    const int limit = 1 << max(13, BD + 1);
    const int di = g_h * ((const int16_t *)job.src1)[l] + g_v * ((const int16_t *)job.aux)[l];
    const int val = p[0] + clip3(di, -limit, limit - 1);
    uint8_t *drow = (uint8_t *)job.dst + (ptrdiff_t)y * job.dst_stride;
    if (job.mode == 0) {
        ((int16_t *)drow)[x] = (int16_t)val;
    } else if (job.mode == 1) {
        const int sh = 14 - BD;
        st_px<BD>(drow, x, clip_px<BD>((val + (1 << (sh - 1))) >> sh));
    } else {
        const int sh = job.denom + max(2, 14 - BD);
        st_px<BD>(drow, x, clip_px<BD>(((val * job.w0 + (1 << (sh - 1))) >> sh) + job.o0 * (1 << (BD - 8))));
    }
}

// ------------------------------------------------------------------------------------------------ integer-sample ring fetch

// bdof_fetch_samples / fetch_samples (:101,:130): ring of (w+2)x(h+2) minus interior around an int16 block.
// job.dst = int16 block origin (stride job.dst_stride bytes), job.src0 = pixel plane at the block's integer position,
// job.w0 / job.w1 = x_frac / y_frac.
template <int BD>
__global__ void fetch_ring_kernel(const vvc355_blend_job *__restrict__ jobs)
{
    const vvc355_blend_job job = load_uniform(jobs + (blockIdx.x));
    const int w = job.w, h = job.h;
    const int x_off = (job.w0 >> 3) - 1, y_off = (job.w1 >> 3) - 1;
    const int ds = job.dst_stride >> 1;
    const int n = 2 * (w + 2) + 2 * h;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        int x, y;
        if (i < w + 2)            { y = -1; x = i - 1; }
        else if (i < 2 * (w + 2)) { y = h;  x = i - (w + 2) - 1; }
        else                      { const int k = i - 2 * (w + 2); y = k >> 1; x = (k & 1) ? w : -1; }
        const int v = ld_px<BD>((const uint8_t *)job.src0 + (ptrdiff_t)(y + 1 + y_off) * job.src0_stride, x + 1 + x_off);
        ((int16_t *)job.dst)[(ptrdiff_t)y * ds + x] = (int16_t)(v << (14 - BD));
    }
}

// ------------------------------------------------------------------------------------------------ DMVR

// dmvr[vfrac][hfrac] (:324-413); job.hf[0] = mx, job.vf[0] = my (0..15)
template <int BD>
__global__ __launch_bounds__(256) void dmvr_kernel(const vvc355_mc_job *__restrict__ jobs)
{
    const vvc355_mc_job job = load_uniform(jobs + (blockIdx.y));
    const int w = job.w, h = job.h, mx = job.hf[0], my = job.vf[0];
    const int hfrac = job.hfrac, vfrac = job.vfrac;
    const uint8_t *src = (const uint8_t *)job.src;
    const ptrdiff_t ss = job.src_stride / (ptrdiff_t)sizeof(typename Px<BD>::type);
    const int sh1 = BD - 6, off1 = 1 << (sh1 - 1);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < w * h; i += gridDim.x * blockDim.x) {
        const int y = i / w, x = i - y * w;
        const ptrdiff_t o = (ptrdiff_t)y * ss + x;
        int v;
        if (hfrac && vfrac) {
            const int t0 = (int16_t)(((16 - mx) * ld_px<BD>(src, o) + mx * ld_px<BD>(src, o + 1) + off1) >> sh1);
            const int t1 = (int16_t)(((16 - mx) * ld_px<BD>(src, o + ss) + mx * ld_px<BD>(src, o + ss + 1) + off1) >> sh1);
            v = ((16 - my) * t0 + my * t1 + 8) >> 4;
        } else if (hfrac) {
            v = ((16 - mx) * ld_px<BD>(src, o) + mx * ld_px<BD>(src, o + 1) + off1) >> sh1;
        } else if (vfrac) {
            v = ((16 - my) * ld_px<BD>(src, o) + my * ld_px<BD>(src, o + ss) + off1) >> sh1;
        } else if (BD > 10) {
            v = (ld_px<BD>(src, o) + (1 << (BD - 11))) >> (BD - 10);
        } else {
            v = ld_px<BD>(src, o) << (10 - BD);
        }
        ((int16_t *)((uint8_t *)job.dst + (ptrdiff_t)y * job.dst_stride))[x] = (int16_t)v;
    }
}

// vvc_sad (vvcdsp.c:49): one wave per job; out[job] = sum |a - b| over every other row
__global__ __launch_bounds__(64) void sad_kernel(const vvc355_blend_job *__restrict__ jobs, int *out)
{
    const vvc355_blend_job job = load_uniform(jobs + (blockIdx.x));
    const int s0 = job.src0_stride >> 1, s1 = job.src1_stride >> 1;
    const int dx = job.w0 - 2, dy = job.w1 - 2;
    const int16_t *a = (const int16_t *)job.src0 + (ptrdiff_t)(2 + dy) * s0 + 2 + dx;
    const int16_t *b = (const int16_t *)job.src1 + (ptrdiff_t)(2 - dy) * s1 + 2 - dx;
    const int w = job.w, rows = (job.h + 1) >> 1;
    int acc = 0;
    for (int i = threadIdx.x; i < w * rows; i += 64) {
        const int r = i / w, x = i - r * w;
        acc += abs((int)a[(ptrdiff_t)2 * r * s0 + x] - (int)b[(ptrdiff_t)2 * r * s1 + x]);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
        acc += __shfl_xor(acc, m, 64);
    if (threadIdx.x == 0)
        out[blockIdx.x] = acc;
}

// ------------------------------------------------------------------------------------------------ launchers

static void launch_mc(int bd, const vvc355_mc_job *jobs, int n, int max_w, int max_h, hipStream_t st)
{
    if (n <= 0) return;
    const int tiles = ((max_w + kMcTile - 1) / kMcTile) * ((max_h + kMcTile - 1) / kMcTile);
    VVC355_BD_DISPATCH(bd, hipLaunchKernelGGL((mc_kernel<BD>), dim3(tiles, n), dim3(256), 0, st, jobs));
    HIP_CHECK(hipGetLastError());
}

static void launch_blend(int bd, const vvc355_blend_job *jobs, int n, int max_w, int max_h, hipStream_t st)
{
    if (n <= 0) return;
    const int gx = max(1, min(16, (max_w * max_h + 1023) / 1024));
    VVC355_BD_DISPATCH(bd, hipLaunchKernelGGL((blend_kernel<BD>), dim3(gx, n), dim3(256), 0, st, jobs));
    HIP_CHECK(hipGetLastError());
}

static void check_block(const char *what, int w, int h, int max_w, int max_h)
{
    if (w <= 0 || h <= 0 || w > max_w || h > max_h) {
        fprintf(stderr, "vvc_mi355: %s block %dx%d outside the slot's domain (<= %dx%d)\n", what, w, h, max_w, max_h);
        abort();
    }
}

static void fill_mc_job(vvc355_mc_job &job, int chroma, int kind, int vfrac, int hfrac, const int8_t *hf, const int8_t *vf,
                        int w, int h, int denom, int wx, int ox)
{
    const int ntap = chroma ? 4 : 8;
    for (int k = 0; k < ntap; k++) {
        job.hf[k] = (hfrac && hf) ? hf[k] : 0;
        job.vf[k] = (vfrac && vf) ? vf[k] : 0;
    }
    job.w = (int16_t)w; job.h = (int16_t)h;
    job.kind = (uint8_t)kind; job.chroma = (uint8_t)chroma; job.hfrac = (uint8_t)!!hfrac; job.vfrac = (uint8_t)!!vfrac;
    job.denom = (int16_t)denom; job.wx = (int16_t)wx; job.ox = (int16_t)ox;
}

// stage one put-family slot call (host pointers) and launch it
static void slot_mc(int bd, int chroma, int kind, int vfrac, int hfrac, void *dst, ptrdiff_t dst_stride,
                    const uint8_t *src, ptrdiff_t src_stride, int height, int denom, int wx, int ox,
                    const int8_t *hf, const int8_t *vf, int width)
{
    check_block("put", width, height, 128, 128);
    const int px = bd > 8 ? 2 : 1;
    const int lead = chroma ? 1 : 3, trail = chroma ? 2 : 4;
    SlotCall call;
    const Staged s = call.rect(src, src_stride, -(hfrac ? lead : 0) * px, (width + (hfrac ? trail : 0)) * px,
                               -(vfrac ? lead : 0), height + (vfrac ? trail : 0), true, false);
    const Staged d = kind == 0 ? call.rect(dst, VVC355_PB * 2, 0, width * 2, 0, height, false, true)
                               : call.rect(dst, dst_stride, 0, width * px, 0, height, false, true);
    vvc355_mc_job job = {};
    job.dst = (uint64_t)d.dev; job.src = (uint64_t)s.dev;
    job.dst_stride = (int32_t)d.pitch; job.src_stride = (int32_t)s.pitch;
    fill_mc_job(job, chroma, kind, vfrac, hfrac, hf, vf, width, height, denom, wx, ox);
    launch_mc(bd, call.upload(&job, 1), 1, width, height, call.stream());
}

} // namespace vvc355

using namespace vvc355;

// =============================================================================================== C ABI

extern "C" {

void vvc355_mc_batch(void *stream, int bd, const vvc355_mc_job *jobs_dev, int n_jobs, int max_w, int max_h)
{
    if (n_jobs <= 0) return;
    launch_mc(bd, jobs_dev, n_jobs, max_w, max_h, (hipStream_t)stream);
}

void vvc355_blend_batch(void *stream, int bd, const vvc355_blend_job *jobs_dev, int n_jobs, int max_w, int max_h)
{
    if (n_jobs <= 0) return;
    launch_blend(bd, jobs_dev, n_jobs, max_w, max_h, (hipStream_t)stream);
}

void vvc355_bdof_batch(void *stream, int bd, const vvc355_blend_job *jobs_dev, int n_jobs)
{
    if (n_jobs <= 0) return;
    VVC355_BD_DISPATCH(bd, hipLaunchKernelGGL((bdof_kernel<BD>), dim3(n_jobs), dim3(256), 0, (hipStream_t)stream, jobs_dev));
    HIP_CHECK(hipGetLastError());
}

void vvc355_put(int bd, int chroma, int vfrac, int hfrac, int16_t *dst, const uint8_t *src, ptrdiff_t src_stride,
                int height, const int8_t *hf, const int8_t *vf, int width)
{
    slot_mc(bd, chroma, 0, vfrac, hfrac, dst, 0, src, src_stride, height, 0, 0, 0, hf, vf, width);
}

void vvc355_put_uni(int bd, int chroma, int vfrac, int hfrac, uint8_t *dst, ptrdiff_t dst_stride,
                    const uint8_t *src, ptrdiff_t src_stride, int height, const int8_t *hf, const int8_t *vf, int width)
{
    slot_mc(bd, chroma, 1, vfrac, hfrac, dst, dst_stride, src, src_stride, height, 0, 0, 0, hf, vf, width);
}

void vvc355_put_uni_w(int bd, int chroma, int vfrac, int hfrac, uint8_t *dst, ptrdiff_t dst_stride,
                      const uint8_t *src, ptrdiff_t src_stride, int height, int denom, int wx, int ox,
                      const int8_t *hf, const int8_t *vf, int width)
{
    slot_mc(bd, chroma, 2, vfrac, hfrac, dst, dst_stride, src, src_stride, height, denom, wx, ox, hf, vf, width);
}

static void slot_blend(int bd, int mode, uint8_t *dst, ptrdiff_t dst_stride, const int16_t *src0, const int16_t *src1,
                       int width, int height, int denom, int w0, int w1, int o0, int o1)
{
    check_block("avg", width, height, 128, 128);
    const int px = bd > 8 ? 2 : 1;
    SlotCall call;
    const Staged a = call.rect(src0, VVC355_PB * 2, 0, width * 2, 0, height, true, false);
    const Staged b = call.rect(src1, VVC355_PB * 2, 0, width * 2, 0, height, true, false);
    const Staged d = call.rect(dst, dst_stride, 0, width * px, 0, height, false, true);
    vvc355_blend_job job = {};
    job.dst = (uint64_t)d.dev; job.src0 = (uint64_t)a.dev; job.src1 = (uint64_t)b.dev;
    job.dst_stride = (int32_t)d.pitch; job.src0_stride = (int32_t)a.pitch; job.src1_stride = (int32_t)b.pitch;
    job.w = (int16_t)width; job.h = (int16_t)height; job.mode = (int16_t)mode;
    job.denom = (int16_t)denom; job.w0 = (int16_t)w0; job.w1 = (int16_t)w1; job.o0 = (int16_t)o0; job.o1 = (int16_t)o1;
    launch_blend(bd, call.upload(&job, 1), 1, width, height, call.stream());
}

void vvc355_avg(int bd, uint8_t *dst, ptrdiff_t dst_stride, const int16_t *src0, const int16_t *src1, int width, int height)
{
    slot_blend(bd, 0, dst, dst_stride, src0, src1, width, height, 0, 0, 0, 0, 0);
}

void vvc355_w_avg(int bd, uint8_t *dst, ptrdiff_t dst_stride, const int16_t *src0, const int16_t *src1, int width, int height,
                  int denom, int w0, int w1, int o0, int o1)
{
    slot_blend(bd, 1, dst, dst_stride, src0, src1, width, height, denom, w0, w1, o0, o1);
}

void vvc355_put_ciip(int bd, uint8_t *dst, ptrdiff_t dst_stride, int width, int height,
                     const uint8_t *inter, ptrdiff_t inter_stride, int intra_weight)
{
    check_block("put_ciip", width, height, 128, 128);
    const int px = bd > 8 ? 2 : 1;
    SlotCall call;
    const Staged a = call.rect(inter, inter_stride, 0, width * px, 0, height, true, false);
    const Staged d = call.rect(dst, dst_stride, 0, width * px, 0, height, true, true);
    vvc355_blend_job job = {};
    job.dst = (uint64_t)d.dev; job.src0 = (uint64_t)a.dev;
    job.dst_stride = (int32_t)d.pitch; job.src0_stride = (int32_t)a.pitch;
    job.w = (int16_t)width; job.h = (int16_t)height; job.mode = 2; job.w0 = (int16_t)intra_weight;
    launch_blend(bd, call.upload(&job, 1), 1, width, height, call.stream());
}

void vvc355_put_gpm(int bd, uint8_t *dst, ptrdiff_t dst_stride, int width, int height,
                    const int16_t *src0, const int16_t *src1, const uint8_t *weights, int step_x, int step_y)
{
    check_block("put_gpm", width, height, 128, 128);
    const int px = bd > 8 ? 2 : 1;
    // byte extent of the weight mask touched by (x*step_x + y*step_y), steps may be negative
    const ptrdiff_t ex = (ptrdiff_t)(width - 1) * step_x, ey = (ptrdiff_t)(height - 1) * step_y;
    const ptrdiff_t lo = (ex < 0 ? ex : 0) + (ey < 0 ? ey : 0), hi = (ex > 0 ? ex : 0) + (ey > 0 ? ey : 0);
    SlotCall call;
    const Staged a = call.rect(src0, VVC355_PB * 2, 0, width * 2, 0, height, true, false);
    const Staged b = call.rect(src1, VVC355_PB * 2, 0, width * 2, 0, height, true, false);
    const Staged d = call.rect(dst, dst_stride, 0, width * px, 0, height, false, true);
    const uint8_t *wd = (const uint8_t *)call.linear(weights + lo, (size_t)(hi - lo + 1), true, false) - lo;
    vvc355_blend_job job = {};
    job.dst = (uint64_t)d.dev; job.src0 = (uint64_t)a.dev; job.src1 = (uint64_t)b.dev; job.aux = (uint64_t)wd;
    job.dst_stride = (int32_t)d.pitch; job.src0_stride = (int32_t)a.pitch; job.src1_stride = (int32_t)b.pitch;
    job.w = (int16_t)width; job.h = (int16_t)height; job.mode = 3; job.step_x = step_x; job.step_y = step_y;
    launch_blend(bd, call.upload(&job, 1), 1, width, height, call.stream());
}

void vvc355_bdof_fetch_samples(int bd, int16_t *dst, const uint8_t *src, ptrdiff_t src_stride, int x_frac, int y_frac,
                               int width, int height)
{
    check_block("bdof_fetch_samples", width, height, 126, 126);
    const int px = bd > 8 ? 2 : 1;
    const int x_off = (x_frac >> 3) - 1, y_off = (y_frac >> 3) - 1;
    SlotCall call;
    // the int16 block's interior must survive: stage it in and out together with the ring
    const Staged d = call.rect(dst, VVC355_PB * 2, -2, (width + 1) * 2, -1, height + 1, true, true);
    const Staged s = call.rect(src, src_stride, x_off * px, (x_off + width + 2) * px, y_off, y_off + height + 2, true, false);
    vvc355_blend_job job = {};
    job.dst = (uint64_t)d.dev; job.src0 = (uint64_t)s.dev;
    job.dst_stride = (int32_t)d.pitch; job.src0_stride = (int32_t)s.pitch;
    job.w = (int16_t)width; job.h = (int16_t)height; job.w0 = (int16_t)x_frac; job.w1 = (int16_t)y_frac;
    const vvc355_blend_job *jd = call.upload(&job, 1);
    VVC355_BD_DISPATCH(bd, hipLaunchKernelGGL((fetch_ring_kernel<BD>), dim3(1), dim3(256), 0, call.stream(), jd));
    HIP_CHECK(hipGetLastError());
}

void vvc355_fetch_samples(int bd, int16_t *dst, const uint8_t *src, ptrdiff_t src_stride, int x_frac, int y_frac)
{
    vvc355_bdof_fetch_samples(bd, dst, src, src_stride, x_frac, y_frac, 4, 4);
}

void vvc355_prof_grad_filter(int bd, int16_t *gradient_h, int16_t *gradient_v, ptrdiff_t gradient_stride,
                             const int16_t *src, ptrdiff_t src_stride, int width, int height, int pad)
{
    (void)bd;
    check_block("prof_grad_filter", width, height, 128, 128);
    SlotCall call;
    const int gw = width + 2 * pad, gh_rows = height + 2 * pad;
    const Staged g0 = call.rect(gradient_h, gradient_stride * 2, 0, gw * 2, 0, gh_rows, true, true);
    const Staged g1 = call.rect(gradient_v, gradient_stride * 2, 0, gw * 2, 0, gh_rows, true, true);
    const Staged s = call.rect(src, src_stride * 2, -2, (width + 1) * 2, -1, height + 1, true, false);
    hipLaunchKernelGGL(prof_grad_kernel, dim3(1), dim3(256), 0, call.stream(), (int16_t *)g0.dev, (int16_t *)g1.dev,
                       (int)(g0.pitch >> 1), (const int16_t *)s.dev, (int)(s.pitch >> 1), width, height, pad);
    HIP_CHECK(hipGetLastError());
    // both gradient planes share one pitch by construction (same geometry)
    if (g0.pitch != g1.pitch) { fprintf(stderr, "vvc_mi355: internal pitch mismatch\n"); abort(); }
}

static void slot_prof(int bd, int mode, void *dst, ptrdiff_t dst_stride, const int16_t *src,
                      const int16_t *dmx, const int16_t *dmy, int denom, int wx, int ox)
{
    const int px = bd > 8 ? 2 : 1;
    SlotCall call;
    const Staged s = call.rect(src, VVC355_PB * 2, -2, 5 * 2, -1, 5, true, false);
    const Staged d = mode == 0 ? call.rect(dst, VVC355_PB * 2, 0, 4 * 2, 0, 4, false, true)
                               : call.rect(dst, dst_stride, 0, 4 * px, 0, 4, false, true);
    vvc355_blend_job job = {};
    job.dst = (uint64_t)d.dev; job.src0 = (uint64_t)s.dev;
    job.src1 = (uint64_t)call.linear(dmx, 32, true, false); job.aux = (uint64_t)call.linear(dmy, 32, true, false);
    job.dst_stride = (int32_t)d.pitch; job.src0_stride = (int32_t)s.pitch;
    job.w = job.h = 4; job.mode = (int16_t)mode; job.denom = (int16_t)denom; job.w0 = (int16_t)wx; job.o0 = (int16_t)ox;
    const vvc355_blend_job *jd = call.upload(&job, 1);
    VVC355_BD_DISPATCH(bd, hipLaunchKernelGGL((prof_kernel<BD>), dim3(1), dim3(256), 0, call.stream(), jd, 1));
    HIP_CHECK(hipGetLastError());
}

void vvc355_apply_prof(int bd, int16_t *dst, const int16_t *src, const int16_t *diff_mv_x, const int16_t *diff_mv_y)
{
    slot_prof(bd, 0, dst, 0, src, diff_mv_x, diff_mv_y, 0, 0, 0);
}

void vvc355_apply_prof_uni(int bd, uint8_t *dst, ptrdiff_t dst_stride, const int16_t *src,
                           const int16_t *diff_mv_x, const int16_t *diff_mv_y)
{
    slot_prof(bd, 1, dst, dst_stride, src, diff_mv_x, diff_mv_y, 0, 0, 0);
}

void vvc355_apply_prof_uni_w(int bd, uint8_t *dst, ptrdiff_t dst_stride, const int16_t *src,
                             const int16_t *diff_mv_x, const int16_t *diff_mv_y, int denom, int wx, int ox)
{
    slot_prof(bd, 2, dst, dst_stride, src, diff_mv_x, diff_mv_y, denom, wx, ox);
}

void vvc355_apply_bdof(int bd, uint8_t *dst, ptrdiff_t dst_stride, int16_t *src0, int16_t *src1, int block_w, int block_h)
{
    check_block("apply_bdof", block_w, block_h, 16, 16);
    if ((block_w & 3) || (block_h & 3)) { fprintf(stderr, "vvc_mi355: apply_bdof needs multiples of 4\n"); abort(); }
    const int px = bd > 8 ? 2 : 1;
    SlotCall call;
    const Staged a = call.rect(src0, VVC355_PB * 2, -2, (block_w + 1) * 2, -1, block_h + 1, true, true);
    const Staged b = call.rect(src1, VVC355_PB * 2, -2, (block_w + 1) * 2, -1, block_h + 1, true, true);
    const Staged d = call.rect(dst, dst_stride, 0, block_w * px, 0, block_h, false, true);
    vvc355_blend_job job = {};
    job.dst = (uint64_t)d.dev; job.src0 = (uint64_t)a.dev; job.src1 = (uint64_t)b.dev;
    job.dst_stride = (int32_t)d.pitch; job.src0_stride = (int32_t)a.pitch; job.src1_stride = (int32_t)b.pitch;
    job.w = (int16_t)block_w; job.h = (int16_t)block_h;
    const vvc355_blend_job *jd = call.upload(&job, 1);
    VVC355_BD_DISPATCH(bd, hipLaunchKernelGGL((bdof_kernel<BD>), dim3(1), dim3(256), 0, call.stream(), jd));
    HIP_CHECK(hipGetLastError());
}

int vvc355_sad(const int16_t *src0, const int16_t *src1, int dx, int dy, int block_w, int block_h)
{
    check_block("sad", block_w, block_h, 128, 128);
    int result = 0;
    {
        SlotCall call;
        // operands: (w+4)x(h+4) bilinear planes; rows (2+dy-2 ..) stay inside [0, h+4)
        const Staged a = call.rect(src0, VVC355_PB * 2, 0, (block_w + 4) * 2, 0, block_h + 4, true, false);
        const Staged b = call.rect(src1, VVC355_PB * 2, 0, (block_w + 4) * 2, 0, block_h + 4, true, false);
        int *out = (int *)call.linear(&result, sizeof(int), false, true);
        vvc355_blend_job job = {};
        job.src0 = (uint64_t)a.dev; job.src1 = (uint64_t)b.dev;
        job.src0_stride = (int32_t)a.pitch; job.src1_stride = (int32_t)b.pitch;
        job.w = (int16_t)block_w; job.h = (int16_t)block_h; job.w0 = (int16_t)dx; job.w1 = (int16_t)dy;
        hipLaunchKernelGGL(sad_kernel, dim3(1), dim3(64), 0, call.stream(), call.upload(&job, 1), out);
        HIP_CHECK(hipGetLastError());
    }   // ~SlotCall copies `result` back and synchronises
    return result;
}

void vvc355_dmvr(int bd, int vfrac, int hfrac, int16_t *dst, const uint8_t *src, ptrdiff_t src_stride, int height,
                 intptr_t mx, intptr_t my, int width)
{
    check_block("dmvr", width, height, 128, 128);
    const int px = bd > 8 ? 2 : 1;
    SlotCall call;
    const Staged s = call.rect(src, src_stride, 0, (width + (hfrac ? 1 : 0)) * px, 0, height + (vfrac ? 1 : 0), true, false);
    const Staged d = call.rect(dst, VVC355_PB * 2, 0, width * 2, 0, height, false, true);
    vvc355_mc_job job = {};
    job.dst = (uint64_t)d.dev; job.src = (uint64_t)s.dev;
    job.dst_stride = (int32_t)d.pitch; job.src_stride = (int32_t)s.pitch;
    job.w = (int16_t)width; job.h = (int16_t)height; job.hfrac = (uint8_t)!!hfrac; job.vfrac = (uint8_t)!!vfrac;
    job.hf[0] = (int8_t)mx; job.vf[0] = (int8_t)my;
    const vvc355_mc_job *jd = call.upload(&job, 1);
    const int gx = (width * height + 1023) / 1024;
    VVC355_BD_DISPATCH(bd, hipLaunchKernelGGL((dmvr_kernel<BD>), dim3(gx, 1), dim3(256), 0, call.stream(), jd));
    HIP_CHECK(hipGetLastError());
}

} // extern "C"
