// ALF (adaptive loop filter) kernels for gfx950: luma 7x7 diamond (per-4x4 coefficient sets, or fused with the
// block classification + coefficient gather), chroma 5x5 diamond, cross-component ALF, and the stand-alone
// classify / recon_coeff_and_clip slots.
//
// Reference behaviour: libavcodec/vvc/vvc_filter_template.c:38-408 (alf_clip :38, alf_filter_luma :43,
// alf_filter_chroma :137, alf_filter_cc :223, alf_get_idx :270, alf_classify :299, alf_recon_coeff_and_clip :383)
// and the caller's padding rules libavcodec/vvc/vvc_filter.c:1105-1137 (alf_prepare_buffer = clamp-to-edge).
//
// Layout: one workgroup (256 lanes) owns a 128-wide x 32-tall strip of one job rectangle.  The strip plus a
// 3-sample apron is staged through LDS as uint16 with 16-byte global loads; every lane then owns ONE 4x4 block
// (the unit that shares a coefficient set), pulls its 10x12-sample window into registers with aligned
// ds_read_b64, and classifies + filters from registers.  Rows next to the virtual boundary take a slower LDS path.
#include "common.hpp"
#include "runtime.hpp"
#include "../../include/vvc_mi355.h"

namespace vvc355 {

static constexpr int kStripH  = 32;            // rows per workgroup
static constexpr int kTileW   = 144;           // LDS row: columns -8 .. 135  (index = column + 8)
static constexpr int kTileH   = kStripH + 6;   // rows -3 .. 34
static constexpr int kColOff  = 8;

// fixed filter sets and class maps for the stage driver (generated from the reference's data tables, tools/gen_tables.py)
#define VVC355_TABLE(type, name, count) __device__ static const type t_##name[count]
#include "tables.inc"
#undef VVC355_TABLE

__device__ static const uint8_t kAlfPerm[4][12] = {       // vvc_filter_template.c:387-392
    { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11 },
    { 9, 4, 10, 8, 1, 5, 11, 7, 3, 0, 2, 6 },
    { 0, 3, 2, 1, 8, 7, 6, 5, 4, 9, 10, 11 },
    { 9, 8, 10, 4, 3, 7, 11, 5, 1, 0, 2, 6 },
};
__device__ static const uint8_t kAlfVarTab[16] = { 0, 1, 2, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 3, 3, 4 };
// luma diamond taps (dy, dx), paired with (-dy, -dx): vvc_filter_template.c:102-113
__device__ static constexpr int8_t kLumaTap[12][2] = {
    { 3, 0 }, { 2, 1 }, { 2, 0 }, { 2, -1 }, { 1, 2 }, { 1, 1 }, { 1, 0 }, { 1, -1 }, { 1, -2 }, { 0, 3 }, { 0, 2 }, { 0, 1 },
};
// chroma 5x5 diamond taps (dy, dx), each paired with its mirror: (2,0) (1,1) (1,0) (1,-1) (0,2) (0,1) — unrolled in alf_chroma_kernel

// clamp(x, -c, c) in one VALU op (the compiler cannot prove -c <= c, so it would emit max + min)
__device__ __forceinline__ int clamp_sym(int x, int c)
{
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(-c), "v"(c));
    return r;
}
// acc + f * p with 24-bit operands (full-rate v_mad_i32_i24; f is an int16 coefficient, |p| <= 2^(bd+1))
// f * p + acc in one instruction.  Written as __mul24(f, p) + acc the compiler prefers v_mul_i32_i24 + v_add3_u32 (three
// instructions per two taps instead of two); both operands are far below 24 bits here (|f| <= 2^10, |p| <= 2^13).
__device__ __forceinline__ int mad24(int f, int p, int acc)
{
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(f), "v"(p), "v"(acc));
    return r;
}

// rows between row y and the virtual boundary on y's own side (0 = adjacent): taps fold to min(k, dist)
// 8 consecutive samples as four registers of two 16-bit samples each, whatever the storage type
template <int BD> __device__ __forceinline__ void load8_u16(const typename Px<BD>::type *p, uint32_t (&d)[4])
{
    if (BD > 8) {
        const uint4 q = gld<uint4>(p);
        d[0] = q.x; d[1] = q.y; d[2] = q.z; d[3] = q.w;
    } else {
        const uint2 q = gld<uint2>(p);
        d[0] = __builtin_amdgcn_perm(0, q.x, 0x0c010c00u); d[1] = __builtin_amdgcn_perm(0, q.x, 0x0c030c02u);
        d[2] = __builtin_amdgcn_perm(0, q.y, 0x0c010c00u); d[3] = __builtin_amdgcn_perm(0, q.y, 0x0c030c02u);
    }
}

__device__ __forceinline__ int vb_dist(int y, int vb_pos) { return y < vb_pos ? vb_pos - 1 - y : y - vb_pos; }

// ---------------------------------------------------------------------------------------------- tile staging

// Stage rows [y_base-3, y_base+rows+3) x columns [-8, 136) of the job's source rectangle into LDS as uint16,
// replicating at the readable-apron limits (ext_*), which is what the reference's alf_prepare_buffer produces.
template <int BD>
__device__ __forceinline__ void stage_tile(uint16_t (*tile)[kTileW], const vvc355_alf_job &job, int y_base, int rows, int apron)
{
    using px_t = typename Px<BD>::type;
    const uint8_t *src = (const uint8_t *)job.src;
    const int w = job.w, h = job.h;
    const int x_min = -min((int)job.ext_l, apron), x_max = w - 1 + min((int)job.ext_r, apron);
    const int y_min = -min((int)job.ext_t, apron), y_max = h - 1 + min((int)job.ext_b, apron);
    const int nchunk = min(kTileW / 8, (w + 23) >> 3);      // columns -8 .. w+7 only
    const bool aligned = (((uintptr_t)src | (uintptr_t)job.src_stride) & (sizeof(px_t) * 8 - 1)) == 0;
    // (wave-uniform base, unsigned per-lane offset) addressing: the base sits three rows and eight samples before the rectangle
    const uint8_t *src_m = src - 3 * job.src_stride - 8 * (int)sizeof(px_t);

    for (int i = threadIdx.x; i < (rows + 6) * nchunk; i += blockDim.x) {
        const int r = i / nchunk, k = i - r * nchunk;
        const int y = clip3(y_base + r - 3, y_min, y_max);
        const int c0 = k * 8 - kColOff;
        const px_t *row = (const px_t *)(src + (ptrdiff_t)y * job.src_stride);
        const uint32_t row_m = (uint32_t)__mul24(y + 3, job.src_stride);       // y >= -3
        uint16_t v[8];
        if (aligned && c0 >= 0 && c0 + 8 <= w) {
            if (BD > 8) {
                const uint4 q = gld_at<uint4>(src_m, row_m + (c0 + 8) * (int)sizeof(px_t));
                *(uint4 *)&tile[r][k * 8] = q;
                continue;
            } else {
                const uint2 q = gld_at<uint2>(src_m, row_m + (c0 + 8) * (int)sizeof(px_t));
                const uint32_t d[2] = { q.x, q.y };
#pragma unroll
                for (int j = 0; j < 8; j++)
                    v[j] = (d[j >> 2] >> ((j & 3) * 8)) & 0xff;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++)
                v[j] = gld<px_t>(row + clip3(c0 + j, x_min, x_max));
        }
        uint4 q;
        q.x = v[0] | (v[1] << 16); q.y = v[2] | (v[3] << 16); q.z = v[4] | (v[5] << 16); q.w = v[6] | (v[7] << 16);
        *(uint4 *)&tile[r][k * 8] = q;
    }
}

// ---------------------------------------------------------------------------------------------- classification

// vvc_filter_template.c:270 — direction sums {V,H,D0,D1} -> class 0..24 and transpose 0..3
template <int BD>
__device__ __forceinline__ void block_class(const int sum[4], int ac, int &cls, int &tr)
{
    const int v = sum[0], h = sum[1], d0 = sum[2], d1 = sum[3];
    const int dir_hv = v <= h, dir_d = d0 <= d1;
    const int hv_hi = max(v, h), hv_lo = min(v, h), d_hi = max(d0, d1), d_lo = min(d0, d1);
    const int main_hv = (uint64_t)(uint32_t)d_hi * (uint32_t)hv_lo <= (uint64_t)(uint32_t)hv_hi * (uint32_t)d_lo;
    const int hi = main_hv ? hv_hi : d_hi, lo = main_hv ? hv_lo : d_lo;
    cls = kAlfVarTab[clip3(((h + v) * ac) >> (BD - 1), 0, 15)];
    if (hi * 2 > 9 * lo)
        cls += ((main_hv << 1) + 2) * 5;
    else if (hi > 2 * lo)
        cls += ((main_hv << 1) + 1) * 5;
    tr = dir_d * 2 + dir_hv;
}

// register window of one 4x4 block: rows y-3 .. y+6, columns x-4 .. x+7, two samples per dword
struct Win {
    uint32_t d[10][6];
    __device__ __forceinline__ int at(int r, int c) const { return (d[r][c >> 1] >> ((c & 1) * 16)) & 0xffff; }
};

// Classification of the block at CTB row yb from its window (vvc_filter_template.c:299-381 restricted to one block).
template <int BD>
__device__ __forceinline__ void classify_win(const Win &w, int yb, int vb_pos, int &cls, int &tr)
{
    int first = 0, last = 4, ac = 2;
    if (yb + 4 == vb_pos) { last = 3; ac = 3; }
    else if (yb == vb_pos) { first = 1; ac = 3; }
    int sum[4] = { 0, 0, 0, 0 };
#pragma unroll
    for (int i = 0; i < 4; i++) {
        // gradient row i covers loop row yy = yb + 2i of the reference: sample A on window row 1+2i, B on 2+2i
        const int yy = yb + 2 * i;
        const bool fold_dn = yy == vb_pos;        // B's lower neighbour row replaced by B's row
        const bool fold_up = yy == vb_pos + 2;    // A's upper neighbour row replaced by A's row
        const int ra = 1 + 2 * i, rb = ra + 1;
        unsigned g[4] = { 0, 0, 0, 0 };
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int ca = 2 + 2 * j, cb = ca + 1;
            const int a2 = w.at(ra, ca) << 1, b2 = w.at(rb, cb) << 1;
#define UPA(c) (fold_up ? w.at(ra, c) : w.at(ra - 1, c))
#define DNB(c) (fold_dn ? w.at(rb, c) : w.at(rb + 1, c))
            // |2p - n1 - n2| accumulated with one v_sad_u32 per term (all operands are non-negative)
            g[0] = __sad(a2, UPA(ca) + w.at(rb, ca), __sad(b2, w.at(ra, cb) + DNB(cb), g[0]));
            g[1] = __sad(a2, w.at(ra, ca - 1) + w.at(ra, ca + 1), __sad(b2, w.at(rb, cb - 1) + w.at(rb, cb + 1), g[1]));
            g[2] = __sad(a2, UPA(ca - 1) + w.at(rb, ca + 1), __sad(b2, w.at(ra, cb - 1) + DNB(cb + 1), g[2]));
            g[3] = __sad(a2, UPA(ca + 1) + w.at(rb, ca - 1), __sad(b2, w.at(ra, cb + 1) + DNB(cb - 1), g[3]));
#undef UPA
#undef DNB
        }
        if (i >= first && i < last) {
            sum[0] += g[0]; sum[1] += g[1]; sum[2] += g[2]; sum[3] += g[3];
        }
    }
    block_class<BD>(sum, ac, cls, tr);
}

// 7x7 diamond on one 4x4 block entirely from the register window.  KIND 0: no virtual boundary nearby; KIND 1: the block's
// last row is adjacent to the boundary from above (row i is 3 - i rows away); KIND 2: its first row is adjacent from below
// (row i is i rows away).  Tap rows fold to min(dy, dist) and the adjacent row uses the >> 10 rounding (:76-96,:115-118);
// everything is a compile-time index, so the window stays in registers.
template <int BD, int KIND>
__device__ __forceinline__ void filter_block_regs(const Win &win, const int (&f)[12], const int (&c)[12], uint8_t *drow, int dst_stride)
{
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int dist = KIND == 0 ? 3 : KIND == 1 ? 3 - i : i;
        int out[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int r0 = 3 + i, c0 = 4 + j;
            const int cur = win.at(r0, c0);
            int sum = 0;
#pragma unroll
            for (int k = 0; k < 12; k++) {
                const int dy = kLumaTap[k][0] < dist ? kLumaTap[k][0] : dist, dx = kLumaTap[k][1];
                const int a = win.at(r0 + dy, c0 + dx), b = win.at(r0 - dy, c0 - dx);
                sum = mad24(f[k], clamp_sym(a - cur, c[k]) + clamp_sym(b - cur, c[k]), sum);
            }
            out[j] = clip_px<BD>((dist == 0 ? (sum + 512) >> 10 : (sum + 64) >> 7) + cur);
        }
        uint8_t *d = drow + (ptrdiff_t)i * dst_stride;
        if (BD > 8)
            gst<uint2>(d, make_uint2(out[0] | (out[1] << 16), out[2] | (out[3] << 16)));
        else
            gst<uint32_t>(d, out[0] | (out[1] << 8) | (out[2] << 16) | (out[3] << 24));
    }
}

// ---------------------------------------------------------------------------------------------- luma kernel

// MODE 0: per-4x4 coefficient/clip arrays (the reference's alf.filter[LUMA] slot).
// MODE 1: fused classify -> coefficient gather -> filter (the three slots the caller chains, vvc_filter.c:1139-1186).
// MODE 2: classify only (alf.classify slot): writes class_idx / transpose_idx ints.
template <int BD, int MODE>
__global__ __launch_bounds__(256) void alf_luma_kernel(const vvc355_alf_job *__restrict__ jobs)
{
    __shared__ __attribute__((aligned(16))) uint16_t tile[kTileH][kTileW];
    // fused mode: the CTB's filter set, expanded for the 4 transposes: [transpose][class][tap] = coeff | clip << 16
    __shared__ __attribute__((aligned(16))) uint32_t ftab[MODE == 1 ? 4 * 25 * 12 : 4];
    const int wg = xcd_chunked(blockIdx.x, gridDim.x);       // an XCD's L2 sees a contiguous run of CTBs (shared aprons)
    const vvc355_alf_job job = jobs[wg >> 2];      // (scalar load_uniform measured 6 % slower here: this kernel is VALU-bound and register-tight)
    const int y_base = (wg & 3) * kStripH;
    if (y_base >= job.h)
        return;
    const int rows = min(kStripH, job.h - y_base);
    stage_tile<BD>(tile, job, y_base, rows, 3);
    if (MODE == 1) {
        // alf_recon_coeff_and_clip (:383) for every (transpose, class) once per workgroup
        const int16_t *coeff_set = (const int16_t *)job.coeff;
        const uint8_t *clip_idx = (const uint8_t *)job.clip, *c2f = (const uint8_t *)job.class_to_filt;
        for (int e = threadIdx.x; e < 4 * 25 * 12; e += blockDim.x) {
            const int t = e / 300, r = e - t * 300, cls = r / 12, k = r - cls * 12;
            const int idx = kAlfPerm[t][k];
            const int q = gld<uint8_t>(clip_idx + cls * 12 + idx);
            const int cv = 1 << (BD - (q == 0 ? 0 : 2 * q + 1));       // {2^bd, 2^(bd-3), 2^(bd-5), 2^(bd-7)}
            ftab[e] = (uint32_t)(uint16_t)gld<int16_t>(coeff_set + gld<uint8_t>(c2f + cls) * 12 + idx) | ((uint32_t)cv << 16);
        }
    }
    __syncthreads();

    const int bx = threadIdx.x & 31, by = threadIdx.x >> 5;
    const int x = bx * 4, yl = by * 4, yb = y_base + yl;
    if (x >= job.w || yl >= rows)
        return;

    Win win;
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint2 *p = (const uint2 *)&tile[yl + r][x + kColOff - 4];
        const uint2 a = p[0], b = p[1], c = p[2];
        win.d[r][0] = a.x; win.d[r][1] = a.y; win.d[r][2] = b.x; win.d[r][3] = b.y; win.d[r][4] = c.x; win.d[r][5] = c.y;
    }

    const int vb_pos = job.vb_pos;
    const int blk = (yb >> 2) * (job.w >> 2) + bx;
    int f[12], c[12];
    if (MODE == 0) {
        const int16_t *fp = (const int16_t *)job.coeff + blk * 12, *cp = (const int16_t *)job.clip + blk * 12;
#pragma unroll
        for (int k = 0; k < 12; k++) { f[k] = fp[k]; c[k] = cp[k]; }
    } else {
        int cls, tr;
        classify_win<BD>(win, yb, vb_pos, cls, tr);
        if (MODE == 2) {
            ((int *)job.coeff)[blk] = cls;
            ((int *)job.clip)[blk] = tr;
            return;
        }
        const uint4 *e = (const uint4 *)&ftab[(tr * 25 + cls) * 12];
        const uint4 e0 = e[0], e1 = e[1], e2 = e[2];
        const uint32_t ev[12] = { e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w, e2.x, e2.y, e2.z, e2.w };
#pragma unroll
        for (int k = 0; k < 12; k++) {
            f[k] = (int)(int16_t)(ev[k] & 0xffff);
            c[k] = (int)(ev[k] >> 16);
        }
    }

    uint8_t *dst = (uint8_t *)job.dst;
    // rows of this block closer than 3 to the virtual boundary fold their taps: generic LDS path
    const bool near_vb = vb_dist(yb, vb_pos) < 3 || vb_dist(yb + 3, vb_pos) < 3 || (yb < vb_pos && yb + 3 >= vb_pos);
    uint8_t *drow = dst + (ptrdiff_t)yb * job.dst_stride + x * (BD > 8 ? 2 : 1);
    if (!near_vb) {
        filter_block_regs<BD, 0>(win, f, c, drow, job.dst_stride);
    } else if (yb + 4 == vb_pos) {
        filter_block_regs<BD, 1>(win, f, c, drow, job.dst_stride);      // the block row just above the boundary
    } else if (yb == vb_pos) {
        filter_block_regs<BD, 2>(win, f, c, drow, job.dst_stride);      // the block row just below it
    } else {
        for (int i = 0; i < 4; i++) {
            const int y = yb + i, dist = vb_dist(y, vb_pos);
            const int tr0 = yl + i + 3;
            for (int j = 0; j < 4; j++) {
                const int tc0 = x + j + kColOff;
                const int cur = tile[tr0][tc0];
                int sum = 0;
                for (int k = 0; k < 12; k++) {
                    const int dy = min((int)kLumaTap[k][0], dist), dx = kLumaTap[k][1];
                    const int a = tile[tr0 + dy][tc0 + dx], b = tile[tr0 - dy][tc0 - dx];
                    sum += f[k] * (clip3(a - cur, -c[k], c[k]) + clip3(b - cur, -c[k], c[k]));
                }
                sum = dist == 0 ? (sum + 512) >> 10 : (sum + 64) >> 7;
                st_px<BD>(dst + (ptrdiff_t)y * job.dst_stride, x + j, clip_px<BD>(sum + cur));
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- chroma kernel

// alf.filter[CHROMA] (:137): one 6-tap set per rectangle; rectangle up to 128x128 (4:4:4), usually 64x64.
template <int BD>
__global__ __launch_bounds__(256) void alf_chroma_kernel(const vvc355_alf_job *__restrict__ jobs, int log2_strips)
{
    // 1 << log2_strips workgroups per job, kStripH rows each (the launcher knows the tallest rectangle, or assumes 128 rows)
    __shared__ __attribute__((aligned(16))) uint16_t tile[kTileH][kTileW];
    const int wg = xcd_chunked(blockIdx.x, gridDim.x);
    const vvc355_alf_job job = load_uniform(jobs + (wg >> log2_strips));
    const int y_base = (wg & ((1 << log2_strips) - 1)) * kStripH;
    if (y_base >= job.h)
        return;
    const int rows = min(kStripH, job.h - y_base);
    stage_tile<BD>(tile, job, y_base, rows, 2);
    __syncthreads();

    int f[6], c[6];
#pragma unroll
    for (int k = 0; k < 6; k++) { f[k] = ((const int16_t *)job.coeff)[k]; c[k] = ((const int16_t *)job.clip)[k]; }
    uint8_t *dst = (uint8_t *)job.dst;
    const int vb_pos = job.vb_pos;
    // lane -> 4 consecutive samples of one row
    const int lpr = job.w >> 2;
    for (int i = threadIdx.x; i < rows * lpr; i += blockDim.x) {
        const int yl = i / lpr, x = (i - yl * lpr) * 4;
        const int y = y_base + yl, dist = vb_dist(y, vb_pos), tr0 = yl + 3;
        // 5 rows x 8 columns (x - 2 .. x + 5) of the tile in registers: rows 0 / +-1 / +-2 folded onto the virtual boundary
        const int d1 = min(1, dist), d2 = min(2, dist);
        uint32_t win[5][4];
        {
            const int rr[5] = { tr0 - d2, tr0 - d1, tr0, tr0 + d1, tr0 + d2 };
#pragma unroll
            for (int r = 0; r < 5; r++) {
                const uint32_t *p = (const uint32_t *)&tile[rr[r]][x + kColOff - 2];       // even index: 4-byte aligned
#pragma unroll
                for (int m = 0; m < 4; m++) win[r][m] = p[m];
            }
        }
#define WPX(r, c) ((int)((win[r][(c) >> 1] >> (((c) & 1) * 16)) & 0xffff))
        int out[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int cur = WPX(2, j + 2);
            int sum = 0;
            sum = mad24(f[0], clamp_sym(WPX(4, j + 2) - cur, c[0]) + clamp_sym(WPX(0, j + 2) - cur, c[0]), sum);     // (2, 0)
            sum = mad24(f[1], clamp_sym(WPX(3, j + 3) - cur, c[1]) + clamp_sym(WPX(1, j + 1) - cur, c[1]), sum);     // (1, 1)
            sum = mad24(f[2], clamp_sym(WPX(3, j + 2) - cur, c[2]) + clamp_sym(WPX(1, j + 2) - cur, c[2]), sum);     // (1, 0)
            sum = mad24(f[3], clamp_sym(WPX(3, j + 1) - cur, c[3]) + clamp_sym(WPX(1, j + 3) - cur, c[3]), sum);     // (1, -1)
            sum = mad24(f[4], clamp_sym(WPX(2, j + 4) - cur, c[4]) + clamp_sym(WPX(2, j) - cur, c[4]), sum);         // (0, 2)
            sum = mad24(f[5], clamp_sym(WPX(2, j + 3) - cur, c[5]) + clamp_sym(WPX(2, j + 1) - cur, c[5]), sum);     // (0, 1)
            sum = dist == 0 ? (sum + 512) >> 10 : (sum + 64) >> 7;
            out[j] = clip_px<BD>(sum + cur);
        }
#undef WPX
        uint8_t *d = dst + (ptrdiff_t)y * job.dst_stride;
        if (BD > 8)
            gst<uint2>(d + x * 2, make_uint2(out[0] | (out[1] << 16), out[2] | (out[3] << 16)));
        else
            gst<uint32_t>(d + x, out[0] | (out[1] << 8) | (out[2] << 16) | (out[3] << 24));
    }
}

// ---------------------------------------------------------------------------------------------- CC-ALF kernel

// alf.filter_cc (:223): dst = chroma rectangle (w x h), src = co-located luma; job.coeff = int16[7].  The luma neighbourhood
// (one column left / right, one row above, two below the co-located rectangle) is read in place on the sides whose ext_* is
// non-zero; on a side with ext_* == 0 the rectangle's own border samples stand in, which is what the reference's padded luma
// copy holds there (alf_prepare_buffer, vvc_filter.c:1105-1137).
template <int BD>
__global__ __launch_bounds__(256) void alf_cc_kernel(const vvc355_alf_job *__restrict__ jobs)
{
    const int lin = xcd_chunked(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int by = lin / (int)gridDim.x, bx = lin - by * (int)gridDim.x;
    const vvc355_alf_job job = load_uniform(jobs + by);
    const int hs = job.hs, vs = job.vs, vb_pos = job.vb_pos;
    const uint8_t *luma = (const uint8_t *)job.src;
    const ptrdiff_t ls = job.src_stride / (ptrdiff_t)sizeof(typename Px<BD>::type);
    int f[7];
#pragma unroll
    for (int k = 0; k < 7; k++) f[k] = ((const int16_t *)job.coeff)[k];
    const int row_min = job.ext_t ? -1 : 0, row_max = (job.h << vs) - 1 + (job.ext_b ? 2 : 0);
    if (hs == 1 && (job.w & 3) == 0) {
        // 4:2:0 / 4:2:2: a lane corrects 4 consecutive chroma samples; the 8 co-located luma samples of each of the four rows
        // involved come as one vector (the taps at 2x - 1 / 2x + 1 are the neighbouring halves of its registers), plus the one
        // sample left of the vector for the two rows that have side taps.  All loads go out before the first use.
        using px_t = typename Px<BD>::type;
        const int wq = job.w >> 2;
        for (int i = bx * blockDim.x + threadIdx.x; i < wq * job.h; i += gridDim.x * blockDim.x) {
            const int y = i / wq, x = (i - y * wq) * 4;
            const int ly = y << vs;
            if (!vs && (ly == vb_pos || ly == vb_pos + 1))
                continue;
            int up = -1, dn = 1, dn2 = 2;
            if (ly == vb_pos - 2 || ly == vb_pos + 1) dn2 = 1;
            else if (ly == vb_pos - 1 || ly == vb_pos) up = dn = dn2 = 0;
            const px_t *l0 = (const px_t *)luma + (ptrdiff_t)ly * ls + 2 * x;
            up = max(ly + up, row_min) - ly; dn = min(ly + dn, row_max) - ly; dn2 = min(ly + dn2, row_max) - ly;
            const int left = (x | job.ext_l) ? 1 : 0;        // column -1 of the rectangle only where it is readable
            uint32_t ru[4], rc[4], rd[4], r2[4];
            load8_u16<BD>(l0 + up * ls, ru);
            load8_u16<BD>(l0, rc);
            load8_u16<BD>(l0 + dn * ls, rd);
            load8_u16<BD>(l0 + dn2 * ls, r2);
            const int ec = gld<px_t>(l0 - left), ed = gld<px_t>(l0 + dn * ls - left);
            uint8_t *d = (uint8_t *)job.dst + (ptrdiff_t)y * job.dst_stride + x * (int)sizeof(px_t);
            uint32_t cv[2];
            if (BD > 8) { const uint2 q = gld<uint2>(d); cv[0] = q.x; cv[1] = q.y; }
            else { const uint32_t q = gld<uint32_t>(d); cv[0] = __builtin_amdgcn_perm(0, q, 0x0c010c00u); cv[1] = __builtin_amdgcn_perm(0, q, 0x0c030c02u); }
            int out[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int c = rc[j] & 0xffff;
                const int lc = j ? (int)(rc[j - 1] >> 16) : ec, ld = j ? (int)(rd[j - 1] >> 16) : ed;
                int sum = 0;
                sum += f[0] * ((int)(ru[j] & 0xffff) - c);
                sum += f[1] * (lc - c);
                sum += f[2] * ((int)(rc[j] >> 16) - c);
                sum += f[3] * (ld - c);
                sum += f[4] * ((int)(rd[j] & 0xffff) - c);
                sum += f[5] * ((int)(rd[j] >> 16) - c);
                sum += f[6] * ((int)(r2[j] & 0xffff) - c);
                sum = clip3((sum + 64) >> 7, -(1 << (BD - 1)), (1 << (BD - 1)) - 1);
                out[j] = clip_px<BD>(sum + (int)((cv[j >> 1] >> ((j & 1) * 16)) & 0xffff));
            }
            if (BD > 8)
                gst<uint2>(d, make_uint2(out[0] | (out[1] << 16), out[2] | (out[3] << 16)));
            else
                gst<uint32_t>(d, out[0] | (out[1] << 8) | (out[2] << 16) | (out[3] << 24));
        }
        return;
    }
    for (int i = bx * blockDim.x + threadIdx.x; i < job.w * job.h; i += gridDim.x * blockDim.x) {
        const int y = i / job.w, x = i - y * job.w;
        const int ly = y << vs;
        if (!vs && (ly == vb_pos || ly == vb_pos + 1))
            continue;
        int up = -1, dn = 1, dn2 = 2;
        if (ly == vb_pos - 2 || ly == vb_pos + 1) dn2 = 1;
        else if (ly == vb_pos - 1 || ly == vb_pos) up = dn = dn2 = 0;
        up = max(ly + up, row_min) - ly; dn = min(ly + dn, row_max) - ly; dn2 = min(ly + dn2, row_max) - ly;
        const int lx = x << hs;
        const int xl = max(lx - 1, job.ext_l ? -1 : 0) - lx, xr = min(lx + 1, (job.w << hs) - 1 + (job.ext_r ? 1 : 0)) - lx;
        const ptrdiff_t o = (ptrdiff_t)ly * ls + lx;
        const int c = ld_px<BD>(luma, o);
        int sum = 0;
        sum += f[0] * (ld_px<BD>(luma, o + up * ls) - c);
        sum += f[1] * (ld_px<BD>(luma, o + xl) - c);
        sum += f[2] * (ld_px<BD>(luma, o + xr) - c);
        sum += f[3] * (ld_px<BD>(luma, o + dn * ls + xl) - c);
        sum += f[4] * (ld_px<BD>(luma, o + dn * ls) - c);
        sum += f[5] * (ld_px<BD>(luma, o + dn * ls + xr) - c);
        sum += f[6] * (ld_px<BD>(luma, o + dn2 * ls) - c);
        sum = clip3((sum + 64) >> 7, -(1 << (BD - 1)), (1 << (BD - 1)) - 1);
        uint8_t *d = (uint8_t *)job.dst + (ptrdiff_t)y * job.dst_stride;
        st_px<BD>(d, x, clip_px<BD>(sum + ld_px<BD>(d, x)));
    }
}

// alf.recon_coeff_and_clip (:383) as its own slot
template <int BD>
__global__ void alf_recon_kernel(int16_t *coeff, int16_t *clip, const int *class_idx, const int *transpose_idx, int size,
                                 const int16_t *coeff_set, const uint8_t *clip_idx_set, const uint8_t *class_to_filt)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= size * 12)
        return;
    const int b = i / 12, j = i - b * 12;
    const int cls = class_idx[b], idx = kAlfPerm[transpose_idx[b]][j];
    const int q = clip_idx_set[cls * 12 + idx];
    coeff[i] = coeff_set[class_to_filt[cls] * 12 + idx];
    clip[i] = (int16_t)(1 << (BD - (q == 0 ? 0 : 2 * q + 1)));
}

// ---------------------------------------------------------------------------------------------- stage driver

// Zero filter set: a CTB component with alf_ctb_flag off passes through (sum = 0 -> dst = src); also the harmless operand of
// CC-ALF jobs that are switched off (w = h = 0).
__device__ static const int16_t kAlfZeroSet[25 * 12] = { 0 };

// ff_vvc_alf_filter (vvc_filter.c:1254-1318) per CTB as a descriptor builder: one lane per CTB writes its luma job, two chroma
// jobs and two CC-ALF jobs.  edges[] (:1264-1278) become ext_* = 0 (replicate) / 3 (read the neighbour in place).
template <int BD>
__global__ void alf_build_kernel(const vvc355_alf_frame *__restrict__ frame, int n_ctbs, vvc355_alf_job *luma, vvc355_alf_job *chroma,
                                 vvc355_alf_job *cc, int16_t *clips)
{
    const vvc355_alf_frame F = load_uniform(frame);       // scalar loads, once: the fields are read dozens of times
    const vvc355_alf_frame *fp = &F;
    const int rs = blockIdx.x * blockDim.x + threadIdx.x;
    if (rs >= n_ctbs)
        return;
    const int px = BD > 8 ? 2 : 1;
    const int cw = fp->ctb_width, chh = fp->ctb_height, yc = rs / cw, xc = rs - yc * cw;
    const int ctb_size = 1 << fp->ctb_log2;
    const vvc355_alf_ctb alf = ((const vvc355_alf_ctb *)fp->alf)[rs];
    const int16_t *slice = (const int16_t *)fp->slice_idx;
    const int16_t *col_bd = (const int16_t *)fp->ctb_to_col_bd, *row_bd = (const int16_t *)fp->ctb_to_row_bd;
    const int me = slice[rs];
    const vvc355_alf_slice *sl = (const vvc355_alf_slice *)fp->slices + me;
    bool e_l = xc == 0, e_t = yc == 0, e_r = xc == cw - 1, e_b = yc == chh - 1;
    if (!fp->lfate) {
        e_l = e_l || col_bd[xc] == xc;
        e_t = e_t || row_bd[yc] == yc;
        e_r = e_r || col_bd[xc] != col_bd[xc + 1];
        e_b = e_b || row_bd[yc] != row_bd[yc + 1];
    }
    if (!fp->lfase) {
        e_l = e_l || slice[rs - 1] != me;                  // e_* already set where the neighbour does not exist
        e_t = e_t || slice[rs - cw] != me;
        e_r = e_r || slice[rs + 1] != me;
        e_b = e_b || slice[rs + cw] != me;
    }
    vvc355_alf_job j = {};
    j.ext_l = e_l ? 0 : 3; j.ext_t = e_t ? 0 : 3; j.ext_r = e_r ? 0 : 3; j.ext_b = e_b ? 0 : 3;
    // ---- luma: alf_filter_luma (:1171) with alf_get_coeff_and_clip (:1142)
    {
        const int x0 = xc * ctb_size, y0 = yc * ctb_size;
        j.dst = fp->dst[0] + (uint64_t)y0 * fp->dst_stride[0] + x0 * px;
        j.src = fp->src[0] + (uint64_t)y0 * fp->src_stride[0] + x0 * px;
        j.dst_stride = fp->dst_stride[0]; j.src_stride = fp->src_stride[0];
        j.w = (int16_t)min(fp->width - x0, ctb_size); j.h = (int16_t)min(fp->height - y0, ctb_size);
        j.vb_pos = (int16_t)(ctb_size - 4);
        if (!alf.ctb_flag[0]) {
            j.coeff = (uint64_t)kAlfZeroSet; j.clip = (uint64_t)kAlfZeroSet; j.class_to_filt = (uint64_t)kAlfZeroSet;
        } else if (alf.filt_set_idx_y < 16) {
            j.coeff = (uint64_t)t_alf_fix_filt_coeff; j.clip = (uint64_t)kAlfZeroSet;
            j.class_to_filt = (uint64_t)(t_alf_class_to_filt_map + alf.filt_set_idx_y * 25);
        } else {
            j.coeff = sl->luma_coeff[alf.filt_set_idx_y - 16]; j.clip = sl->luma_clip_idx[alf.filt_set_idx_y - 16];
            j.class_to_filt = (uint64_t)t_alf_aps_class_to_filt_map;
        }
        luma[rs] = j;
    }
    if (fp->n_comp < 3)
        return;
    const int hs = fp->hs, vs = fp->vs;
    const int x0 = (xc * ctb_size) >> hs, y0 = (yc * ctb_size) >> vs;
    const int w = min((fp->width >> hs) - x0, ctb_size >> hs), h = min((fp->height >> vs) - y0, ctb_size >> vs);
    for (int c = 1; c < 3; c++) {
        // ---- chroma: alf_filter_chroma (:1195)
        j.dst = fp->dst[c] + (uint64_t)y0 * fp->dst_stride[c] + x0 * px;
        j.src = fp->src[c] + (uint64_t)y0 * fp->src_stride[c] + x0 * px;
        j.dst_stride = fp->dst_stride[c]; j.src_stride = fp->src_stride[c];
        j.w = (int16_t)w; j.h = (int16_t)h; j.vb_pos = (int16_t)((ctb_size >> vs) - 2);
        j.class_to_filt = 0; j.hs = j.vs = 0;
        int16_t *cl = clips + (2 * rs + c - 1) * 8;
        if (alf.ctb_flag[c]) {
            const int idx = alf.alt_idx[c - 1];
            const uint8_t *ci = (const uint8_t *)sl->chroma_clip_idx + idx * 6;
            for (int k = 0; k < 6; k++) {
                const int q = ci[k];
                cl[k] = (int16_t)(1 << (BD - (q == 0 ? 0 : 2 * q + 1)));      // alf_clip_from_idx (:1188)
            }
            j.coeff = sl->chroma_coeff + idx * 12;
        } else {
            for (int k = 0; k < 6; k++) cl[k] = 0;
            j.coeff = (uint64_t)kAlfZeroSet;
        }
        j.clip = (uint64_t)cl;
        chroma[2 * rs + c - 1] = j;
        // ---- CC-ALF: alf_filter_cc (:1212) on the co-located luma
        const uint64_t cc_set = sl->cc_coeff[c - 1];
        const bool on = alf.cc_idc[c - 1] && cc_set;
        j.src = fp->src[0] + (uint64_t)(y0 << vs) * fp->src_stride[0] + (x0 << hs) * px;
        j.src_stride = fp->src_stride[0];
        j.coeff = on ? cc_set + (alf.cc_idc[c - 1] - 1) * 14 : (uint64_t)kAlfZeroSet;
        j.clip = 0;
        j.w = (int16_t)(on ? w : 0); j.h = (int16_t)(on ? h : 0);
        j.vb_pos = (int16_t)(ctb_size - 4);
        j.hs = (int8_t)hs; j.vs = (int8_t)vs;
        cc[2 * rs + c - 1] = j;
    }
}

// ---------------------------------------------------------------------------------------------- launchers

static void launch_luma(int bd, int mode, const vvc355_alf_job *jobs, int n, hipStream_t st)
{
    if (n <= 0) return;
    const dim3 grid(n * 4), block(256);
    VVC355_BD_DISPATCH(bd, {
        if (mode == 0)      hipLaunchKernelGGL((alf_luma_kernel<BD, 0>), grid, block, 0, st, jobs);
        else if (mode == 1) hipLaunchKernelGGL((alf_luma_kernel<BD, 1>), grid, block, 0, st, jobs);
        else                hipLaunchKernelGGL((alf_luma_kernel<BD, 2>), grid, block, 0, st, jobs);
    });
    HIP_CHECK(hipGetLastError());
}

static void launch_chroma(int bd, const vvc355_alf_job *jobs, int n, hipStream_t st, int max_h = 128)
{
    if (n <= 0) return;
    const int log2_strips = max_h <= kStripH ? 0 : max_h <= 2 * kStripH ? 1 : 2;
    VVC355_BD_DISPATCH(bd, hipLaunchKernelGGL((alf_chroma_kernel<BD>), dim3(n << log2_strips), dim3(256), 0, st, jobs, log2_strips));
    HIP_CHECK(hipGetLastError());
}

static void launch_cc(int bd, const vvc355_alf_job *jobs, int n, hipStream_t st)
{
    if (n <= 0) return;
    VVC355_BD_DISPATCH(bd, hipLaunchKernelGGL((alf_cc_kernel<BD>), dim3(4, n), dim3(256), 0, st, jobs));
    HIP_CHECK(hipGetLastError());
}

static void check_luma_dims(int w, int h)
{
    if (w <= 0 || h <= 0 || w > 128 || h > 128 || (w & 3) || (h & 3)) {
        fprintf(stderr, "vvc_mi355: ALF rectangle %dx%d outside the slot's domain (multiples of 4, <= 128)\n", w, h);
        abort();
    }
}

} // namespace vvc355

using namespace vvc355;

// =============================================================================================== C ABI

extern "C" {

// ---- batched (device-resident) entries
void vvc355_alf_luma_batch(void *stream, int bd, int fused, const vvc355_alf_job *jobs_dev, int n_jobs)
{
    if (n_jobs <= 0) return;
    launch_luma(bd, fused ? 1 : 0, jobs_dev, n_jobs, (hipStream_t)stream);
}
void vvc355_alf_chroma_batch(void *stream, int bd, const vvc355_alf_job *jobs_dev, int n_jobs)
{
    if (n_jobs <= 0) return;
    launch_chroma(bd, jobs_dev, n_jobs, (hipStream_t)stream);
}
void vvc355_alf_cc_batch(void *stream, int bd, const vvc355_alf_job *jobs_dev, int n_jobs)
{
    if (n_jobs <= 0) return;
    launch_cc(bd, jobs_dev, n_jobs, (hipStream_t)stream);
}

// ---- ALF stage driver
size_t vvc355_alf_frame_work_bytes(int n_ctbs)
{
    return (size_t)n_ctbs * (5 * sizeof(vvc355_alf_job) + 2 * 8 * sizeof(int16_t));
}

void vvc355_alf_frame_pass(void *stream, int bd, const vvc355_alf_frame *frame_dev, const vvc355_alf_frame *frame_host, void *work_dev)
{
    const int n = frame_host->ctb_width * frame_host->ctb_height;
    if (n <= 0) return;
    if (frame_host->ctb_log2 < 5 || frame_host->ctb_log2 > 7 || (frame_host->width & 7) || (frame_host->height & 7)) {
        fprintf(stderr, "vvc_mi355: ALF frame %dx%d (CTB log2 %d) outside the driver's domain\n", frame_host->width, frame_host->height, frame_host->ctb_log2);
        abort();
    }
    vvc355_alf_job *luma = (vvc355_alf_job *)work_dev, *chroma = luma + n, *cc = chroma + 2 * n;
    int16_t *clips = (int16_t *)(cc + 2 * n);
    hipStream_t st = (hipStream_t)stream;
    VVC355_BD_DISPATCH(bd, hipLaunchKernelGGL((alf_build_kernel<BD>), dim3((n + 63) / 64), dim3(64), 0, st, frame_dev, n, luma, chroma, cc, clips));
    HIP_CHECK(hipGetLastError());
    launch_luma(bd, 1, luma, n, st);
    if (frame_host->n_comp >= 3) {
        launch_chroma(bd, chroma, 2 * n, st, (1 << frame_host->ctb_log2) >> frame_host->vs);
        launch_cc(bd, cc, 2 * n, st);
    }
}

// ---- synchronous per-slot entries (host pointers; reference signatures + leading bd)
void vvc355_alf_filter_luma(int bd, uint8_t *dst, ptrdiff_t dst_stride, const uint8_t *src, ptrdiff_t src_stride,
                            int width, int height, const int16_t *filter, const int16_t *clip, int vb_pos)
{
    check_luma_dims(width, height);
    const int px = bd > 8 ? 2 : 1, nblk = (width >> 2) * (height >> 2);
    SlotCall call;
    const Staged s = call.rect(src, src_stride, -3 * px, (width + 3) * px, -3, height + 3, true, false);
    const Staged d = call.rect(dst, dst_stride, 0, width * px, 0, height, false, true);
    vvc355_alf_job job = {};
    job.dst = (uint64_t)d.dev; job.src = (uint64_t)s.dev;
    job.dst_stride = (int32_t)d.pitch; job.src_stride = (int32_t)s.pitch;
    job.coeff = (uint64_t)call.linear(filter, (size_t)nblk * 24, true, false);
    job.clip = (uint64_t)call.linear(clip, (size_t)nblk * 24, true, false);
    job.w = (int16_t)width; job.h = (int16_t)height; job.vb_pos = (int16_t)vb_pos;
    job.ext_l = job.ext_r = job.ext_t = job.ext_b = 3;
    launch_luma(bd, 0, call.upload(&job, 1), 1, call.stream());
}

void vvc355_alf_filter_chroma(int bd, uint8_t *dst, ptrdiff_t dst_stride, const uint8_t *src, ptrdiff_t src_stride,
                              int width, int height, const int16_t *filter, const int16_t *clip, int vb_pos)
{
    check_luma_dims(width, height);
    const int px = bd > 8 ? 2 : 1;
    SlotCall call;
    const Staged s = call.rect(src, src_stride, -2 * px, (width + 2) * px, -2, height + 2, true, false);
    const Staged d = call.rect(dst, dst_stride, 0, width * px, 0, height, false, true);
    vvc355_alf_job job = {};
    job.dst = (uint64_t)d.dev; job.src = (uint64_t)s.dev;
    job.dst_stride = (int32_t)d.pitch; job.src_stride = (int32_t)s.pitch;
    job.coeff = (uint64_t)call.linear(filter, 12, true, false);
    job.clip = (uint64_t)call.linear(clip, 12, true, false);
    job.w = (int16_t)width; job.h = (int16_t)height; job.vb_pos = (int16_t)vb_pos;
    job.ext_l = job.ext_r = job.ext_t = job.ext_b = 2;
    launch_chroma(bd, call.upload(&job, 1), 1, call.stream());
}

void vvc355_alf_filter_cc(int bd, uint8_t *dst, ptrdiff_t dst_stride, const uint8_t *luma, ptrdiff_t luma_stride,
                          int width, int height, int hs, int vs, const int16_t *filter, int vb_pos)
{
    if (width <= 0 || height <= 0)
        return;
    const int px = bd > 8 ? 2 : 1;
    SlotCall call;
    const Staged s = call.rect(luma, luma_stride, -px, ((width << hs) + 1) * px, -1, (height << vs) + 2, true, false);
    const Staged d = call.rect(dst, dst_stride, 0, width * px, 0, height, true, true);
    vvc355_alf_job job = {};
    job.dst = (uint64_t)d.dev; job.src = (uint64_t)s.dev;
    job.dst_stride = (int32_t)d.pitch; job.src_stride = (int32_t)s.pitch;
    job.coeff = (uint64_t)call.linear(filter, 14, true, false);
    job.w = (int16_t)width; job.h = (int16_t)height; job.vb_pos = (int16_t)vb_pos;
    job.hs = (int8_t)hs; job.vs = (int8_t)vs;
    job.ext_l = job.ext_r = job.ext_t = job.ext_b = 3;     // the slot's caller hands over a padded buffer
    launch_cc(bd, call.upload(&job, 1), 1, call.stream());
}

void vvc355_alf_classify(int bd, int *class_idx, int *transpose_idx, const uint8_t *src, ptrdiff_t src_stride,
                         int width, int height, int vb_pos, int *gradient_tmp)
{
    (void)gradient_tmp;      // the reference's scratch plane; the kernel keeps gradients in registers
    check_luma_dims(width, height);
    const int px = bd > 8 ? 2 : 1, nblk = (width >> 2) * (height >> 2);
    SlotCall call;
    const Staged s = call.rect(src, src_stride, -3 * px, (width + 3) * px, -3, height + 3, true, false);
    vvc355_alf_job job = {};
    job.src = (uint64_t)s.dev; job.src_stride = (int32_t)s.pitch;
    job.coeff = (uint64_t)call.linear(class_idx, (size_t)nblk * 4, false, true);
    job.clip = (uint64_t)call.linear(transpose_idx, (size_t)nblk * 4, false, true);
    job.w = (int16_t)width; job.h = (int16_t)height; job.vb_pos = (int16_t)vb_pos;
    job.ext_l = job.ext_r = job.ext_t = job.ext_b = 3;
    launch_luma(bd, 2, call.upload(&job, 1), 1, call.stream());
}

void vvc355_alf_recon_coeff_and_clip(int bd, int16_t *coeff, int16_t *clip, const int *class_idx, const int *transpose_idx,
                                     int size, const int16_t *coeff_set, const uint8_t *clip_idx_set,
                                     const uint8_t *class_to_filt)
{
    if (size <= 0)
        return;
    // the slot does not say how many filters coeff_set holds; the class map (host memory) bounds what is read
    int n_filters = 0;
    for (int i = 0; i < 25; i++)
        n_filters = class_to_filt[i] + 1 > n_filters ? class_to_filt[i] + 1 : n_filters;
    SlotCall call;
    int16_t *d_coeff = (int16_t *)call.linear(coeff, (size_t)size * 24, false, true);
    int16_t *d_clip = (int16_t *)call.linear(clip, (size_t)size * 24, false, true);
    const int *d_cls = (const int *)call.linear(class_idx, (size_t)size * 4, true, false);
    const int *d_tr = (const int *)call.linear(transpose_idx, (size_t)size * 4, true, false);
    const int16_t *d_set = (const int16_t *)call.linear(coeff_set, (size_t)n_filters * 24, true, false);
    const uint8_t *d_ci = (const uint8_t *)call.linear(clip_idx_set, 25 * 12, true, false);
    const uint8_t *d_map = (const uint8_t *)call.linear(class_to_filt, 25, true, false);
    const int n = size * 12;
    VVC355_BD_DISPATCH(bd, hipLaunchKernelGGL((alf_recon_kernel<BD>), dim3((n + 255) / 256), dim3(256), 0, call.stream(),
                                              d_coeff, d_clip, d_cls, d_tr, size, d_set, d_ci, d_map));
    HIP_CHECK(hipGetLastError());
}

} // extern "C"
