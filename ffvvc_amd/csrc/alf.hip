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

// alf_recon_coeff_and_clip's transpose index lists (vvc_filter_template.c:387-392) and alf_get_idx's arg_var (:272): generated from the
// reference's initialisers (tables_small.inc; tables.cpp exports the same text for the table check), once as device arrays and once
// as compile-time constants for the packed forms below
#define VVC355_TABLE(type, name, count) __device__ static const type t_##name[count]
#include "tables_small.inc"
#undef VVC355_TABLE
#define VVC355_TABLE(type, name, count) static constexpr type c_##name[count]
#include "tables_small.inc"
#undef VVC355_TABLE
#define kAlfVarTab t_alf_arg_var
// luma diamond taps (dy, dx), paired with (-dy, -dx): vvc_filter_template.c:102-113
__device__ static constexpr int8_t kLumaTap[12][2] = {
    { 3, 0 }, { 2, 1 }, { 2, 0 }, { 2, -1 }, { 1, 2 }, { 1, 1 }, { 1, 0 }, { 1, -1 }, { 1, -2 }, { 0, 3 }, { 0, 2 }, { 0, 1 },
};
// chroma 5x5 diamond taps (dy, dx), each paired with its mirror: (2,0) (1,1) (1,0) (1,-1) (0,2) (0,1) — unrolled in alf_chroma_kernel

// taps first .. first + 7 of transpose t's index list, one 4-bit field per tap
constexpr uint32_t alf_perm_nibbles(int t, int first)
{
    uint32_t v = 0;
    for (int k = first; k < 12 && k < first + 8; k++)
        v |= (uint32_t)c_alf_transpose_index[t * 12 + k] << (4 * (k - first));
    return v;
}

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
template <int BD, int TW = kTileW, int RA = 3>
__device__ __forceinline__ void stage_tile(uint16_t (*tile)[TW], const vvc355_alf_job &job, int y_base, int rows, int apron)
{
    using px_t = typename Px<BD>::type;
    const uint8_t *src = (const uint8_t *)job.src;
    const int w = job.w, h = job.h;
    const int x_min = -min((int)job.ext_l, apron), x_max = w - 1 + min((int)job.ext_r, apron);
    const int y_min = -min((int)job.ext_t, apron), y_max = h - 1 + min((int)job.ext_b, apron);
    const int nchunk = min(TW / 8, (w + 23) >> 3);      // columns -8 .. w+7 only
    const bool aligned = (((uintptr_t)src | (uintptr_t)job.src_stride) & (sizeof(px_t) * 8 - 1)) == 0;
    // (wave-uniform base, unsigned per-lane offset) addressing: the base sits RA rows and eight samples before the rectangle
    const uint8_t *src_m = src - RA * job.src_stride - 8 * (int)sizeof(px_t);

    const uint32_t recip = (65536u + nchunk - 1) / nchunk;          // wave-uniform; i / nchunk == (i * recip) >> 16 for i < 4096, nchunk <= 18
    for (int i = threadIdx.x; i < (rows + 2 * RA) * nchunk; i += blockDim.x) {
        const int r = (int)(((uint32_t)i * recip) >> 16), k = i - r * nchunk;
        const int y = clip3(y_base + r - RA, y_min, y_max);
        const int c0 = k * 8 - kColOff;
        const uint32_t row_m = (uint32_t)__mul24(y + RA, job.src_stride) + 8 * (int)sizeof(px_t);       // y >= -RA: byte offset of column 0
        uint4 q;
        if (aligned && c0 >= 0 && c0 + 8 <= w) {
            if (BD > 8) {
                q = gld_at<uint4>(src_m, row_m + c0 * (int)sizeof(px_t));
            } else {
                const uint2 t = gld_at<uint2>(src_m, row_m + c0);
                q.x = __builtin_amdgcn_perm(0, t.x, 0x0c010c00u); q.y = __builtin_amdgcn_perm(0, t.x, 0x0c030c02u);
                q.z = __builtin_amdgcn_perm(0, t.y, 0x0c010c00u); q.w = __builtin_amdgcn_perm(0, t.y, 0x0c030c02u);
            }
        } else {
            uint32_t v[8];
#pragma unroll
            for (int j = 0; j < 8; j++)
                v[j] = gld_at<px_t>(src_m, row_m + clip3(c0 + j, x_min, x_max) * (int)sizeof(px_t));
            q.x = v[0] | (v[1] << 16); q.y = v[2] | (v[3] << 16); q.z = v[4] | (v[5] << 16); q.w = v[6] | (v[7] << 16);
        }
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
// MODE 2: classify only (alf.classify slot): writes class_idx / transpose_idx ints.
// (The fused form the caller chains per CTB, vvc_filter.c:1139-1186, is alf_ctb_kernel below.)
template <int BD, int MODE>
__global__ __launch_bounds__(256) void alf_luma_kernel(const vvc355_alf_job *__restrict__ jobs)
{
    __shared__ __attribute__((aligned(16))) uint16_t tile[kTileH][kTileW];
    const int wg = xcd_chunked(blockIdx.x, gridDim.x);       // an XCD's L2 sees a contiguous run of CTBs (shared aprons)
    const vvc355_alf_job job = jobs[wg >> 2];      // (scalar load_uniform measured 6 % slower here: this kernel is VALU-bound and register-tight)
    const int y_base = (wg & 3) * kStripH;
    if (y_base >= job.h)
        return;
    const int rows = min(kStripH, job.h - y_base);
    stage_tile<BD>(tile, job, y_base, rows, 3);
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
        ((int *)job.coeff)[blk] = cls;
        ((int *)job.clip)[blk] = tr;
        return;
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

// ---------------------------------------------------------------------------------------------- CTB kernel (stage driver)
//
// What ff_vvc_alf_filter does for one CTB (vvc_filter.c:1254-1318) in one workgroup per 32-row strip of the CTB: the luma strip
// (+ 3-sample apron) is staged once and serves the classification, the 7x7 diamond AND the CC-ALF of both chroma components;
// chroma is read once and written once (4:2:0: CHROMA = true; other formats run the separate chroma / CC kernels).
//
//  * Classification without recomputation: a 2x2 "cell" of the subsampled Laplacian grid (:319-343) belongs to the 8x8 windows
//    of four 4x4 blocks.  Every lane computes the four cells inside its own block once (from its register window), lanes 0..99
//    the ring of cells around the strip (from the LDS tile); the cells go to LDS as packed 16-bit {V|H, D0|D1} and every block
//    sums its 4x4 cells from there (:345-380).  16 cells x 2 samples x 2 * 4095 fits 16 bits horizontally at every bit depth;
//    the vertical sum is packed up to 10 bits and 32-bit at 12.
//  * Filter without clamps where they cannot bite: alf_clip (:38) with clip = 2^bd (clip index 0: every fixed filter set,
//    vvc_filter.c:1147-1158, and every APS class without non-linear clipping) never changes |a - cur| < 2^bd, so
//        sum = sum_k f[k] * (a_k + b_k) - 2 * cur * sum_k f[k]                                  (identical modulo 2^32)
//    and the tap sums become v_dot2_i32_i16 of the window's own sample pairs against coefficient pairs laid out per column
//    parity: 16 dot products + one mad for the centre per sample instead of 72 sub / med3 / mad.  The workgroup takes this path
//    when every (class, tap) of the CTB's filter set has clip index 0 or coefficient 0; otherwise the clamped form runs.
// what the stage driver's builder leaves per (CTB, chroma component) for the CTB kernel, 64 bytes, fetched with scalar loads
struct AlfChromaParams {
    int16_t clip[6];         // clip values of the chroma alternative (what vvc355_alf_job.clip of the chroma job points at)
    int16_t clamps;          // some tap has a clip value below 2^bd and a non-zero coefficient
    int16_t cc_on;           // CC-ALF on for this CTB component
    int16_t f[6];            // chroma coefficients
    int16_t pad0[2];
    int16_t g[7];            // CC-ALF coefficients
    int16_t pad1[9];
};
static_assert(sizeof(AlfChromaParams) == 64, "scalar-loaded whole dwords");

static constexpr int kCellRows = kStripH / 2 + 2;      // cell row cr: sample A on strip row 2 * cr - 2
static constexpr int kCellCols = 34;                   // block columns -1 .. 32 (index = column + 1)
static constexpr int kCTileW   = 80;                   // 4:2:0 chroma strip: columns -8 .. 71
static constexpr int kCTileH   = kStripH / 2 + 4;      // rows -2 .. 17

typedef short alf_v2s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int sdot2(uint32_t a, uint32_t b, int acc)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(alf_v2s, a), __builtin_bit_cast(alf_v2s, b), acc, false);
}
// (low half of a) | (low half of b) << 16, whatever the upper halves hold
__device__ __forceinline__ uint32_t pk_ll(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x05040100u); }
// the same written so that wave-uniform operands stay on the scalar unit
__device__ __forceinline__ uint32_t upk_ll(uint32_t a, uint32_t b) { return (a & 0xffffu) | (b << 16); }
__device__ __forceinline__ uint32_t pk_l0(uint32_t a) { return a & 0xffffu; }
__device__ __forceinline__ uint32_t pk_0l(uint32_t a) { return a << 16; }

// One cell of the Laplacian grid (vvc_filter_template.c:325-343): samples A = (row 1, column ca) and B = (row 2, ca + 1) of four
// rows given as dwords of two samples (even column in the low half): up = the row above A (A's own row at the virtual boundary,
// :323-324), ra = A's row, rb = B's row, dn = the row below B (B's own row at the boundary, :321-322); u0 / a0 / b0 / d0 hold
// columns ca-2 ca-1, *1 columns ca ca+1, *2 columns ca+2 ca+3.  A's term sits in the low half, B's in the high half of every
// operand, so one v_sad_u16 per direction adds |2A - n1 - n2| + |2B - n1' - n2'|.  Returns V | H << 16 and D0 | D1 << 16.
__device__ __forceinline__ uint32_t lo_hi(uint32_t lo_src, uint32_t hi_src) { return __builtin_amdgcn_perm(hi_src, lo_src, 0x07060100u); }    // lo(lo_src) | hi(hi_src)
__device__ __forceinline__ uint32_t hi_lo(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x05040302u); }                       // hi(a) | lo(b) << 16
__device__ __forceinline__ void grad_cell_pk(uint32_t u0, uint32_t u1, uint32_t a0, uint32_t a1, uint32_t a2, uint32_t b0, uint32_t b1, uint32_t b2,
                                             uint32_t d1, uint32_t d2, uint32_t &vh, uint32_t &dd)
{
    const uint32_t x2 = lo_hi(a1, b1) << 1;                                    // (2A, 2B)
    // (neighbour of A, neighbour of B) pairs; their sums stay below 2^16, so a plain add serves both halves
    const uint32_t v  = __builtin_amdgcn_sad_u16(x2, lo_hi(u1, a1) + lo_hi(b1, d1), 0u);        // A: up(ca) + rb(ca);        B: ra(cb) + dn(cb)
    const uint32_t h  = __builtin_amdgcn_sad_u16(x2, hi_lo(a0, b1) + hi_lo(a1, b2), 0u);        // A: ra(ca-1) + ra(ca+1);    B: rb(cb-1) + rb(cb+1)
    const uint32_t e0 = __builtin_amdgcn_sad_u16(x2, hi_lo(u0, a1) + hi_lo(b1, d2), 0u);        // A: up(ca-1) + rb(ca+1);    B: ra(cb-1) + dn(cb+1)
    const uint32_t e1 = __builtin_amdgcn_sad_u16(x2, hi_lo(u1, a2) + hi_lo(b0, d1), 0u);        // A: up(ca+1) + rb(ca-1);    B: ra(cb+1) + dn(cb-1)
    vh = v | (h << 16);
    dd = e0 | (e1 << 16);
}

// 7x7 diamond without clamps on one 4x4 block from the register window (KIND as in filter_block_regs).  ev[k] holds tap k's
// coefficient in its low half.  Window column c = 4 + j of row r sits in dword c >> 1; per column parity the taps of a row pair
// up with the dwords as they are: even c: row 0 (c-4 c-3)(c-2 c-1)(c c+1)(c+2 c+3), odd c: (c-3 c-2)(c-1 c)(c+1 c+2)(c+3 c+4), ...
template <int BD, int KIND>
__device__ __forceinline__ void filter_block_fast(const Win &win, const uint32_t (&ev)[12], int cen, uint8_t *drow, int dst_stride)
{
    // cen = -2 * sum f.  ((sum + 64) >> 7) + cur == (sum + 64 + 128 * cur) >> 7: the centre sample's weight becomes cen + 128
    // (cen + 1024 on the rows that round with >> 10) and sits in the one free slot of the row-0 pairs; the caller has checked that
    // both fit 16 bits.
    const uint32_t c7 = (uint32_t)(cen + 128), c10 = (uint32_t)(cen + 1024);
    uint32_t outp[4][2];
#pragma unroll
    for (int par = 0; par < 2; par++) {
        uint32_t z0[4], zc7, zc10, p1[3], m1[3], p2[2], m2[2], p3;
        if (par == 0) {
            z0[0] = pk_0l(ev[9]); z0[1] = pk_ll(ev[10], ev[11]); z0[3] = pk_ll(ev[10], ev[9]);
            zc7 = pk_ll(c7, ev[11]); zc10 = pk_ll(c10, ev[11]);
            p1[0] = pk_ll(ev[8], ev[7]); p1[1] = pk_ll(ev[6], ev[5]); p1[2] = pk_l0(ev[4]);
            m1[0] = pk_ll(ev[4], ev[5]); m1[1] = pk_ll(ev[6], ev[7]); m1[2] = pk_l0(ev[8]);
            p2[0] = pk_0l(ev[3]); p2[1] = pk_ll(ev[2], ev[1]);
            m2[0] = pk_0l(ev[1]); m2[1] = pk_ll(ev[2], ev[3]);
            p3 = pk_l0(ev[0]);
        } else {
            z0[0] = pk_ll(ev[9], ev[10]); z0[2] = pk_ll(ev[11], ev[10]); z0[3] = pk_l0(ev[9]);
            zc7 = pk_ll(ev[11], c7); zc10 = pk_ll(ev[11], c10);
            p1[0] = pk_0l(ev[8]); p1[1] = pk_ll(ev[7], ev[6]); p1[2] = pk_ll(ev[5], ev[4]);
            m1[0] = pk_0l(ev[4]); m1[1] = pk_ll(ev[5], ev[6]); m1[2] = pk_ll(ev[7], ev[8]);
            p2[0] = pk_ll(ev[3], ev[2]); p2[1] = pk_l0(ev[1]);
            m2[0] = pk_ll(ev[1], ev[2]); m2[1] = pk_l0(ev[3]);
            p3 = pk_0l(ev[0]);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int dist = KIND == 0 ? 3 : KIND == 1 ? 3 - i : i;
            const int r0 = 3 + i, d1 = dist < 1 ? dist : 1, d2 = dist < 2 ? dist : 2, d3 = dist < 3 ? dist : 3;
            z0[par ? 1 : 2] = dist == 0 ? zc10 : zc7;
#pragma unroll
            for (int jj = 0; jj < 2; jj++) {
                const int c = 4 + par + 2 * jj, m = c >> 1;
                const int b0 = par ? m - 1 : m - 2, b2 = par ? m : m - 1;
                int sum = dist == 0 ? 512 : 64;
#pragma unroll
                for (int q = 0; q < 4; q++) sum = sdot2(win.d[r0][b0 + q], z0[q], sum);
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    sum = sdot2(win.d[r0 + d1][m - 1 + q], p1[q], sum);
                    sum = sdot2(win.d[r0 - d1][m - 1 + q], m1[q], sum);
                }
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    sum = sdot2(win.d[r0 + d2][b2 + q], p2[q], sum);
                    sum = sdot2(win.d[r0 - d2][b2 + q], m2[q], sum);
                }
                sum = sdot2(win.d[r0 + d3][m], p3, sum);
                sum = sdot2(win.d[r0 - d3][m], p3, sum);
                const int o = clip_px<BD>(dist == 0 ? sum >> 10 : sum >> 7);
                outp[i][jj] = par ? outp[i][jj] | ((uint32_t)o << 16) : (uint32_t)o;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint8_t *d = drow + (ptrdiff_t)i * dst_stride;
        if (BD > 8)
            gst<uint2>(d, make_uint2(outp[i][0], outp[i][1]));
        else
            gst<uint32_t>(d, __builtin_amdgcn_perm(outp[i][1], outp[i][0], 0x06040200u));
    }
}

// chroma 5x5 diamond (:137) on four consecutive samples x .. x + 3 of one row from a 5-row x 8-column register window (rows
// -d2, -d1, 0, +d1, +d2 already folded onto the virtual boundary; columns x - 2 .. x + 5, two per dword); clamped form
template <int BD>
__device__ __forceinline__ void chroma_quad(const uint32_t (&win)[5][4], const int (&f)[6], const int (&c)[6], int dist, int (&out)[4])
{
#define WPX(r, c) ((int)((win[r][(c) >> 1] >> (((c) & 1) * 16)) & 0xffff))
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
}

// the same without clamps (every clip value 2^bd, or a zero coefficient): 9 dot products per sample, the centre weight
// cen = -2 * sum f and the "+ cur" in the free slot of the row-0 pairs (the caller has checked that cen + 128 / + 1024 fit 16 bits)
template <int BD>
__device__ __forceinline__ void chroma_quad_fast(const uint32_t (&win)[5][4], const int (&f)[6], int cen, int dist, int (&out)[4])
{
    const uint32_t F0 = f[0], F1 = f[1], F2 = f[2], F3 = f[3], F4 = f[4], F5 = f[5];
    const uint32_t cw = (uint32_t)(cen + (dist == 0 ? 1024 : 128));
    const uint32_t ce = upk_ll(cw, F5), co = upk_ll(F5, cw);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int c = j + 2, m = c >> 1;
        int sum = dist == 0 ? 512 : 64;
        if (!(c & 1)) {                         // dwords m-1 (c-2 c-1), m (c c+1), m+1 (c+2 c+3)
            sum = sdot2(win[2][m - 1], upk_ll(F4, F5), sum); sum = sdot2(win[2][m], ce, sum); sum = sdot2(win[2][m + 1], pk_l0(F4), sum);
            sum = sdot2(win[3][m - 1], pk_0l(F3), sum); sum = sdot2(win[3][m], upk_ll(F2, F1), sum);
            sum = sdot2(win[1][m - 1], pk_0l(F1), sum); sum = sdot2(win[1][m], upk_ll(F2, F3), sum);
            sum = sdot2(win[4][m], pk_l0(F0), sum); sum = sdot2(win[0][m], pk_l0(F0), sum);
        } else {                                // dwords m-1 (c-3 c-2), m (c-1 c), m+1 (c+1 c+2)
            sum = sdot2(win[2][m - 1], pk_0l(F4), sum); sum = sdot2(win[2][m], co, sum); sum = sdot2(win[2][m + 1], upk_ll(F5, F4), sum);
            sum = sdot2(win[3][m], upk_ll(F3, F2), sum); sum = sdot2(win[3][m + 1], pk_l0(F1), sum);
            sum = sdot2(win[1][m], upk_ll(F1, F2), sum); sum = sdot2(win[1][m + 1], pk_l0(F3), sum);
            sum = sdot2(win[4][m], pk_0l(F0), sum); sum = sdot2(win[0][m], pk_0l(F0), sum);
        }
        out[j] = clip_px<BD>(dist == 0 ? sum >> 10 : sum >> 7);
    }
}

template <int BD, bool CHROMA>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 8))) void alf_ctb_kernel(const vvc355_alf_job *__restrict__ luma_jobs, const vvc355_alf_job *__restrict__ chroma_jobs,
                                                      const vvc355_alf_job *__restrict__ cc_jobs, int log2_strips)
{
    __shared__ __attribute__((aligned(16))) uint16_t tile[kTileH][kTileW];
    __shared__ __attribute__((aligned(16))) uint32_t ftab[25 * 12];             // [class][tap] = coeff | clip << 16 (untransposed)
    __shared__ __attribute__((aligned(16))) uint4 cells[kCellRows][kCellCols];   // {left cell V|H, D0|D1, right cell V|H, D0|D1} of a block
    __shared__ __attribute__((aligned(16))) uint16_t ctile[CHROMA ? 2 : 1][CHROMA ? kCTileH : 1][kCTileW];
    const int wg = xcd_chunked(blockIdx.x, gridDim.x);
    const int ctb = wg >> log2_strips;
    const vvc355_alf_job job = load_uniform(luma_jobs + ctb);
    const int y_base = (wg & ((1 << log2_strips) - 1)) * kStripH;
    if (y_base >= job.h)
        return;
    const int rows = min(kStripH, job.h - y_base);
    // the CTB's filter set: every class's filter (class_to_filt resolved) with its clip values; does any tap clamp?
    int clamps = 0;
    {
        const int16_t *coeff_set = (const int16_t *)job.coeff;
        const uint8_t *clip_idx = (const uint8_t *)job.clip, *c2f = (const uint8_t *)job.class_to_filt;
        for (int e = threadIdx.x; e < 25 * 12; e += blockDim.x) {
            const int cls = (e * 5462) >> 16, k = e - cls * 12;               // e / 12 for e < 300
            const int q = gld<uint8_t>(clip_idx + e);
            const int cv = 1 << (BD - (q == 0 ? 0 : 2 * q + 1));       // {2^bd, 2^(bd-3), 2^(bd-5), 2^(bd-7)}
            const int fv = gld<int16_t>(coeff_set + gld<uint8_t>(c2f + cls) * 12 + k);
            clamps |= (q != 0) & (fv != 0);
            ftab[e] = (uint32_t)(uint16_t)fv | ((uint32_t)cv << 16);
        }
    }
    stage_tile<BD>(tile, job, y_base, rows, 3);
    if (CHROMA) {
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const vvc355_alf_job cj = load_uniform(chroma_jobs + 2 * ctb + c);
            stage_tile<BD, kCTileW, 2>(ctile[c], cj, y_base >> 1, rows >> 1, 2);
        }
    }
    const bool noclip = !__syncthreads_or(clamps);

    const int bx = threadIdx.x & 31, by = threadIdx.x >> 5;
    const int x = bx * 4, yl = by * 4, yb = y_base + yl;
    const bool active = x < job.w && yl < rows;
    const int vb_pos = job.vb_pos;

    Win win;
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint2 *p = (const uint2 *)&tile[yl + r][x + kColOff - 4];
        const uint2 a = p[0], b = p[1], c = p[2];
        win.d[r][0] = a.x; win.d[r][1] = a.y; win.d[r][2] = b.x; win.d[r][3] = b.y; win.d[r][4] = c.x; win.d[r][5] = c.y;
    }

    // ---- Laplacian cells: the four inside this lane's block (cell rows 2 * by + 1, + 2; sample A on window rows 3 and 5) ...
    if (active) {
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int ra = 3 + 2 * q, ya = yb + 2 * q;
            const bool fold_up = ya == vb_pos, fold_dn = ya == vb_pos - 2;
            uint32_t up[6], dn[6];
#pragma unroll
            for (int m = 1; m < 5; m++) {
                up[m] = fold_up ? win.d[ra][m] : win.d[ra - 1][m];
                dn[m] = fold_dn ? win.d[ra + 1][m] : win.d[ra + 2][m];
            }
            const uint32_t *A = win.d[ra], *B = win.d[ra + 1];
            uint4 e;
            grad_cell_pk(up[1], up[2], A[1], A[2], A[3], B[1], B[2], B[3], dn[2], dn[3], e.x, e.y);      // A on window column 4
            grad_cell_pk(up[2], up[3], A[2], A[3], A[4], B[2], B[3], B[4], dn[3], dn[4], e.z, e.w);      // A on window column 6
            cells[2 * by + 1 + q][bx + 1] = e;
        }
    }
    // ... and the ring around the strip, from the tile: cell row 0, cell row ncr + 1, block column -1, block column w / 4
    {
        const int ncr = rows >> 1, nbx = job.w >> 2;
        const int e = threadIdx.x;
        if (e < 68 + 2 * ncr) {
            int cr, bc;
            if (e < 34)            { cr = 0; bc = e - 1; }
            else if (e < 68)       { cr = ncr + 1; bc = e - 35; }
            else if (e < 68 + ncr) { cr = e - 67; bc = -1; }
            else                   { cr = e - 67 - ncr; bc = nbx; }
            const int ya = y_base - 2 + 2 * cr, ta = 2 * cr + 1;
            const int rsel[4] = { ya == vb_pos ? ta : ta - 1, ta, ta + 1, ya == vb_pos - 2 ? ta + 1 : ta + 2 };
            uint32_t R[4][4];                       // columns 4 bc - 2 .. 4 bc + 5 of the four rows
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint32_t *p = (const uint32_t *)&tile[rsel[r]][4 * bc + kColOff - 2];
#pragma unroll
                for (int m = 0; m < 4; m++) R[r][m] = p[m];
            }
            uint4 v;
            grad_cell_pk(R[0][0], R[0][1], R[1][0], R[1][1], R[1][2], R[2][0], R[2][1], R[2][2], R[3][1], R[3][2], v.x, v.y);
            grad_cell_pk(R[0][1], R[0][2], R[1][1], R[1][2], R[1][3], R[2][1], R[2][2], R[2][3], R[3][2], R[3][3], v.z, v.w);
            cells[cr][bc + 1] = v;
        }
    }
    __syncthreads();

    if (active) {
        // ---- class and transpose of the block from its 4 x 4 cells (:345-380)
        int first = 0, last = 4, ac = 2;
        if (yb + 4 == vb_pos) { last = 3; ac = 3; }
        else if (yb == vb_pos) { first = 1; ac = 3; }
        int sum[4] = { 0, 0, 0, 0 };
        uint32_t svh = 0, sdd = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint4 *row = &cells[2 * by + i][bx];
            const uint2 l = *(const uint2 *)&row[0].z;          // right cell of the block on the left
            const uint4 o = row[1];
            const uint2 r = *(const uint2 *)&row[2].x;          // left cell of the block on the right
            const uint32_t hvh = l.x + o.x + o.z + r.x, hdd = l.y + o.y + o.w + r.y;     // halves stay below 2^16: no carry between them
            const bool on = i >= first && i < last;
            if (BD <= 10) {
                svh += on ? hvh : 0u;
                sdd += on ? hdd : 0u;
            } else if (on) {
                sum[0] += hvh & 0xffff; sum[1] += hvh >> 16; sum[2] += hdd & 0xffff; sum[3] += hdd >> 16;
            }
        }
        if (BD <= 10) { sum[0] = svh & 0xffff; sum[1] = svh >> 16; sum[2] = sdd & 0xffff; sum[3] = sdd >> 16; }
        int cls, tr;
        block_class<BD>(sum, ac, cls, tr);

        // alf_recon_coeff_and_clip (:383): tap k of the block = entry kAlfPerm[transpose][k] of the class's filter; the four
        // permutations as 4-bit fields (taps 0..7 | taps 8..11)
        uint32_t ev[12];
        {
            constexpr uint32_t PL[4] = { alf_perm_nibbles(0, 0), alf_perm_nibbles(1, 0), alf_perm_nibbles(2, 0), alf_perm_nibbles(3, 0) };
            constexpr uint32_t PH[4] = { alf_perm_nibbles(0, 8), alf_perm_nibbles(1, 8), alf_perm_nibbles(2, 8), alf_perm_nibbles(3, 8) };
            const uint32_t pl = tr == 0 ? PL[0] : tr == 1 ? PL[1] : tr == 2 ? PL[2] : PL[3];
            const uint32_t ph = tr == 0 ? PH[0] : tr == 1 ? PH[1] : tr == 2 ? PH[2] : PH[3];
            const uint32_t *row = &ftab[cls * 12];
#pragma unroll
            for (int k = 0; k < 12; k++)
                ev[k] = row[((k < 8 ? pl : ph) >> (4 * (k & 7))) & 15];
        }

        uint8_t *dst = (uint8_t *)job.dst;
        uint8_t *drow = dst + row_off(yb, job.dst_stride) + x * (BD > 8 ? 2 : 1);
        // rows of this block closer than 3 to the virtual boundary fold their taps
        const bool near_vb = vb_dist(yb, vb_pos) < 3 || vb_dist(yb + 3, vb_pos) < 3 || (yb < vb_pos && yb + 3 >= vb_pos);
        const int kind = !near_vb ? 0 : yb + 4 == vb_pos ? 1 : yb == vb_pos ? 2 : 3;
        int cen = 0;
#pragma unroll
        for (int k = 0; k < 12; k++) cen += (int)(int16_t)(ev[k] & 0xffff);
        cen *= -2;
        // the clamp-free form carries the centre weight as a 16-bit operand (always true for coefficients in the standard's range)
        if (noclip && kind < 3 && cen >= -32768 && cen + 1024 <= 32767) {
            if (kind == 0)      filter_block_fast<BD, 0>(win, ev, cen, drow, job.dst_stride);
            else if (kind == 1) filter_block_fast<BD, 1>(win, ev, cen, drow, job.dst_stride);
            else                filter_block_fast<BD, 2>(win, ev, cen, drow, job.dst_stride);
        } else {
            // clamped form, any boundary position: one block row at a time from the tile, tap rows folded onto the virtual boundary
            // when they are fetched (:80-96), so one body serves every row
            int f[12], c[12];
#pragma unroll
            for (int k = 0; k < 12; k++) {
                f[k] = (int)(int16_t)(ev[k] & 0xffff);
                c[k] = (int)(ev[k] >> 16);
            }
#pragma unroll 1
            for (int i = 0; i < 4; i++) {
                const int dist = vb_dist(yb + i, vb_pos), tr0 = yl + i + 3;
                const int d1 = min(1, dist), d2 = min(2, dist), d3 = min(3, dist);
                const int rr[7] = { tr0 - d3, tr0 - d2, tr0 - d1, tr0, tr0 + d1, tr0 + d2, tr0 + d3 };
                uint32_t R[7][6];
#pragma unroll
                for (int r = 0; r < 7; r++) {
                    const uint2 *p = (const uint2 *)&tile[rr[r]][x + kColOff - 4];
                    const uint2 a = p[0], b = p[1], cc = p[2];
                    R[r][0] = a.x; R[r][1] = a.y; R[r][2] = b.x; R[r][3] = b.y; R[r][4] = cc.x; R[r][5] = cc.y;
                }
#define RPX(r, q) ((int)((R[r][(q) >> 1] >> (((q) & 1) * 16)) & 0xffff))
                int out[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int cur = RPX(3, 4 + j);
                    int sum = 0;
#pragma unroll
                    for (int k = 0; k < 12; k++) {
                        const int dy = kLumaTap[k][0], dx = kLumaTap[k][1];
                        sum = mad24(f[k], clamp_sym(RPX(3 + dy, 4 + j + dx) - cur, c[k]) + clamp_sym(RPX(3 - dy, 4 + j - dx) - cur, c[k]), sum);
                    }
                    out[j] = clip_px<BD>((dist == 0 ? (sum + 512) >> 10 : (sum + 64) >> 7) + cur);
                }
#undef RPX
                uint8_t *d = drow + row_off(i, job.dst_stride);
                if (BD > 8)
                    gst<uint2>(d, make_uint2(out[0] | (out[1] << 16), out[2] | (out[3] << 16)));
                else
                    gst<uint32_t>(d, out[0] | (out[1] << 8) | (out[2] << 16) | (out[3] << 24));
            }
        }
    }

    if (CHROMA) {
        // ---- 4:2:0 chroma of the strip: lane = four consecutive samples of one row, Cb then Cr; the co-located luma for CC-ALF
        // (:223) comes from the luma tile, whose replication at unreadable sides is the padded buffer's (vvc_filter.c:1105-1137)
        const int cyl = threadIdx.x >> 4, cx = (threadIdx.x & 15) * 4;
        if (cyl >= (rows >> 1) || cx >= (job.w >> 1))
            return;
        const int y = (y_base >> 1) + cyl;
        // luma rows around row 2 * y (tile row 2 * cyl + 3) as CC-ALF folds them onto the luma virtual boundary (:233-241)
        const int ly = 2 * y;
        int up = -1, dn = 1, dn2 = 2;
        if (ly == vb_pos - 2 || ly == vb_pos + 1) dn2 = 1;
        else if (ly == vb_pos - 1 || ly == vb_pos) up = dn = dn2 = 0;
        const int lt = 2 * cyl + 3;
        // dwords (2cx-2 2cx-1) .. (2cx+6 2cx+7) of the centre row and of the row below; the even columns' dwords of the other two
        uint32_t lc[5], ld[5], lu[4], l2[4];
        {
            const uint32_t *rc = (const uint32_t *)&tile[lt][2 * cx + kColOff - 2], *rd = (const uint32_t *)&tile[lt + dn][2 * cx + kColOff - 2];
            const uint32_t *ru = (const uint32_t *)&tile[lt + up][2 * cx + kColOff], *r2 = (const uint32_t *)&tile[lt + dn2][2 * cx + kColOff];
#pragma unroll
            for (int m = 0; m < 5; m++) { lc[m] = rc[m]; ld[m] = rd[m]; }
#pragma unroll
            for (int m = 0; m < 4; m++) { lu[m] = ru[m]; l2[m] = r2[m]; }
        }
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const vvc355_alf_job cj = load_uniform(chroma_jobs + 2 * ctb + c);
            const AlfChromaParams P = load_uniform((const AlfChromaParams *)cj.clip);      // the stage driver's block: all scalar
            int f[6], cl[6];
#pragma unroll
            for (int k = 0; k < 6; k++) { f[k] = P.f[k]; cl[k] = P.clip[k]; }
            const bool cclamps = P.clamps != 0;
            const int dist = vb_dist(y, cj.vb_pos), tr0 = cyl + 2;
            const int d1 = min(1, dist), d2 = min(2, dist);
            uint32_t w5[5][4];
            {
                const int rr[5] = { tr0 - d2, tr0 - d1, tr0, tr0 + d1, tr0 + d2 };
#pragma unroll
                for (int r = 0; r < 5; r++) {
                    const uint32_t *p = (const uint32_t *)&ctile[c][rr[r]][cx + kColOff - 2];
#pragma unroll
                    for (int m = 0; m < 4; m++) w5[r][m] = p[m];
                }
            }
            int out[4];
            const int ccen = -2 * (f[0] + f[1] + f[2] + f[3] + f[4] + f[5]);
            if (cclamps || ccen < -32768 || ccen + 1024 > 32767) chroma_quad<BD>(w5, f, cl, dist, out);
            else                                                 chroma_quad_fast<BD>(w5, f, ccen, dist, out);
            if (P.cc_on) {
                int g[7];
#pragma unroll
                for (int k = 0; k < 7; k++) g[k] = P.g[k];
                const uint32_t G0 = g[0], G1 = g[1], G2 = g[2], G3 = g[3], G4 = g[4], G5 = g[5], G6 = g[6];
                const int gcen = -(g[0] + g[1] + g[2] + g[3] + g[4] + g[5] + g[6]);
                const uint32_t k_c0 = pk_0l(G1), k_c1 = pk_0l(G2), k_d0 = pk_0l(G3), k_d1 = upk_ll(G4, G5), k_u = pk_l0(G0), k_2 = pk_l0(G6);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    int s = sdot2(lc[j], k_c0, 0);
                    s = sdot2(lc[j + 1], k_c1, s);
                    s = sdot2(ld[j], k_d0, s);
                    s = sdot2(ld[j + 1], k_d1, s);
                    s = sdot2(lu[j], k_u, s);
                    s = sdot2(l2[j], k_2, s);
                    s = mad24(gcen, (int)(lc[j + 1] & 0xffff), s);
                    s = clip3((s + 64) >> 7, -(1 << (BD - 1)), (1 << (BD - 1)) - 1);
                    out[j] = clip_px<BD>(out[j] + s);
                }
            }
            uint8_t *d = (uint8_t *)cj.dst + row_off(y, cj.dst_stride);
            if (BD > 8)
                gst<uint2>(d + cx * 2, make_uint2(out[0] | (out[1] << 16), out[2] | (out[3] << 16)));
            else
                gst<uint32_t>(d + cx, out[0] | (out[1] << 8) | (out[2] << 16) | (out[3] << 24));
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
    const int cls = class_idx[b], idx = t_alf_transpose_index[transpose_idx[b] * 12 + j];
    const int q = clip_idx_set[cls * 12 + idx];
    coeff[i] = coeff_set[class_to_filt[cls] * 12 + idx];
    clip[i] = (int16_t)(1 << (BD - (q == 0 ? 0 : 2 * q + 1)));
}

// ---------------------------------------------------------------------------------------------- stage driver

// Zero filter set: a CTB component with alf_ctb_flag off passes through (sum = 0 -> dst = src); also the harmless operand of
// CC-ALF jobs that are switched off (w = h = 0).
__device__ static const int16_t kAlfZeroSet[25 * 12] = { 0 };

// ff_vvc_alf_filter (vvc_filter.c:1254-1318) per CTB as a descriptor builder: three lanes per CTB write its luma job, and per
// chroma component the chroma job, the CC-ALF job and the CTB kernel's parameter block.  edges[] (:1264-1278) become ext_* = 0 (replicate) / 3 (read the neighbour in place).
template <int BD>
__global__ void alf_build_kernel(const vvc355_alf_frame *__restrict__ frame, int n_ctbs, vvc355_alf_job *luma, vvc355_alf_job *chroma,
                                 vvc355_alf_job *cc, AlfChromaParams *params)
{
    const vvc355_alf_frame F = load_uniform(frame);       // scalar loads, once: the fields are read dozens of times
    const vvc355_alf_frame *fp = &F;
    // three lanes per CTB (luma job; Cb jobs; Cr jobs): the chain of dependent table reads per lane is a third as long
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int rs = t / 3, comp = t - rs * 3;
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
    if (comp == 0) {
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
        return;
    }
    if (fp->n_comp < 3)
        return;
    const int hs = fp->hs, vs = fp->vs;
    const int x0 = (xc * ctb_size) >> hs, y0 = (yc * ctb_size) >> vs;
    const int w = min((fp->width >> hs) - x0, ctb_size >> hs), h = min((fp->height >> vs) - y0, ctb_size >> vs);
    {
        const int c = comp;
        // ---- chroma: alf_filter_chroma (:1195)
        j.dst = fp->dst[c] + (uint64_t)y0 * fp->dst_stride[c] + x0 * px;
        j.src = fp->src[c] + (uint64_t)y0 * fp->src_stride[c] + x0 * px;
        j.dst_stride = fp->dst_stride[c]; j.src_stride = fp->src_stride[c];
        j.w = (int16_t)w; j.h = (int16_t)h; j.vb_pos = (int16_t)((ctb_size >> vs) - 2);
        j.class_to_filt = 0; j.hs = j.vs = 0;
        AlfChromaParams P = {};
        if (alf.ctb_flag[c]) {
            const int idx = alf.alt_idx[c - 1];
            const uint8_t *ci = (const uint8_t *)sl->chroma_clip_idx + idx * 6;
            const int16_t *cf = (const int16_t *)sl->chroma_coeff + idx * 6;
            for (int k = 0; k < 6; k++) {
                const int q = ci[k];
                P.clip[k] = (int16_t)(1 << (BD - (q == 0 ? 0 : 2 * q + 1)));      // alf_clip_from_idx (:1188)
                P.f[k] = cf[k];
                P.clamps |= q != 0 && cf[k] != 0;
            }
            j.coeff = sl->chroma_coeff + idx * 12;
        } else {
            j.coeff = (uint64_t)kAlfZeroSet;
        }
        AlfChromaParams *pp = params + (2 * rs + c - 1);
        j.clip = (uint64_t)pp->clip;
        chroma[2 * rs + c - 1] = j;
        // ---- CC-ALF: alf_filter_cc (:1212) on the co-located luma
        const uint64_t cc_set = sl->cc_coeff[c - 1];
        const bool on = alf.cc_idc[c - 1] && cc_set;
        j.src = fp->src[0] + (uint64_t)(y0 << vs) * fp->src_stride[0] + (x0 << hs) * px;
        j.src_stride = fp->src_stride[0];
        j.coeff = on ? cc_set + (alf.cc_idc[c - 1] - 1) * 14 : (uint64_t)kAlfZeroSet;
        j.clip = 0;
        P.cc_on = on;
        for (int k = 0; k < 7; k++) P.g[k] = on ? ((const int16_t *)j.coeff)[k] : 0;
        *pp = P;
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
        if (mode == 0) hipLaunchKernelGGL((alf_luma_kernel<BD, 0>), grid, block, 0, st, jobs);
        else           hipLaunchKernelGGL((alf_luma_kernel<BD, 2>), grid, block, 0, st, jobs);
    });
    HIP_CHECK(hipGetLastError());
}

// the CTB kernel: luma (fused classify + gather + filter) of n rectangles of at most max_h rows; with chroma / cc job arrays (two per
// rectangle, 4:2:0) it also filters both chroma components and applies CC-ALF
static void launch_ctb(int bd, const vvc355_alf_job *luma, const vvc355_alf_job *chroma, const vvc355_alf_job *cc, int n, hipStream_t st, int max_h = 128)
{
    if (n <= 0) return;
    const int log2_strips = max_h <= kStripH ? 0 : max_h <= 2 * kStripH ? 1 : 2;
    const dim3 grid(n << log2_strips), block(256);
    VVC355_BD_DISPATCH(bd, {
        if (chroma) hipLaunchKernelGGL((alf_ctb_kernel<BD, true>), grid, block, 0, st, luma, chroma, cc, log2_strips);
        else        hipLaunchKernelGGL((alf_ctb_kernel<BD, false>), grid, block, 0, st, luma, chroma, cc, log2_strips);
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
    if (fused) launch_ctb(bd, jobs_dev, nullptr, nullptr, n_jobs, (hipStream_t)stream);
    else       launch_luma(bd, 0, jobs_dev, n_jobs, (hipStream_t)stream);
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
    return (size_t)n_ctbs * (5 * sizeof(vvc355_alf_job) + 2 * sizeof(AlfChromaParams));
}

static int alf_frame_ctbs(const vvc355_alf_frame *frame_host)
{
    const int n = frame_host->ctb_width * frame_host->ctb_height;
    if (n > 0 && (frame_host->ctb_log2 < 5 || frame_host->ctb_log2 > 7 || (frame_host->width & 7) || (frame_host->height & 7))) {
        fprintf(stderr, "vvc_mi355: ALF frame %dx%d (CTB log2 %d) outside the driver's domain\n", frame_host->width, frame_host->height, frame_host->ctb_log2);
        abort();
    }
    return n;
}

void vvc355_alf_frame_build(void *stream, int bd, const vvc355_alf_frame *frame_dev, const vvc355_alf_frame *frame_host, void *work_dev)
{
    const int n = alf_frame_ctbs(frame_host);
    if (n <= 0) return;
    vvc355_alf_job *luma = (vvc355_alf_job *)work_dev, *chroma = luma + n, *cc = chroma + 2 * n;
    AlfChromaParams *params = (AlfChromaParams *)(cc + 2 * n);
    VVC355_BD_DISPATCH(bd, hipLaunchKernelGGL((alf_build_kernel<BD>), dim3((3 * n + 63) / 64), dim3(64), 0, (hipStream_t)stream, frame_dev, n, luma, chroma, cc, params));
    HIP_CHECK(hipGetLastError());
}

void vvc355_alf_frame_pass(void *stream, int bd, const vvc355_alf_frame *frame_dev, const vvc355_alf_frame *frame_host, void *work_dev)
{
    vvc355_alf_frame_build(stream, bd, frame_dev, frame_host, work_dev);
    vvc355_alf_frame_filter(stream, bd, frame_host, work_dev);
}

void vvc355_alf_frame_filter(void *stream, int bd, const vvc355_alf_frame *frame_host, const void *work_dev)
{
    const int n = alf_frame_ctbs(frame_host);
    if (n <= 0) return;
    const vvc355_alf_job *luma = (const vvc355_alf_job *)work_dev, *chroma = luma + n, *cc = chroma + 2 * n;
    hipStream_t st = (hipStream_t)stream;
    const int ctb_size = 1 << frame_host->ctb_log2;
    if (frame_host->n_comp >= 3 && frame_host->hs == 1 && frame_host->vs == 1) {
        launch_ctb(bd, luma, chroma, cc, n, st, ctb_size);         // 4:2:0: one kernel, every plane read once and written once
    } else {
        launch_ctb(bd, luma, nullptr, nullptr, n, st, ctb_size);
        if (frame_host->n_comp >= 3) {
            launch_chroma(bd, chroma, 2 * n, st, ctb_size >> frame_host->vs);
            launch_cc(bd, cc, 2 * n, st);
        }
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
