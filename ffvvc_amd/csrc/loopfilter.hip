// In-loop filter kernels for gfx950 other than ALF: LMCS luma mapping, SAO (band / edge / edge-restore) and the
// deblocking filter (luma incl. the long-tap filters, chroma, LADF level).
//
// Reference behaviour: libavcodec/vvc/vvc_filter_template.c:25 (lmcs), :466-804 (deblock decisions + long filters),
// libavcodec/h26x/h2656_sao_template.c:24-215, libavcodec/h26x/h2656_deblock_template.c:25-99, and for the batched SAO
// stage the caller's border rules libavcodec/vvc/vvc_filter.c:154-300.
#include "common.hpp"
#include "runtime.hpp"
#include "../../include/vvc_mi355.h"

namespace vvc355 {

// ------------------------------------------------------------------------------------------------ LMCS

// job.src0 = LUT (pixel-typed, 2^bd entries); in place on job.dst.  The job's LUT is staged in LDS once per workgroup; a lane
// maps 8 consecutive samples per step (one 16-byte load / store at 10-bit) when the rectangle's rows are vector-aligned.
template <int BD>
__global__ __launch_bounds__(256) void lmcs_kernel(const vvc355_blend_job *__restrict__ jobs)
{
    using px_t = typename Px<BD>::type;
    constexpr int VB = 8 * (int)sizeof(px_t);                 // bytes per 8-sample vector
    __shared__ __attribute__((aligned(16))) px_t lut_lds[1 << BD];
    const vvc355_blend_job job = load_uniform(jobs + (blockIdx.y));
    const uint8_t *lut = (const uint8_t *)job.src0;
    if ((job.src0 & 15) == 0) {
        for (int i = threadIdx.x; i < (int)((sizeof(px_t) << BD) / 16); i += 256)
            ((uint4 *)lut_lds)[i] = gld<uint4>(lut + i * 16);
    } else {
        for (int i = threadIdx.x; i < (1 << BD); i += 256)
            lut_lds[i] = (px_t)ld_px<BD>(lut, i);
    }
    __syncthreads();
    const int w = job.w, h = job.h;
    uint8_t *dst = (uint8_t *)job.dst;
    const bool vec = ((job.dst | (uint64_t)(uint32_t)job.dst_stride) & (VB - 1)) == 0;
    const int wv = vec ? w >> 3 : 0;                          // whole vectors per row
    for (int i = blockIdx.x * 256 + threadIdx.x; i < wv * h; i += gridDim.x * 256) {
        const int y = i / wv, xv = i - y * wv;
        uint8_t *p = dst + row_off(y, job.dst_stride) + xv * VB;
        px_t t[8];
        if (BD > 8) { const uint4 q = gld<uint4>(p); __builtin_memcpy(t, &q, sizeof(t)); }
        else { const uint2 q = gld<uint2>(p); __builtin_memcpy(t, &q, sizeof(t)); }
#pragma unroll
        for (int k = 0; k < 8; k++) t[k] = lut_lds[t[k]];
        if (BD > 8) { uint4 q; __builtin_memcpy(&q, t, sizeof(t)); gst<uint4>(p, q); }
        else { uint2 q; __builtin_memcpy(&q, t, sizeof(t)); gst<uint2>(p, q); }
    }
    const int x_tail = wv * 8, wt = w - x_tail;               // columns left to the per-sample path
    for (int i = blockIdx.x * 256 + threadIdx.x; i < wt * h; i += gridDim.x * 256) {
        const int y = i / wt, x = x_tail + i - y * wt;
        uint8_t *row = dst + row_off(y, job.dst_stride);
        st_px<BD>(row, x, lut_lds[ld_px<BD>(row, x)]);
    }
}

// ------------------------------------------------------------------------------------------------ SAO

__device__ static const uint8_t kSaoCat[5] = { 1, 2, 0, 3, 4 };
__device__ static const int8_t kSaoNb[4][4] = { { -1, 0, 1, 0 }, { 0, -1, 0, 1 }, { -1, -1, 1, 1 }, { 1, -1, -1, 1 } };

// true when restore rules of h2656_sao_template.c:81/:131 override sample (x, y); `v` receives the value
template <int BD>
__device__ __forceinline__ bool sao_restore_px(const vvc355_sao_job &job, int x, int y, int src_px, int &v)
{
    const int w = job.w, h = job.h, eo = job.eo;
    const int off0 = job.offset_val[0];
    int x0 = 0, y0 = 0, x1 = w, y1 = h;
    bool hit = false;
    if (eo != 1) {
        if (job.borders[0]) { if (x == 0) { v = clip_px<BD>(src_px + off0); hit = true; } x0 = 1; }
        if (job.borders[2]) { if (x == w - 1) { v = clip_px<BD>(src_px + off0); hit = true; } x1--; }
    }
    if (eo != 0) {
        if (job.borders[1]) { if (y == 0 && x >= x0 && x < x1) { v = clip_px<BD>(src_px + off0); hit = true; } if (job.restore) y0 = 1; }
        if (job.borders[3]) { if (y == h - 1 && x >= x0 && x < x1) { v = clip_px<BD>(src_px + off0); hit = true; } y1--; }
    }
    if (job.restore) {
        const int keep_ul = !job.diag_edge[0] && eo == 2 && !job.borders[0] && !job.borders[1];
        const int keep_ur = !job.diag_edge[1] && eo == 3 && !job.borders[1] && !job.borders[2];
        const int keep_lr = !job.diag_edge[2] && eo == 2 && !job.borders[2] && !job.borders[3];
        const int keep_ll = !job.diag_edge[3] && eo == 3 && !job.borders[0] && !job.borders[3];
        bool r = false;
        r |= job.vert_edge[0] && eo != 1 && x == 0 && y >= y0 + keep_ul && y < y1 - keep_ll;
        r |= job.vert_edge[1] && eo != 1 && x == x1 - 1 && y >= y0 + keep_ur && y < y1 - keep_lr;
        r |= job.horiz_edge[0] && eo != 0 && y == 0 && x >= x0 + keep_ul && x < x1 - keep_ur;
        r |= job.horiz_edge[1] && eo != 0 && y == y1 - 1 && x >= x0 + keep_ll && x < x1 - keep_lr;
        r |= job.diag_edge[0] && eo == 2 && x == 0 && y == 0;
        r |= job.diag_edge[1] && eo == 3 && x == x1 - 1 && y == 0;
        r |= job.diag_edge[2] && eo == 2 && x == x1 - 1 && y == y1 - 1;
        r |= job.diag_edge[3] && eo == 3 && x == 0 && y == y1 - 1;
        if (r) { v = src_px; hit = true; }
    }
    return hit;
}

// type: 1 band (:24), 2 edge (:50), 3 edge + restore fused (batched stage), 4 restore only (edge_restore slot)
template <int BD>
__global__ __launch_bounds__(256) void sao_kernel(const vvc355_sao_job *__restrict__ jobs)
{
    const vvc355_sao_job job = load_uniform(jobs + (blockIdx.y));
    const int w = job.w, h = job.h, type = job.type;
    const uint8_t *src = (const uint8_t *)job.src;
    const ptrdiff_t ss = job.src_stride / (ptrdiff_t)sizeof(typename Px<BD>::type);
    int band[4];
#pragma unroll
    for (int k = 0; k < 4; k++) band[k] = (k + job.band_position) & 31;
    const int eo = job.eo & 3;
    const ptrdiff_t oa = kSaoNb[eo][0] + kSaoNb[eo][1] * ss, ob = kSaoNb[eo][2] + kSaoNb[eo][3] * ss;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < w * h; i += gridDim.x * blockDim.x) {
        const int y = i / w, x = i - y * w;
        const ptrdiff_t o = (ptrdiff_t)y * ss + x;
        const int s = ld_px<BD>(src, o);
        uint8_t *drow = (uint8_t *)job.dst + (ptrdiff_t)y * job.dst_stride;
        int v = s;
        if (type == 1) {
            const int b = (s >> (BD - 5)) & 31;
            int off = 0;
            // later table writes win when band positions wrap onto each other (:38-39)
#pragma unroll
            for (int k = 0; k < 4; k++) if (b == band[k]) off = job.offset_val[k + 1];
            st_px<BD>(drow, x, clip_px<BD>(s + off));
            continue;
        }
        if (type == 4) {
            if (sao_restore_px<BD>(job, x, y, s, v))
                st_px<BD>(drow, x, v);
            continue;
        }
        if (type == 3 && sao_restore_px<BD>(job, x, y, s, v)) {
            st_px<BD>(drow, x, v);
            continue;
        }
        const int k = 2 + sign_of(s - ld_px<BD>(src, o + oa)) + sign_of(s - ld_px<BD>(src, o + ob));
        st_px<BD>(drow, x, clip_px<BD>(s + job.offset_val[kSaoCat[k]]));
    }
}

// ---- packed 16-bit helpers of the vectorised SAO: a register holds two samples
typedef short pk16 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pk16 pk(uint32_t v) { return __builtin_bit_cast(pk16, v); }
__device__ __forceinline__ uint32_t un(pk16 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ pk16 pk_splat(int v) { return pk16{ (short)v, (short)v }; }
// the compiler lowers min / max against small constants to compare + select per half: pin the packed instructions
__device__ __forceinline__ pk16 pk_min(pk16 a, pk16 b)
{
    pk16 r;
    asm("v_pk_min_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ pk16 pk_max(pk16 a, pk16 b)
{
    pk16 r;
    asm("v_pk_max_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <int BD> __device__ __forceinline__ void load8_pk(const typename Px<BD>::type *p, uint32_t (&d)[4])
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
template <int BD> __device__ __forceinline__ void store8_pk(typename Px<BD>::type *p, const uint32_t (&d)[4])
{
    if (BD > 8)
        gst<uint4>(p, make_uint4(d[0], d[1], d[2], d[3]));
    else
        gst<uint2>(p, make_uint2(__builtin_amdgcn_perm(d[1], d[0], 0x06040200u), __builtin_amdgcn_perm(d[3], d[2], 0x06040200u)));
}

// the aligned vector t displaced by DX samples; e = the one sample beyond its end on that side
template <int DX> __device__ __forceinline__ void shift8_pk(const uint32_t (&t)[4], uint32_t e, uint32_t (&o)[4])
{
    if (DX == 0) {
#pragma unroll
        for (int i = 0; i < 4; i++) o[i] = t[i];
    } else if (DX < 0) {
        o[0] = (t[0] << 16) | e;
#pragma unroll
        for (int i = 1; i < 4; i++) o[i] = __builtin_amdgcn_alignbit(t[i], t[i - 1], 16);
    } else {
#pragma unroll
        for (int i = 0; i < 3; i++) o[i] = __builtin_amdgcn_alignbit(t[i + 1], t[i], 16);
        o[3] = (t[3] >> 16) | (e << 16);
    }
}

struct SaoLut { uint32_t lo_a, lo_b, hi_a, hi_b; };      // bytes of the five offsets by category index 0..3 | 4

// edge offset of 8 samples (h2656_sao_template.c:50-79) from registers: c = the samples, ta / tb = the aligned 8-sample vectors
// of the rows that hold neighbours a / b, displaced by DXA / DXB columns (ea / eb = the sample just beyond the vector's end)
template <int BD, int DXA, int DXB>
__device__ __forceinline__ void sao_edge8(const uint32_t (&c)[4], const uint32_t (&ta)[4], uint32_t ea, const uint32_t (&tb)[4],
                                          uint32_t eb, const SaoLut &lut, uint32_t (&out)[4])
{
    uint32_t a[4], b[4];
    shift8_pk<DXA>(ta, ea, a);
    shift8_pk<DXB>(tb, eb, b);
    const pk16 one = pk_splat(1), mone = pk_splat(-1), two = pk_splat(2), zero = pk_splat(0), top = pk_splat((1 << BD) - 1);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const pk16 s1 = pk_max(pk_min(pk(c[i]) - pk(a[i]), one), mone);
        const pk16 s2 = pk_max(pk_min(pk(c[i]) - pk(b[i]), one), mone);
        const uint32_t idx = un(s1 + s2 + two);                              // 0..4 in the low byte of each half
        const uint32_t off = __builtin_amdgcn_perm(lut.lo_b, lut.lo_a, idx | 0x0c000c00u)
                           | __builtin_amdgcn_perm(lut.hi_b, lut.hi_a, (idx << 8) | 0x000c000cu);
        const pk16 v = __builtin_elementwise_add_sat(pk(c[i]), pk(off));     // saturating: exact for any 16-bit offset
        out[i] = un(pk_min(pk_max(v, zero), top));
    }
}

// Vectorised form for the batched stage (types 1 and 3).  A lane owns 8 consecutive samples of FOUR consecutive rows: the six
// row vectors it needs (one 16-byte load each at 10-bit) and the samples beside them are requested up front, so a wave has ~100 bytes per lane in flight and lives for one memory round trip per 2048
// samples.  A workgroup covers 128 columns x 64 rows (narrower rectangles: fewer lanes across, more rows).  Only rows / lanes that
// touch the rectangle's outer ring where a border / restore flag is set take the per-sample path of sao_restore_px.
template <int BD>
__device__ __forceinline__ void sao_vec_body(const vvc355_sao_job &job, int bx, int tid)
{
    using px_t = typename Px<BD>::type;
    const int w = job.w, h = job.h, type = job.type;
    const int lxl = w > 64 ? 4 : w > 32 ? 3 : w > 16 ? 2 : w > 8 ? 1 : 0;        // log2 of the lanes across
    const int x0 = (tid & ((1 << lxl) - 1)) * 8;
    const int y0 = (bx * (256 >> lxl) + (tid >> lxl)) * 4;
    if (y0 >= h || x0 >= w)
        return;
    const px_t *src = (const px_t *)job.src;
    const ptrdiff_t ss = job.src_stride / (ptrdiff_t)sizeof(px_t);
    const int eo = job.eo & 3;
    const int dxa = kSaoNb[eo][0], dya = kSaoNb[eo][1], dxb = kSaoNb[eo][2], dyb = kSaoNb[eo][3];
    // the per-sample rules only matter where a border / restore flag of that side is set (or the vector is partial)
    const bool f_l = job.borders[0] | job.vert_edge[0] | job.diag_edge[0] | job.diag_edge[3];
    const bool f_r = job.borders[2] | job.vert_edge[1] | job.diag_edge[1] | job.diag_edge[2];
    const bool f_t = job.borders[1] | job.horiz_edge[0] | job.diag_edge[0] | job.diag_edge[1];
    const bool f_b = job.borders[3] | job.horiz_edge[1] | job.diag_edge[2] | job.diag_edge[3];
    const bool xring = (x0 == 0 && f_l) || (x0 + 8 >= w && (f_r || x0 + 8 > w));

    auto per_sample_row = [&](int y) {
        // picture borders, unfilterable slice / tile edges, partial vectors
        const px_t *srow = src + (ptrdiff_t)y * ss;
        px_t *drow = (px_t *)((uint8_t *)job.dst + (ptrdiff_t)y * job.dst_stride);
        for (int x = x0; x < min(x0 + 8, w); x++) {
            const int s = srow[x];
            int v;
            if (!sao_restore_px<BD>(job, x, y, s, v)) {
                const int k = 2 + sign_of(s - (int)srow[x + dxa + dya * ss]) + sign_of(s - (int)srow[x + dxb + dyb * ss]);
                v = clip_px<BD>(s + job.offset_val[kSaoCat[k]]);
            }
            drow[x] = (px_t)v;
        }
    };
    // rows y0 - 1 .. y0 + 4, clamped to the rows that may be read (a flagged top / bottom ring row never reads beyond itself)
    uint32_t v[6][4];
    const int ylo = f_t ? 0 : -1, yhi = f_b ? h - 1 : h;
    const bool vert = type == 3 && eo != 0;
#pragma unroll
    for (int k = 0; k < 6; k++) {
        if ((k == 0 || k == 5) && !vert) {
            v[k][0] = v[k][1] = v[k][2] = v[k][3] = 0;
            continue;
        }
        const int yk = min(max(y0 - 1 + k, ylo), yhi);
        load8_pk<BD>((const px_t *)((const uint8_t *)src + row_off(yk, job.src_stride)) + x0, v[k]);
    }
    uint32_t out[4][4];
    if (type == 1) {
        // band filter (h2656_sao_template.c:24) two samples per operation: band = sample >> (bd - 5); (band - band_position) & 31
        // below 4 selects offset_val[1..4], anything else adds nothing — the same five-entry byte look-up as the edge categories
        uint32_t lo_a = 0, hi_a = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t o = (uint16_t)job.offset_val[k + 1];
            lo_a |= (o & 0xff) << (8 * k);
            hi_a |= (o >> 8) << (8 * k);
        }
        const pk16 pos = pk_splat(job.band_position), four = pk_splat(4), zero = pk_splat(0), top = pk_splat((1 << BD) - 1);
        const uint32_t sh = (BD - 5) * 0x10001u;
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                uint32_t band;
                asm("v_pk_lshrrev_b16 %0, %1, %2" : "=v"(band) : "v"(sh), "v"(v[r + 1][i]));
                const uint32_t idx = un(pk_min(pk(un(pk(band) - pos) & 0x001f001fu), four));
                const uint32_t off = __builtin_amdgcn_perm(0u, lo_a, idx | 0x0c000c00u) | __builtin_amdgcn_perm(0u, hi_a, (idx << 8) | 0x000c000cu);
                const pk16 t = __builtin_elementwise_add_sat(pk(v[r + 1][i]), pk(off));
                out[r][i] = un(pk_min(pk_max(t, zero), top));
            }
    } else {
        // category -> offset as two byte look-ups (v_perm_b32): low bytes and high bytes of offset_val[kSaoCat[0..4]]
        uint32_t lo_a = 0, hi_a = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t o = (uint16_t)job.offset_val[kSaoCat[k]];
            lo_a |= (o & 0xff) << (8 * k);
            hi_a |= (o >> 8) << (8 * k);
        }
        const uint32_t o4 = (uint16_t)job.offset_val[kSaoCat[4]];
        const SaoLut lut = { lo_a, o4 & 0xff, hi_a, o4 >> 8 };
        // the sample left / right of each row vector.  Loaded by every lane, unconditionally and right behind the row vectors,
        // so that all of a wave's loads are in flight together (taking them from the neighbouring lanes' registers needs
        // separate loads at the rectangle's first / last vector, and those cost the whole wave another memory round trip).
        // On a flagged left / right ring the address is clamped into the rectangle; the value is not used there.
        uint32_t eL[6], eR[6];
        const int xl = (x0 == 0 && f_l) ? 0 : x0 - 1, xr = (x0 + 8 >= w && (f_r || x0 + 8 > w)) ? w - 1 : x0 + 8;
#pragma unroll
        for (int k = 0; k < 6; k++) {
            eL[k] = eR[k] = 0;
            if (eo == 1 || ((k == 0 || k == 5) && eo == 0))
                continue;
            const int yk = min(max(y0 - 1 + k, ylo), yhi);
            const px_t *rowk = (const px_t *)((const uint8_t *)src + row_off(yk, job.src_stride));
            eL[k] = (uint32_t)gld<px_t>(rowk + xl);
            eR[k] = (uint32_t)gld<px_t>(rowk + xr);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int k = r + 1;
            switch (eo) {
            case 0:  sao_edge8<BD, -1, 1>(v[k], v[k], eL[k], v[k], eR[k], lut, out[r]); break;
            case 1:  sao_edge8<BD, 0, 0>(v[k], v[k - 1], 0, v[k + 1], 0, lut, out[r]); break;
            case 2:  sao_edge8<BD, -1, 1>(v[k], v[k - 1], eL[k - 1], v[k + 1], eR[k + 1], lut, out[r]); break;
            default: sao_edge8<BD, 1, -1>(v[k], v[k - 1], eR[k - 1], v[k + 1], eL[k + 1], lut, out[r]); break;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int y = y0 + r;
        if (y >= h)
            break;
        if (type == 3 && (xring || (y == 0 && f_t) || (y == h - 1 && f_b))) {
            per_sample_row(y);
            continue;
        }
        px_t *drow = (px_t *)((uint8_t *)job.dst + row_off(y, job.dst_stride));
        if (x0 + 8 <= w) {
            store8_pk<BD>(drow + x0, out[r]);
        } else {
            for (int j = 0; j < w - x0; j++) drow[x0 + j] = (px_t)((out[r][j >> 1] >> ((j & 1) * 16)) & 0xffff);
        }
    }
}

template <int BD>
__global__ __launch_bounds__(256) void sao_vec_kernel(const vvc355_sao_job *__restrict__ jobs)
{
    const vvc355_sao_job job = load_uniform(jobs + blockIdx.y);
    sao_vec_body<BD>(job, blockIdx.x, threadIdx.x);
}

// a[c] for c in 0..2 as selects between the three values: indexing a register copy of a descriptor with a run-time index would
// put the whole descriptor into scratch memory
template <typename T> __device__ __forceinline__ T sel3(int c, const T (&a)[3]) { return c == 0 ? a[0] : c == 1 ? a[1] : a[2]; }

// SAO stage driver (ff_vvc_sao_filter, vvc_filter.c:154-300): blockIdx.y = CTB, blockIdx.x = tile of the CTB: tiles_l luma tiles
// (a tile = what one workgroup of the vector body covers), then the chroma tiles — or, when a chroma CTB needs only half a
// workgroup (4:2:0), ONE tile whose first two waves take Cb and last two Cr.  (Workgroups and waves that find nothing to do are
// not free: launching them costs about 3 ns per workgroup, which was a quarter of this kernel's time.)  The job the vector body
// works on is derived here, on the scalar unit, from the per-CTB tables: picture-border flags (:172-175), unfilterable slice /
// tile edges (:177-215), type / band position / edge class / offsets of the component.  CTBs without SAO are copied.
template <int BD>
__global__ __launch_bounds__(256) void sao_frame_kernel(const vvc355_sao_frame *__restrict__ fp, int xg, int tiles_l, int tiles_c, int packed)
{
    using px_t = typename Px<BD>::type;
    const vvc355_sao_frame F = load_uniform(fp);
    // XCD-grouped numbering (xg = the workgroups of four CTBs): the lines holding the samples beside a CTB are shared with the
    // neighbouring CTB, which then sits in the same L2.  (Measured: L2 fetch traffic 177 -> 103 MB per 8K frame, 0.087 -> 0.082 ms;
    // one contiguous eighth of the picture per XCD was slower, the XCDs then stream from eight distant DRAM regions.)
    const int lin = xcd_grouped(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y, xg);
    const int ctb = lin / (int)gridDim.x, t = lin - ctb * (int)gridDim.x;
    int c, bx, tid = threadIdx.x, group = 256, ntiles;
    if (t < tiles_l)  { c = 0; bx = t; ntiles = tiles_l; }
    else if (packed)  { c = 1 + __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 7); bx = 0; tid = threadIdx.x & 127; group = 128; ntiles = 1; }
    else              { const int t2 = t - tiles_l; c = 1 + t2 / tiles_c; bx = t2 - (c - 1) * tiles_c; ntiles = tiles_c; }
    const int yc = ctb / F.ctb_width, xc = ctb - yc * F.ctb_width;
    const vvc355_sao_ctb P = load_uniform((const vvc355_sao_ctb *)F.sao + ctb);
    const int hs = c ? F.hs : 0, vs = c ? F.vs : 0;
    const int pw = F.width >> hs, ph = F.height >> vs;
    const int x0 = (xc << F.ctb_log2) >> hs, y0 = (yc << F.ctb_log2) >> vs;
    vvc355_sao_job job = {};
    job.w = (int16_t)min((1 << F.ctb_log2) >> hs, pw - x0);
    job.h = (int16_t)min((1 << F.ctb_log2) >> vs, ph - y0);
    job.dst_stride = sel3(c, F.dst_stride); job.src_stride = sel3(c, F.src_stride);
    job.dst = sel3(c, F.dst) + (uint64_t)((ptrdiff_t)y0 * job.dst_stride + x0 * (int)sizeof(px_t));
    job.src = sel3(c, F.src) + (uint64_t)((ptrdiff_t)y0 * job.src_stride + x0 * (int)sizeof(px_t));
    const int type_idx = sel3(c, P.type_idx);
    if (type_idx == 0) {
        // SAO not applied: the samples pass through (16-byte vectors where the row allows, else sample by sample)
        const int w = job.w, h = job.h, wv = w >> 3;
        for (int i = bx * group + tid; i < (wv + 1) * h; i += ntiles * group) {
            const int y = i / (wv + 1), xv = i - y * (wv + 1);
            const px_t *sp = (const px_t *)((const uint8_t *)job.src + row_off(y, job.src_stride)) + xv * 8;
            px_t *dp = (px_t *)((uint8_t *)job.dst + row_off(y, job.dst_stride)) + xv * 8;
            if (xv < wv) {
                uint32_t v[4];
                load8_pk<BD>(sp, v);
                store8_pk<BD>(dp, v);
            } else {
                for (int x = 0; x < (w & 7); x++) dp[x] = sp[x];
            }
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < 5; k++) job.offset_val[k] = c == 0 ? P.offset_val[0][k] : c == 1 ? P.offset_val[1][k] : P.offset_val[2][k];
    job.type = type_idx == 1 ? 1 : 3;
    job.eo = sel3(c, P.eo_class); job.band_position = sel3(c, P.band_position);
    const int eL = xc == 0, eT = yc == 0, eR = xc == F.ctb_width - 1, eB = yc == F.ctb_height - 1;
    job.borders[0] = eL; job.borders[1] = eT; job.borders[2] = eR; job.borders[3] = eB;
    const int restore = F.no_tile_filter || !F.lfase;
    job.restore = restore;
    if (restore) {
        const int16_t *slice = (const int16_t *)F.slice_idx;
        const int16_t *col_bd = (const int16_t *)F.ctb_to_col_bd, *row_bd = (const int16_t *)F.ctb_to_row_bd;
        const int cw = F.ctb_width, nl = !F.lfase;
        const int me = slice[ctb];
        int lt = 0, rt = 0, ut = 0, bt = 0;
        if (!eL) { lt = F.no_tile_filter && col_bd[xc] == xc; job.vert_edge[0] = (nl && me != slice[ctb - 1]) || lt; }
        if (!eR) { rt = F.no_tile_filter && col_bd[xc] != col_bd[xc + 1]; job.vert_edge[1] = (nl && me != slice[ctb + 1]) || rt; }
        if (!eT) { ut = F.no_tile_filter && row_bd[yc] == yc; job.horiz_edge[0] = (nl && me != slice[ctb - cw]) || ut; }
        if (!eB) { bt = F.no_tile_filter && row_bd[yc] != row_bd[yc + 1]; job.horiz_edge[1] = (nl && me != slice[ctb + cw]) || bt; }
        if (!eL && !eT) job.diag_edge[0] = (nl && me != slice[ctb - cw - 1]) || lt || ut;
        if (!eT && !eR) job.diag_edge[1] = (nl && me != slice[ctb - cw + 1]) || rt || ut;
        if (!eR && !eB) job.diag_edge[2] = (nl && me != slice[ctb + cw + 1]) || rt || bt;
        if (!eL && !eB) job.diag_edge[3] = (nl && me != slice[ctb + cw - 1]) || lt || bt;
    }
    sao_vec_body<BD>(job, bx, tid);
}

// ------------------------------------------------------------------------------------------------ deblock

// One 4-line (or 2-line) segment held in registers: p[i] at pix - (i+1)*xs, q[i] at pix + i*xs, lines ys apart (in pixels).
// Everything a decision can read is requested up front as 4-sample vectors (one memory round trip instead of one per decision
// stage); only the samples a filter changes are written back, one by one, because the neighbouring edges' lanes own the rest.
template <int BD> struct Dbk {
    using px_t = typename Px<BD>::type;
    uint8_t *pix;
    int xs, ys;                                  // in pixels; int: offsets stay 32-bit (full-rate 24-bit multiplies)
    int P[4][8], Q[4][8];
    __device__ __forceinline__ int p(int l, int i) const { return P[l][i]; }
    __device__ __forceinline__ int q(int l, int i) const { return Q[l][i]; }
    // i in {3, 5, 7} / l in {1, 3}: selects between VALUES (the empty asm keeps the compiler from folding them back into one
    // load from a selected address, which would force the whole struct into scratch memory)
    static __device__ __forceinline__ int opaque(int v) { asm("" : "+v"(v)); return v; }
    __device__ __forceinline__ int p_at(int l, int i) const { const int a = opaque(P[l][7]), b = opaque(P[l][5]), c = opaque(P[l][3]); return i == 7 ? a : i == 5 ? b : c; }
    __device__ __forceinline__ int q_at(int l, int i) const { const int a = opaque(Q[l][7]), b = opaque(Q[l][5]), c = opaque(Q[l][3]); return i == 7 ? a : i == 5 ? b : c; }
    __device__ __forceinline__ int pn(int l2, int i) const { const int a = opaque(P[1][i]), b = opaque(P[3][i]); return l2 == 1 ? a : b; }
    __device__ __forceinline__ int qn(int l2, int i) const { const int a = opaque(Q[1][i]), b = opaque(Q[3][i]); return l2 == 1 ? a : b; }
    __device__ __forceinline__ void sp(int l, int i, int v) const { st_px<BD>(pix, l * ys - (i + 1) * xs, v); }
    __device__ __forceinline__ void sq(int l, int i, int v) const { st_px<BD>(pix, l * ys + i * xs, v); }

    // n (2 or 4) consecutive samples at pixel offset o
    __device__ __forceinline__ void vec(int o, int n, int (&e)[4]) const
    {
        const px_t *a = (const px_t *)pix + o;
        if (BD > 8) {
            uint2 v = make_uint2(0, 0);
            if (n == 4) v = gld<uint2>(a); else v.x = gld<uint32_t>(a);
            e[0] = v.x & 0xffff; e[1] = v.x >> 16; e[2] = v.y & 0xffff; e[3] = v.y >> 16;
        } else {
            const uint32_t v = n == 4 ? gld<uint32_t>(a) : (uint32_t)gld<uint16_t>(a);
            e[0] = v & 0xff; e[1] = (v >> 8) & 0xff; e[2] = (v >> 16) & 0xff; e[3] = v >> 24;
        }
    }
    // samples [4 * half, 4 * half + 4) of both sides (which = 1: p, 2: q, 3: both) for `lines` lines
    __device__ __forceinline__ void load(int lines, int half, int which)
    {
        const int b = 4 * half;
        if (xs == 1) {
            // p3..p0 | q0..q3 of a line are consecutive in memory
#pragma unroll
            for (int l = 0; l < 4; l++) {
                int e[4];
                if (l < lines && (which & 1)) {
                    vec(l * ys - b - 4, 4, e);
#pragma unroll
                    for (int k = 0; k < 4; k++) P[l][b + 3 - k] = e[k];
                }
                if (l < lines && (which & 2)) {
                    vec(l * ys + b, 4, e);
#pragma unroll
                    for (int k = 0; k < 4; k++) Q[l][b + k] = e[k];
                }
            }
        } else {
            // the lines are consecutive in memory, p[i] / q[i] one row each
#pragma unroll
            for (int i = 0; i < 4; i++) {
                int e[4];
                if (which & 1) {
                    vec(-(b + i + 1) * xs, lines, e);
#pragma unroll
                    for (int l = 0; l < 4; l++) P[l][b + i] = e[l];
                }
                if (which & 2) {
                    vec((b + i) * xs, lines, e);
#pragma unroll
                    for (int l = 0; l < 4; l++) Q[l][b + i] = e[l];
                }
            }
        }
    }
};
__device__ __forceinline__ int d2(int a, int b, int c) { return abs(a - 2 * b + c); }

// long-filter interpolation weights {53,32,11} {58,45,32,19,6} {59,50,41,32,23,14,5} and tc multipliers {6,4,2} {6,5,4,3,2} {6,5,4,3,2,1,1}
// (vvc_filter_template.c:466-530) as arithmetic on the tap index
__device__ __forceinline__ int long_w(int n, int i) { return n == 3 ? 53 - 21 * i : n == 5 ? 58 - 13 * i : 59 - 9 * i; }
__device__ __forceinline__ int long_t(int n, int i) { return n == 3 ? 6 - 2 * i : n == 5 ? 6 - i : max(6 - i, 1); }

template <int BD>
__device__ __forceinline__ void dbk_luma_large(const Dbk<BD> &d, int tc, int no_p, int no_q, int len_p, int len_q)
{
#pragma unroll
    for (int l = 0; l < 4; l++) {
        int p[8], q[8], m;
#pragma unroll
        for (int i = 0; i < 8; i++) { p[i] = d.p(l, i); q[i] = d.q(l, i); }
        if (len_p == 5 && len_q == 5)
            m = (p[4] + p[3] + 2 * (p[2] + p[1] + p[0] + q[0] + q[1] + q[2]) + q[3] + q[4] + 8) >> 4;
        else if (len_p == len_q)
            m = (p[6] + p[5] + p[4] + p[3] + p[2] + p[1] + 2 * (p[0] + q[0]) + q[1] + q[2] + q[3] + q[4] + q[5] + q[6] + 8) >> 4;
        else if (len_p + len_q == 12)
            m = (p[5] + p[4] + p[3] + p[2] + 2 * (p[1] + p[0] + q[0] + q[1]) + q[2] + q[3] + q[4] + q[5] + 8) >> 4;
        else if (len_p + len_q == 8)
            m = (p[3] + p[2] + p[1] + p[0] + q[0] + q[1] + q[2] + q[3] + 4) >> 3;
        else if (len_q == 7)
            m = (2 * (p[2] + p[1] + p[0] + q[0]) + p[0] + p[1] + q[1] + q[2] + q[3] + q[4] + q[5] + q[6] + 8) >> 4;
        else
            m = (p[6] + p[5] + p[4] + p[3] + p[2] + p[1] + 2 * (q[2] + q[1] + q[0] + p[0]) + q[0] + q[1] + 8) >> 4;
        if (!no_p) {
            const int n = len_p == 3 ? 3 : len_p == 5 ? 5 : 7;
            int ref = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) if (i == len_p || i == len_p - 1) ref += p[i];
            ref = (ref + 1) >> 1;
#pragma unroll
            for (int i = 0; i < 7; i++)
                if (i < n) {
                    const int lim = (tc * long_t(n, i)) >> 1, wt = long_w(n, i);
                    d.sp(l, i, p[i] + clip3(((m * wt + ref * (64 - wt) + 32) >> 6) - p[i], -lim, lim));
                }
        }
        if (!no_q) {
            const int n = len_q == 3 ? 3 : len_q == 5 ? 5 : 7;
            int ref = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) if (i == len_q || i == len_q - 1) ref += q[i];
            ref = (ref + 1) >> 1;
#pragma unroll
            for (int i = 0; i < 7; i++)
                if (i < n) {
                    const int lim = (tc * long_t(n, i)) >> 1, wt = long_w(n, i);
                    d.sq(l, i, q[i] + clip3(((m * wt + ref * (64 - wt) + 32) >> 6) - q[i], -lim, lim));
                }
        }
    }
}

template <int BD>
__device__ __forceinline__ void dbk_luma_segment(const Dbk<BD> &d, int tc_in, int beta_in, int no_p, int no_q, int len_p, int len_q, int hor_ctu_edge)
{
    const int tc = BD < 10 ? (tc_in + (1 << (9 - BD))) >> (10 - BD) : tc_in << (BD - 10);
    if (!tc)
        return;
    const int dp0 = d2(d.p(0, 2), d.p(0, 1), d.p(0, 0)), dq0 = d2(d.q(0, 2), d.q(0, 1), d.q(0, 0));
    const int dp3 = d2(d.p(3, 2), d.p(3, 1), d.p(3, 0)), dq3 = d2(d.q(3, 2), d.q(3, 1), d.q(3, 0));
    const int d0 = dp0 + dq0, d3 = dp3 + dq3;
    const int tc25 = (tc * 5 + 1) >> 1;
    const int large_p = len_p > 3 && !hor_ctu_edge, large_q = len_q > 3;
    const int beta = beta_in << (BD - 8);

    if (large_p || large_q) {
        const int dp0l = large_p ? (dp0 + d2(d.p(0, 5), d.p(0, 4), d.p(0, 3)) + 1) >> 1 : dp0;
        const int dq0l = large_q ? (dq0 + d2(d.q(0, 5), d.q(0, 4), d.q(0, 3)) + 1) >> 1 : dq0;
        const int dp3l = large_p ? (dp3 + d2(d.p(3, 5), d.p(3, 4), d.p(3, 3)) + 1) >> 1 : dp3;
        const int dq3l = large_q ? (dq3 + d2(d.q(3, 5), d.q(3, 4), d.q(3, 3)) + 1) >> 1 : dq3;
        const int d0l = dp0l + dq0l, d3l = dp3l + dq3l;
        const int beta53 = (beta * 3) >> 5, beta_4 = beta >> 4;
        len_p = large_p ? len_p : 3;
        len_q = large_q ? len_q : 3;
        if (d0l + d3l < beta) {
            const int sp0l = abs(d.p(0, 3) - d.p(0, 0)) + (len_p == 7 ? abs(d.p(0, 7) - d.p(0, 6) - d.p(0, 5) + d.p(0, 4)) : 0);
            const int sq0l = abs(d.q(0, 0) - d.q(0, 3)) + (len_q == 7 ? abs(d.q(0, 4) - d.q(0, 5) - d.q(0, 6) + d.q(0, 7)) : 0);
            const int sp3l = abs(d.p(3, 3) - d.p(3, 0)) + (len_p == 7 ? abs(d.p(3, 7) - d.p(3, 6) - d.p(3, 5) + d.p(3, 4)) : 0);
            const int sq3l = abs(d.q(3, 0) - d.q(3, 3)) + (len_q == 7 ? abs(d.q(3, 4) - d.q(3, 5) - d.q(3, 6) + d.q(3, 7)) : 0);
            const int sp0 = large_p ? (sp0l + abs(d.p(0, 3) - d.p_at(0, len_p)) + 1) >> 1 : sp0l;
            const int sp3 = large_p ? (sp3l + abs(d.p(3, 3) - d.p_at(3, len_p)) + 1) >> 1 : sp3l;
            const int sq0 = large_q ? (sq0l + abs(d.q(0, 3) - d.q_at(0, len_q)) + 1) >> 1 : sq0l;
            const int sq3 = large_q ? (sq3l + abs(d.q(3, 3) - d.q_at(3, len_q)) + 1) >> 1 : sq3l;
            if (sp0 + sq0 < beta53 && abs(d.p(0, 0) - d.q(0, 0)) < tc25 &&
                sp3 + sq3 < beta53 && abs(d.p(3, 0) - d.q(3, 0)) < tc25 &&
                (d0l << 1) < beta_4 && (d3l << 1) < beta_4) {
                dbk_luma_large<BD>(d, tc, no_p, no_q, len_p, len_q);
                return;
            }
        }
    }
    if (d0 + d3 >= beta)
        return;
    const int beta_3 = beta >> 3, beta_2 = beta >> 2;
    if (len_p > 2 && len_q > 2 &&
        abs(d.p(0, 3) - d.p(0, 0)) + abs(d.q(0, 3) - d.q(0, 0)) < beta_3 && abs(d.p(0, 0) - d.q(0, 0)) < tc25 &&
        abs(d.p(3, 3) - d.p(3, 0)) + abs(d.q(3, 3) - d.q(3, 0)) < beta_3 && abs(d.p(3, 0) - d.q(3, 0)) < tc25 &&
        (d0 << 1) < beta_2 && (d3 << 1) < beta_2) {
        // strong filter, h2656_deblock_template.c:25
        const int tc2 = tc << 1, tc3 = tc * 3;
#pragma unroll
        for (int l = 0; l < 4; l++) {
            const int p3 = d.p(l, 3), p2 = d.p(l, 2), p1 = d.p(l, 1), p0 = d.p(l, 0);
            const int q0 = d.q(l, 0), q1 = d.q(l, 1), q2 = d.q(l, 2), q3 = d.q(l, 3);
            if (!no_p) {
                d.sp(l, 0, p0 + clip3(((p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3) - p0, -tc3, tc3));
                d.sp(l, 1, p1 + clip3(((p2 + p1 + p0 + q0 + 2) >> 2) - p1, -tc2, tc2));
                d.sp(l, 2, p2 + clip3(((2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3) - p2, -tc, tc));
            }
            if (!no_q) {
                d.sq(l, 0, q0 + clip3(((p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3) - q0, -tc3, tc3));
                d.sq(l, 1, q1 + clip3(((p0 + q0 + q1 + q2 + 2) >> 2) - q1, -tc2, tc2));
                d.sq(l, 2, q2 + clip3(((2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3) - q2, -tc, tc));
            }
        }
    } else {
        // weak filter, h2656_deblock_template.c:52
        int nd_p = 1, nd_q = 1;
        if (len_p > 1 && len_q > 1) {
            const int side = (beta + (beta >> 1)) >> 3;
            if (dp0 + dp3 < side) nd_p = 2;
            if (dq0 + dq3 < side) nd_q = 2;
        }
        const int tc_2 = tc >> 1;
#pragma unroll
        for (int l = 0; l < 4; l++) {
            const int p2 = d.p(l, 2), p1 = d.p(l, 1), p0 = d.p(l, 0);
            const int q0 = d.q(l, 0), q1 = d.q(l, 1), q2 = d.q(l, 2);
            int delta = (9 * (q0 - p0) - 3 * (q1 - p1) + 8) >> 4;
            if (abs(delta) >= 10 * tc)
                continue;
            delta = clip3(delta, -tc, tc);
            if (!no_p) d.sp(l, 0, clip_px<BD>(p0 + delta));
            if (!no_q) d.sq(l, 0, clip_px<BD>(q0 - delta));
            if (!no_p && nd_p > 1) d.sp(l, 1, clip_px<BD>(p1 + clip3((((p2 + p0 + 1) >> 1) - p1 + delta) >> 1, -tc_2, tc_2)));
            if (!no_q && nd_q > 1) d.sq(l, 1, clip_px<BD>(q1 + clip3((((q2 + q0 + 1) >> 1) - q1 - delta) >> 1, -tc_2, tc_2)));
        }
    }
}

template <int BD>
__device__ __forceinline__ void dbk_chroma_segment(const Dbk<BD> &d, int lines, int tc_in, int beta_in, int no_p, int no_q, int len_p, int len_q)
{
    const int tc = BD < 10 ? (tc_in + (1 << (9 - BD))) >> (10 - BD) : tc_in << (BD - 10);
    if (!tc || !len_p || !len_q)
        return;
    const int l2 = lines == 2 ? 1 : 3;
    const int beta = beta_in << (BD - 8), beta_3 = beta >> 3, beta_2 = beta >> 2, tc25 = (tc * 5 + 1) >> 1;
    if (len_q == 3) {
        const bool one = len_p == 1;
        const int p0 = d.p(0, 0), p1 = d.p(0, 1), p2 = one ? p1 : d.p(0, 2), p3 = one ? p1 : d.p(0, 3);
        const int p0n = d.pn(l2, 0), p1n = d.pn(l2, 1), p2n = one ? p1n : d.pn(l2, 2);
        const int q0 = d.q(0, 0), q1 = d.q(0, 1), q2 = d.q(0, 2), q3 = d.q(0, 3);
        const int q0n = d.qn(l2, 0), q1n = d.qn(l2, 1), q2n = d.qn(l2, 2);
        const int dd0 = d2(p2, p1, p0) + d2(q2, q1, q0), dd1 = d2(p2n, p1n, p0n) + d2(q2n, q1n, q0n);
        bool strong = false;
        if (dd0 + dd1 < beta) {
            const int p3n = one ? p1n : d.pn(l2, 3), q3n = d.qn(l2, 3);
            const bool ok0 = (dd0 << 1) < beta_2 && abs(p3 - p0) + abs(q0 - q3) < beta_3 && abs(p0 - q0) < tc25;
            const bool ok1 = (dd1 << 1) < beta_2 && abs(p3n - p0n) + abs(q0n - q3n) < beta_3 && abs(p0n - q0n) < tc25;
            strong = ok0 && ok1;
        }
        if (!strong)
            len_p = len_q = 1;
    }
    const int kind = (len_p == 3 && len_q == 3) ? 2 : (len_q == 3) ? 1 : 0;
#pragma unroll
    for (int l = 0; l < 4; l++) {
        if (l >= lines)
            continue;
        const int p3 = d.p(l, 3), p2 = d.p(l, 2), p1 = d.p(l, 1), p0 = d.p(l, 0);
        const int q0 = d.q(l, 0), q1 = d.q(l, 1), q2 = d.q(l, 2), q3 = d.q(l, 3);
        if (kind == 2) {
            if (!no_p) {
                d.sp(l, 0, clip3((p3 + p2 + p1 + 2 * p0 + q0 + q1 + q2 + 4) >> 3, p0 - tc, p0 + tc));
                d.sp(l, 1, clip3((2 * p3 + p2 + 2 * p1 + p0 + q0 + q1 + 4) >> 3, p1 - tc, p1 + tc));
                d.sp(l, 2, clip3((3 * p3 + 2 * p2 + p1 + p0 + q0 + 4) >> 3, p2 - tc, p2 + tc));
            }
            if (!no_q) {
                d.sq(l, 0, clip3((p2 + p1 + p0 + 2 * q0 + q1 + q2 + q3 + 4) >> 3, q0 - tc, q0 + tc));
                d.sq(l, 1, clip3((p1 + p0 + q0 + 2 * q1 + q2 + 2 * q3 + 4) >> 3, q1 - tc, q1 + tc));
                d.sq(l, 2, clip3((p0 + q0 + q1 + 2 * q2 + 3 * q3 + 4) >> 3, q2 - tc, q2 + tc));
            }
        } else if (kind == 1) {
            if (!no_p)
                d.sp(l, 0, clip3((3 * p1 + 2 * p0 + q0 + q1 + q2 + 4) >> 3, p0 - tc, p0 + tc));
            if (!no_q) {
                d.sq(l, 0, clip3((2 * p1 + p0 + 2 * q0 + q1 + q2 + q3 + 4) >> 3, q0 - tc, q0 + tc));
                d.sq(l, 1, clip3((p1 + p0 + q0 + 2 * q1 + q2 + 2 * q3 + 4) >> 3, q1 - tc, q1 + tc));
                d.sq(l, 2, clip3((p0 + q0 + q1 + 2 * q2 + 3 * q3 + 4) >> 3, q2 - tc, q2 + tc));
            }
        } else {
            const int delta = clip3((((q0 - p0) * 4) + p1 - q1 + 4) >> 3, -tc, tc);
            if (!no_p) d.sp(l, 0, clip_px<BD>(p0 + delta));
            if (!no_q) d.sq(l, 0, clip_px<BD>(q0 - delta));
        }
    }
}

// ---- one segment, everything given: the two kernels below only differ in where the parameters come from
template <int BD>
__device__ __forceinline__ void deblock_luma_seg(uint8_t *pix, int xs, int ys, int tc, int beta, int no_p, int no_q, int len_p, int len_q, int flag)
{
    Dbk<BD> d;
    d.xs = xs; d.ys = ys; d.pix = pix;
    const int far = (len_p > 3 && !flag ? 1 : 0) | (len_q > 3 ? 2 : 0);
    d.load(4, 0, 3);
#pragma unroll
    for (int l = 0; l < 4; l++)
#pragma unroll
        for (int i = 4; i < 8; i++) d.P[l][i] = d.Q[l][i] = 0;
    if (far)
        d.load(4, 1, far);
    dbk_luma_segment<BD>(d, tc, beta, no_p, no_q, len_p, len_q, flag);
}
template <int BD>
__device__ __forceinline__ void deblock_chroma_seg(uint8_t *pix, int xs, int ys, int lines, int tc, int beta, int no_p, int no_q, int len_p, int len_q)
{
    Dbk<BD> d;
    d.xs = xs; d.ys = ys; d.pix = pix;
#pragma unroll
    for (int l = 0; l < 4; l++)
#pragma unroll
        for (int i = 0; i < 8; i++) d.P[l][i] = d.Q[l][i] = 0;
    d.load(lines, 0, 3);
    dbk_chroma_segment<BD>(d, lines, tc, beta, no_p, no_q, len_p, len_q);
}

// Two lanes per job; a job is one reference slot call (8 samples along the edge): a lane takes one 4-line segment of a luma or
// 4-line chroma job, or two of the four 2-line chroma segments.
template <int BD>
__global__ __launch_bounds__(256) void deblock_kernel(const vvc355_deblock_job *__restrict__ jobs, int n_jobs)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = t >> 1;
    if (j >= n_jobs)
        return;
    // fields indexed by the segment number are fetched individually (a register copy of the job would be indexed dynamically)
    const vvc355_deblock_job *jp = jobs + j;
    const uint32_t kind = gld<uint32_t>(&jp->dir);             // dir | chroma << 8 | flag << 16
    const int dir = kind & 0xff, chroma = (kind >> 8) & 0xff, flag = (kind >> 16) & 0xff;
    const int lines = chroma ? (flag ? 2 : 4) : 4;
    const int pxstride = gld<int32_t>(&jp->stride) / (int)sizeof(typename Px<BD>::type);
    uint8_t *pix0 = (uint8_t *)gld<uint64_t>(&jp->pix);
    const int xs = dir == 0 ? pxstride : 1, ys = dir == 0 ? 1 : pxstride;
    const int seg0 = t & 1;
    const int pxb = (int)sizeof(typename Px<BD>::type);
    if (!chroma) {
        deblock_luma_seg<BD>(pix0 + (ptrdiff_t)(seg0 * 4 * ys * pxb), xs, ys, gld<int32_t>(&jp->tc[seg0]), gld<int32_t>(&jp->beta[seg0]),
                             gld<uint8_t>(&jp->no_p[seg0]), gld<uint8_t>(&jp->no_q[seg0]), gld<uint8_t>(&jp->max_len_p[seg0]),
                             gld<uint8_t>(&jp->max_len_q[seg0]), flag);
        return;
    }
    for (int seg = seg0; seg < 8 / lines; seg += 2)
        deblock_chroma_seg<BD>(pix0 + (ptrdiff_t)(seg * lines * ys * pxb), xs, ys, lines, gld<int32_t>(&jp->tc[seg]), gld<int32_t>(&jp->beta[seg]),
                               gld<uint8_t>(&jp->no_p[seg]), gld<uint8_t>(&jp->no_q[seg]), gld<uint8_t>(&jp->max_len_p[seg]),
                               gld<uint8_t>(&jp->max_len_q[seg]));
}

// Table 43 (beta', tc' from Q), vvc_filter.c:38-52: generated from the reference's initialisers (tables_small.inc), the same text
// tables.cpp exports as vvc355_tab_tc_table / _beta_table for the table check
#define VVC355_TABLE(type, name, count) __device__ static const type lf_tab_##name[count]
#include "tables_small.inc"
#undef VVC355_TABLE
#define kTcTable lf_tab_tc_table
#define kBetaTable lf_tab_beta_table

// One deblocking pass of a picture straight from the decoder's side tables (ff_vvc_deblock_vertical / _horizontal,
// vvc_filter.c:864-1003, per-CTU loop flattened): two lanes per 8-sample unit of an edge, as in deblock_kernel, but each lane
// looks up its segment's boundary strength and derives QP, beta, tc and the filter lengths itself.  units[c] = first unit of
// component c, n_along[c] = units along one edge.
template <int BD>
__global__ __launch_bounds__(256) void deblock_frame_kernel(const vvc355_deblock_frame *__restrict__ fp, int u1, int u2, int n_units,
                                                            int na0, int na1, int ne0, int ne1)
{
    using px_t = typename Px<BD>::type;
    const vvc355_deblock_frame F = load_uniform(fp);
    const int vertical = F.vertical;
    // ---- phase A, every lane: which segment am I, and is there anything to filter?  Fewer than half of the 4-sample grid
    // positions carry a boundary strength, so the lanes with work are compacted to the front of the workgroup (one ballot per
    // wave, wave offsets through LDS) and whole waves leave before the expensive part instead of idling through it.
    __shared__ uint32_t slot[256];
    __shared__ int wave_cnt[4];
    uint32_t desc = 0;
    bool active = false;
    {
        const int t = blockIdx.x * blockDim.x + threadIdx.x;
        const int unit = t >> 1;
        if (unit < n_units) {
            if (unit >= u1) {
                active = true;                                   // chroma lane: decoded again from its own number below
                desc = 0x80000000u | threadIdx.x;
            } else {
                const int n_edges = ne0, n_along = na0;
                int ke, ku;
                if (vertical) { ku = unit / n_edges; ke = unit - ku * n_edges; }
                else          { ke = unit / n_along; ku = unit - ke * n_along; }
                const int e = (ke + 1) * 4, u = ku * 8, seg = t & 1;
                const int x = vertical ? e : u + 4 * seg, y = vertical ? u + 4 * seg : e;
                if (vertical ? y < F.height : x < F.width) {
                    const int bs = gld<uint8_t>((const uint8_t *)F.bs[0] + (y >> 2) * F.min_tu_width + (x >> 2));
                    active = bs != 0;
                    desc = (uint32_t)x | ((uint32_t)y << 14) | ((uint32_t)bs << 28);
                }
            }
        }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint64_t m = __ballot(active);
    if (lane == 0)
        wave_cnt[wave] = __popcll(m);
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int n = wave_cnt[k];
        if (k < wave) base += n;
        total += n;
    }
    if (active)
        slot[base + __popcll(m & ((1ull << lane) - 1))] = desc;
    __syncthreads();
    if ((int)threadIdx.x >= total)
        return;
    const uint32_t mine = slot[threadIdx.x];
    const bool is_chroma = mine >> 31;
    // ---- phase B: the segment's parameters and the filter
    const int t = blockIdx.x * blockDim.x + (is_chroma ? (int)(mine & 255) : 0);
    const int unit = t >> 1;
    const int c = !is_chroma ? 0 : unit >= u2 ? 2 : 1;
    const int local = unit - (c == 2 ? u2 : c == 1 ? u1 : 0);
    const int n_along = c ? na1 : na0;
    const int hs = c ? F.hs : 0, vs = c ? F.vs : 0;
    const int grid = c ? (8 << (vertical ? hs : vs)) : 4, step = 8 << (vertical ? vs : hs);
    const int n_edges = c ? ne1 : ne0;
    int e, u;
    if (is_chroma) {
        // consecutive units walk along the rows of the picture: across the edges for the vertical pass, along the edge for the
        // horizontal one (n_along = units along one edge, n_edges = edges)
        int ke, ku;
        if (vertical) { ku = local / n_edges; ke = local - ku * n_edges; }
        else          { ke = local / n_along; ku = local - ke * n_along; }
        e = (ke + 1) * grid; u = ku * step;                          // edge position across, unit position along (luma units)
    } else {
        const int x = mine & 0x3fff, y = (mine >> 14) & 0x3fff;
        e = vertical ? x : y; u = vertical ? y : x;                  // (u = the segment's own position along the edge)
    }
    const int ux = vertical ? e : u, uy = vertical ? u : e;
    const int hor_ctu_edge = !vertical && !(e & ((1 << F.ctb_log2) - 1));
    const int ctb = (ux >> F.ctb_log2) + (uy >> F.ctb_log2) * F.ctb_width;
    const int8_t *dbp = (const int8_t *)F.db_params + ctb * 6;
    const int shift = vertical ? vs : hs, lines = c ? (shift ? 2 : 4) : 4, nseg = 8 / lines;
    const uint8_t *bs_tab = (const uint8_t *)sel3(c, F.bs);
    uint8_t *plane = (uint8_t *)sel3(c, F.plane);
    const int stride = sel3(c, F.stride), pxstride = stride / (int)sizeof(px_t);
    const int xs = vertical ? 1 : pxstride, ys = vertical ? pxstride : 1;
    if (!c) {
        // luma: this lane's one 4-line segment
        const int x = mine & 0x3fff, y = (mine >> 14) & 0x3fff, bs = (mine >> 28) & 3;
        const int tu = (y >> 2) * F.min_tu_width + (x >> 2);
        const int xp = x - vertical, yp = y - !vertical;
        uint8_t *pix = plane + row_off(y, stride) + x * (int)sizeof(px_t);
        const int8_t *qy = (const int8_t *)F.qp_y;
        const int a = gld<int8_t>(qy + (xp >> F.min_cb_log2) + (yp >> F.min_cb_log2) * F.min_cb_width);
        const int b = gld<int8_t>(qy + (x >> F.min_cb_log2) + (y >> F.min_cb_log2) * F.min_cb_width);
        const int len_p = gld<uint8_t>((const uint8_t *)F.max_len_p + tu), len_q = gld<uint8_t>((const uint8_t *)F.max_len_q + tu);
        const int beta_offset = gld<int8_t>(dbp), tc_offset = gld<int8_t>(dbp + 3);
        int qp = (a + b + 1) >> 1;
        if (F.ladf_enabled) {
            // lf.ladf_level (vvc_filter_template.c:788-803) and the interval search of get_qp_y (:840-846)
            const int level = (ld_px<BD>(pix, -xs) + ld_px<BD>(pix, -xs + 3 * ys) + ld_px<BD>(pix, 0) + ld_px<BD>(pix, 3 * ys)) >> 2;
            int qp_offset = F.ladf_lowest_qp_offset;
            bool go = true;                              // (no break: the loop must unroll so that the table indices are constants)
#pragma unroll
            for (int k = 0; k < 4; k++) {
                go = go && k < F.num_ladf_intervals - 1 && level > F.ladf_lower_bound[k + 1];
                if (go)
                    qp_offset = F.ladf_qp_offset[k];
            }
            qp += qp_offset;
        }
        const int beta = gld<uint8_t>(kBetaTable + clip3(qp + beta_offset, 0, 63));
        const int tc = gld<uint16_t>(kTcTable + clip3(qp + 2 * (bs - 1) + (tc_offset & -2), 0, 65));
        deblock_luma_seg<BD>(pix, xs, ys, tc, beta, 0, 0, len_p, len_q, hor_ctu_edge);
        return;
    }
    const int8_t *qc = (const int8_t *)(c == 1 ? F.qp_c[0] : F.qp_c[1]);
    const uint8_t *tbs = (const uint8_t *)F.tb_size_c;
    for (int seg = t & 1; seg < nseg; seg += 2) {
        const int x = vertical ? e : u + 4 * seg, y = vertical ? u + 4 * seg : e;
        if (vertical ? y >= F.height : x >= F.width)
            continue;
        const int tu = (y >> 2) * F.min_tu_width + (x >> 2);
        const int bs = gld<uint8_t>(bs_tab + tu);
        if (!bs)
            continue;
        const int xp = x - vertical, yp = y - !vertical;
        const int tup = (yp >> 2) * F.min_tu_width + (xp >> 2);
        uint8_t *pix = plane + row_off(y >> vs, stride) + (x >> hs) * (int)sizeof(px_t);
        const int qp = (gld<int8_t>(qc + tup) + gld<int8_t>(qc + tu) - 2 * F.qp_bd_offset + 1) >> 1;
        const int size_p = gld<uint8_t>(tbs + tup), size_q = gld<uint8_t>(tbs + tu);
        int len_p, len_q;
        if (size_p >= 8 && size_q >= 8) {
            len_q = 3;
            len_p = hor_ctu_edge ? 1 : 3;
        } else {
            len_p = len_q = bs == 2;
        }
        const int beta_offset = gld<int8_t>(dbp + c), tc_offset = gld<int8_t>(dbp + 3 + c);
        const int beta = gld<uint8_t>(kBetaTable + clip3(qp + beta_offset, 0, 63));
        const int tc = gld<uint16_t>(kTcTable + clip3(qp + 2 * (bs - 1) + (tc_offset & -2), 0, 65));
        deblock_chroma_seg<BD>(pix, xs, ys, lines, tc, beta, 0, 0, len_p, len_q);
    }
}

template <int BD>
__global__ void ladf_kernel(const uint8_t *pix, ptrdiff_t xs, ptrdiff_t ys, int *out)
{
    *out = (ld_px<BD>(pix, -xs) + ld_px<BD>(pix, -xs + 3 * ys) + ld_px<BD>(pix, 0) + ld_px<BD>(pix, 3 * ys)) >> 2;
}

// ------------------------------------------------------------------------------------------------ launchers

static void launch_sao(int bd, const vvc355_sao_job *jobs, int n, int max_w, int max_h, hipStream_t st)
{
    if (n <= 0) return;
    const int gx = max(1, min(16, (max_w * max_h + 1023) / 1024));
    VVC355_BD_DISPATCH(bd, hipLaunchKernelGGL((sao_kernel<BD>), dim3(gx, n), dim3(256), 0, st, jobs));
    HIP_CHECK(hipGetLastError());
}

// batched stage: every job is type 1 (band) or 3 (edge + restore) on 16-byte aligned planes, width <= 128
static void launch_sao_vec(int bd, const vvc355_sao_job *jobs, int n, int max_h, hipStream_t st)
{
    if (n <= 0) return;
    VVC355_BD_DISPATCH(bd, hipLaunchKernelGGL((sao_vec_kernel<BD>), dim3((max_h + 63) / 64, n), dim3(256), 0, st, jobs));
    HIP_CHECK(hipGetLastError());
}

static void launch_deblock(int bd, const vvc355_deblock_job *jobs, int n, hipStream_t st)
{
    if (n <= 0) return;
    VVC355_BD_DISPATCH(bd, hipLaunchKernelGGL((deblock_kernel<BD>), dim3((n * 2 + 255) / 256), dim3(256), 0, st, jobs, n));
    HIP_CHECK(hipGetLastError());
}

static void slot_deblock(int bd, int dir, int chroma, uint8_t *pix, ptrdiff_t stride, const int32_t *beta, const int32_t *tc,
                         const uint8_t *no_p, const uint8_t *no_q, const uint8_t *max_len_p, const uint8_t *max_len_q, int flag)
{
    const int px = bd > 8 ? 2 : 1;
    const int nseg = chroma ? (flag ? 4 : 2) : 2;
    SlotCall call;
    // samples across the edge: 8 on each side (luma long filters read P7/Q7); 8 along it
    const Staged s = dir == 0 ? call.rect(pix, stride, 0, 8 * px, -8, 8, true, true)
                              : call.rect(pix, stride, -8 * px, 8 * px, 0, 8, true, true);
    vvc355_deblock_job job = {};
    job.pix = (uint64_t)s.dev; job.stride = (int32_t)s.pitch;
    job.dir = (uint8_t)dir; job.chroma = (uint8_t)chroma; job.flag = (uint8_t)flag;
    for (int i = 0; i < nseg; i++) {
        job.beta[i] = beta[i]; job.tc[i] = tc[i];
        job.no_p[i] = no_p[i]; job.no_q[i] = no_q[i]; job.max_len_p[i] = max_len_p[i]; job.max_len_q[i] = max_len_q[i];
    }
    launch_deblock(bd, call.upload(&job, 1), 1, call.stream());
}

// ---------------------------------------------------------------------------------------------- boundary strengths

// boundary_strength (vvc_filter.c:308-372): the motion rule between two inter blocks.  pc / pn = reference POC lists of the
// slices the two blocks belong to (int32 [2][32]).
__device__ __forceinline__ bool mv_far(const int32_t *a, const int32_t *b) { return abs(a[0] - b[0]) >= 8 || abs(a[1] - b[1]) >= 8; }

__device__ __forceinline__ int bs_motion(const vvc355_mvfield &c, const vvc355_mvfield &n, const int *pc, const int *pn)
{
    if (c.pred_flag == 3 && n.pred_flag == 3) {
        const int c0 = gld<int>(pc + c.ref_idx[0]), c1 = gld<int>(pc + 32 + c.ref_idx[1]);
        const int n0 = gld<int>(pn + n.ref_idx[0]), n1 = gld<int>(pn + 32 + n.ref_idx[1]);
        if (c0 == n0 && c0 == c1 && n0 == n1)
            return (mv_far(n.mv[0], c.mv[0]) || mv_far(n.mv[1], c.mv[1])) && (mv_far(n.mv[1], c.mv[0]) || mv_far(n.mv[0], c.mv[1]));
        if (n0 == c0 && n1 == c1)
            return mv_far(n.mv[0], c.mv[0]) || mv_far(n.mv[1], c.mv[1]);
        if (n1 == c0 && n0 == c1)
            return mv_far(n.mv[1], c.mv[0]) || mv_far(n.mv[0], c.mv[1]);
        return 1;
    }
    if (c.pred_flag != 3 && n.pred_flag != 3) {
        const bool c_l0 = c.pred_flag & 1, n_l0 = n.pred_flag & 1;
        const int ra = gld<int>(pc + (c_l0 ? c.ref_idx[0] : 32 + c.ref_idx[1]));
        const int rb = gld<int>(pn + (n_l0 ? n.ref_idx[0] : 32 + n.ref_idx[1]));
        if (ra != rb)
            return 1;
        const int ax = c_l0 ? c.mv[0][0] : c.mv[1][0], ay = c_l0 ? c.mv[0][1] : c.mv[1][1];
        const int bx = n_l0 ? n.mv[0][0] : n.mv[1][0], by = n_l0 ? n.mv[0][1] : n.mv[1][1];
        return abs(ax - bx) >= 8 || abs(ay - by) >= 8;
    }
    return 1;
}

__device__ __forceinline__ vvc355_mvfield ld_mvf(const vvc355_mvfield *p)
{
    uint64_t w[3];
#pragma unroll
    for (int i = 0; i < 3; i++) w[i] = gld<uint64_t>((const uint64_t *)p + i);
    vvc355_mvfield r;
    __builtin_memcpy(&r, w, 24);
    return r;
}

// One lane per 4x4 luma unit; both edge directions (dir 1 = vertical edges: neighbour on the left, dir 0 = horizontal edges:
// neighbour above).  Gather form of vvc_deblock_bs (vvc_filter.c:756-783): the unit asks which rule of the transform unit
// covering it wrote its entry in the reference's scatter loops.
//
// The kernel is a chain of table look-ups, so it is organised by memory round trips, not by rules: phase 1 issues every load whose
// address depends on the lane's position only (its own and the two neighbouring MvFields, the transform-unit tables of both
// trees, the coded / pcm flags of the unit and of its two neighbours, slice and tile numbers), unconditionally and with clamped
// indices; phase 2 the few loads addressed by the transform unit's origin (its coding block); then the rules run on registers.
__global__ __launch_bounds__(256) void deblock_bs_kernel(const vvc355_bs_frame *__restrict__ fp)
{
    const vvc355_bs_frame F = load_uniform(fp);
    const int mtw = F.min_tu_width, mpw = F.min_pu_width, mcl = F.min_cb_log2, mcw = F.min_cb_width;
    const int ux = blockIdx.x * 64 + (threadIdx.x & 63), uy = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int x = ux * 4, y = uy * 4;
    if (x >= F.width || y >= F.height)
        return;
    const int off = uy * mtw + ux;
    const int ctb_log2 = F.ctb_log2, ctb_mask = (1 << ctb_log2) - 1;
    const vvc355_mvfield *mvf = (const vvc355_mvfield *)F.mvf;
    const int16_t *slice = (const int16_t *)F.slice_idx;
    const int ctb = (y >> ctb_log2) * F.ctb_width + (x >> ctb_log2);

    // ---- phase 1
    const int has_n[2] = { uy > 0, ux > 0 };                                     // [dir]: a P side exists
    const int offn[2] = { has_n[0] ? off - mtw : off, has_n[1] ? off - 1 : off };
    const vvc355_mvfield curr = ld_mvf(mvf + uy * mpw + ux);
    const vvc355_mvfield neigh[2] = { ld_mvf(mvf + (uy - has_n[0]) * mpw + ux), ld_mvf(mvf + uy * mpw + ux - has_n[1]) };
    int t0[2][2];                                                                // [tree][dir]: the transform unit's origin across the edge
#pragma unroll
    for (int t = 0; t < 2; t++) {
        t0[t][1] = gld<int>((const int *)F.tb_pos_x0[t] + off);
        t0[t][0] = gld<int>((const int *)F.tb_pos_y0[t] + off);
    }
    const int size_q[2] = { gld<uint8_t>((const uint8_t *)F.tb_height[0] + off), gld<uint8_t>((const uint8_t *)F.tb_width[0] + off) };
    const int size_p[2] = { gld<uint8_t>((const uint8_t *)F.tb_height[0] + offn[0]), gld<uint8_t>((const uint8_t *)F.tb_width[0] + offn[1]) };
    const uint8_t *flag_tab[6] = { (const uint8_t *)F.pcmf[0], (const uint8_t *)F.tu_coded_flag[0], (const uint8_t *)F.pcmf[1],
                                   (const uint8_t *)F.tu_coded_flag[1], (const uint8_t *)F.tu_coded_flag[2], (const uint8_t *)F.tu_joint_cbcr };
    int fq[6], fn[2][6];                                                         // pcm0, cbf0, pcm1, cbf1, cbf2, joint: here / on the P side
#pragma unroll
    for (int k = 0; k < 6; k++) {
        fq[k] = gld<uint8_t>(flag_tab[k] + off);
        fn[0][k] = gld<uint8_t>(flag_tab[k] + offn[0]);
        fn[1][k] = gld<uint8_t>(flag_tab[k] + offn[1]);
    }
    int sb_p[2];                                                                 // the P side lies in a sub-block coding block
#pragma unroll
    for (int d = 0; d < 2; d++) {
        const int px = d ? x - has_n[1] : x, py = d ? y : y - has_n[0];
        const int cbp = (py >> mcl) * mcw + (px >> mcl);
        sb_p[d] = gld<uint8_t>((const uint8_t *)F.msf + cbp) | gld<uint8_t>((const uint8_t *)F.iaf + cbp);
    }
    const int my_slice = gld<int16_t>(slice + ctb);
    int n_slice[2], tile_edge[2];
#pragma unroll
    for (int d = 0; d < 2; d++) {
        const int a = d ? x : y;
        const bool on_ctb_edge = a > 0 && !(a & ctb_mask);
        n_slice[d] = gld<int16_t>(slice + (on_ctb_edge ? (d ? ctb - 1 : ctb - F.ctb_width) : ctb));
        const int16_t *bd = (const int16_t *)(d ? F.ctb_to_col_bd : F.ctb_to_row_bd);
        const int r = max(a >> ctb_log2, 1);
        tile_edge[d] = on_ctb_edge && gld<int16_t>(bd + r) != gld<int16_t>(bd + r - 1);
    }
    // ---- phase 2: the luma transform unit's origin and the coding block there
    const int tx0 = t0[0][1], ty0 = t0[0][0];
    const bool is_intra = gld<uint8_t>((const uint8_t *)(mvf + (ty0 >> 2) * mpw + (tx0 >> 2)) + 20) == 0;
    const int cbo = (ty0 >> mcl) * mcw + (tx0 >> mcl);
    const int cb0[2] = { gld<int>((const int *)F.cb_pos_y + cbo), gld<int>((const int *)F.cb_pos_x + cbo) };
    const int cb_size[2] = { gld<uint8_t>((const uint8_t *)F.cb_height + cbo), gld<uint8_t>((const uint8_t *)F.cb_width + cbo) };
    const bool sb_cu = !is_intra && (gld<uint8_t>((const uint8_t *)F.msf + cbo) | gld<uint8_t>((const uint8_t *)F.iaf + cbo));

    // ---- phase 3: the rules
#pragma unroll
    for (int dir = 0; dir < 2; dir++) {
        const int a = dir ? x : y;                              // coordinate across the edge
        // a CTB edge that must not be filtered (:498-507, :583-591)
        const bool ctb_edge_off = (!F.lfase && n_slice[dir] != my_slice) || (!F.lfate && tile_edge[dir]);
        const bool strong = curr.pred_flag == 0 || neigh[dir].pred_flag == 0 || curr.ciip_flag || neigh[dir].ciip_flag;
        // ---- luma tree
        int bs = 0, len_p = 0, len_q = 0;
        const bool has_sb = sb_cu && cb_size[dir] > 8;
        if (a == t0[0][dir]) {
            if (a > 0 && !ctb_edge_off) {
                // transform-block edge (:509-545)
                const int off_c = cb0[dir] - a;
                if (fn[dir][0] && fq[0])
                    bs = 0;
                else if (strong)
                    bs = 2;
                else if (fq[1] || fn[dir][1])
                    bs = 1;
                else if (off_c && ((off_c & 7) || !has_sb))
                    bs = 0;
                else
                    bs = bs_motion(curr, neigh[dir], (const int *)F.ref_poc + my_slice * 64, (const int *)F.ref_poc + n_slice[dir] * 64);
                // derive_max_filter_length_luma (:374-397)
                if (size_p[dir] <= 4 || size_q[dir] <= 4) {
                    len_p = len_q = 1;
                } else {
                    len_p = size_p[dir] >= 32 ? 7 : 3;
                    len_q = size_q[dir] >= 32 ? 7 : 3;
                }
                if (has_sb)
                    len_q = min(5, len_q);
                if (sb_p[dir])
                    len_p = min(5, len_p);
            }
        } else if (sb_cu && !((a - cb0[dir]) & 7)) {
            // sub-block edge inside the transform unit (:399-475), both sides in the current slice
            const int *rpl = (const int *)F.ref_poc + my_slice * 64;
            bs = bs_motion(curr, neigh[dir], rpl, rpl);
            const int i = a - t0[0][dir], tsize = size_q[dir];
            len_p = len_q = (i == 4 || i == tsize - 4) ? 1 : (i == 8 || i == tsize - 8) ? 2 : 3;
        }
        gst<uint8_t>((uint8_t *)F.bs[dir][0] + off, (uint8_t)bs);
        gst<uint8_t>((uint8_t *)F.max_len_p[dir] + off, (uint8_t)len_p);
        gst<uint8_t>((uint8_t *)F.max_len_q[dir] + off, (uint8_t)len_q);
        if (F.n_comp < 3)
            continue;
        // ---- chroma tree (:642-754): transform-block edges on the 8-sample chroma grid only
        int bs_cb = 0, bs_cr = 0;
        const int grid = (8 << (dir ? F.hs : F.vs)) - 1;
        if (a == t0[1][dir] && a > 0 && !(a & grid) && !ctb_edge_off && !(fn[dir][2] && fq[2])) {
            const int joint = fn[dir][5] | fq[5];
            bs_cb = strong ? 2 : (fn[dir][3] | fq[3] | joint) ? 1 : 0;
            bs_cr = strong ? 2 : (fn[dir][4] | fq[4] | joint) ? 1 : 0;
        }
        gst<uint8_t>((uint8_t *)F.bs[dir][1] + off, (uint8_t)bs_cb);
        gst<uint8_t>((uint8_t *)F.bs[dir][2] + off, (uint8_t)bs_cr);
    }
}

} // namespace vvc355

using namespace vvc355;

extern "C" {

void vvc355_sao_batch(void *stream, int bd, const vvc355_sao_job *jobs_dev, int n_jobs, int max_w, int max_h)
{
    if (n_jobs <= 0) return;
    launch_sao(bd, jobs_dev, n_jobs, max_w, max_h, (hipStream_t)stream);
}

void vvc355_deblock_bs_pass(void *stream, const vvc355_bs_frame *frame_dev, const vvc355_bs_frame *frame_host)
{
    const vvc355_bs_frame &F = *frame_host;        // host copy: geometry only
    if (F.width <= 0 || F.height <= 0) return;
    if ((F.width & 3) || (F.height & 3) || F.min_tu_width < F.width / 4 || F.min_pu_width < F.width / 4) {
        fprintf(stderr, "vvc_mi355: deblock_bs_pass: %dx%d with table pitches %d / %d outside the driver's domain\n", F.width, F.height, F.min_tu_width, F.min_pu_width);
        abort();
    }
    const int nx = F.width / 4, ny = F.height / 4;
    hipLaunchKernelGGL(deblock_bs_kernel, dim3((nx + 63) / 64, (ny + 3) / 4), dim3(256), 0, (hipStream_t)stream, frame_dev);
    HIP_CHECK(hipGetLastError());
}

void vvc355_sao_frame_pass(void *stream, int bd, const vvc355_sao_frame *frame_dev, const vvc355_sao_frame *frame_host)
{
    const vvc355_sao_frame &F = *frame_host;       // host copy: geometry only
    const int n = F.ctb_width * F.ctb_height;
    if (n <= 0) return;
    const int ctb = 1 << F.ctb_log2;
    // rows one workgroup of the vector body covers for a rectangle of width w (sao_vec_body: 8 samples x 4 rows per lane)
    auto rows_per_wg = [](int w) { const int lxl = w > 64 ? 4 : w > 32 ? 3 : w > 16 ? 2 : w > 8 ? 1 : 0; return (256 >> lxl) * 4; };
    const int tiles_l = (ctb + rows_per_wg(ctb) - 1) / rows_per_wg(ctb);
    int tiles_c = 0, packed = 0;
    if (F.n_comp >= 3) {
        const int wc = ctb >> F.hs, hc = ctb >> F.vs;
        packed = 2 * hc <= rows_per_wg(wc);
        tiles_c = (hc + rows_per_wg(wc) - 1) / rows_per_wg(wc);
    }
    const int gx = tiles_l + (F.n_comp >= 3 ? (packed ? 1 : 2 * tiles_c) : 0);
    VVC355_BD_DISPATCH(bd, hipLaunchKernelGGL((sao_frame_kernel<BD>), dim3(gx, n), dim3(256), 0, (hipStream_t)stream, frame_dev, 4 * gx,
                                              tiles_l, tiles_c, packed));
    HIP_CHECK(hipGetLastError());
}

void vvc355_sao_ctb_batch(void *stream, int bd, const vvc355_sao_job *jobs_dev, int n_jobs, int max_h)
{
    if (n_jobs <= 0) return;
    launch_sao_vec(bd, jobs_dev, n_jobs, max_h, (hipStream_t)stream);
}

void vvc355_deblock_frame_pass(void *stream, int bd, const vvc355_deblock_frame *frame_dev, const vvc355_deblock_frame *frame_host)
{
    if (frame_host->width >= (1 << 14) || frame_host->height >= (1 << 14)) {
        fprintf(stderr, "vvc_mi355: deblock_frame_pass: picture %dx%d beyond the 14-bit segment coordinates\n", frame_host->width, frame_host->height);
        abort();
    }
    // unit counts per component (the host copy of the descriptor is only read for the geometry)
    const vvc355_deblock_frame &F = *frame_host;
    int first[4] = { 0, 0, 0, 0 }, n_along[2] = { 0, 0 }, n_edge[2] = { 0, 0 };
    for (int c = 0; c < F.n_comp; c++) {
        const int hs = c ? F.hs : 0, vs = c ? F.vs : 0;
        const int grid = c ? (8 << (F.vertical ? hs : vs)) : 4, step = 8 << (F.vertical ? vs : hs);
        const int across = F.vertical ? F.width : F.height, along = F.vertical ? F.height : F.width;
        const int n_edges = (across - 1) / grid, n_units = (along + step - 1) / step;
        n_along[c ? 1 : 0] = n_units;
        n_edge[c ? 1 : 0] = n_edges;
        first[c + 1] = first[c] + n_edges * n_units;
    }
    const int total = first[F.n_comp];
    if (total <= 0) return;
    VVC355_BD_DISPATCH(bd, hipLaunchKernelGGL((deblock_frame_kernel<BD>), dim3((total * 2 + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                                              frame_dev, first[1], F.n_comp > 2 ? first[2] : total, total, n_along[0], n_along[1], n_edge[0], n_edge[1]));
    HIP_CHECK(hipGetLastError());
}

void vvc355_deblock_batch(void *stream, int bd, const vvc355_deblock_job *jobs_dev, int n_jobs)
{
    if (n_jobs <= 0) return;
    launch_deblock(bd, jobs_dev, n_jobs, (hipStream_t)stream);
}

void vvc355_lmcs_batch(void *stream, int bd, const vvc355_blend_job *jobs_dev, int n_jobs, int max_w, int max_h)
{
    if (n_jobs <= 0) return;
    const int gx = max(1, min(64, (max_w * max_h + 8191) / 8192));      // 4 vectors of 8 samples per lane
    VVC355_BD_DISPATCH(bd, hipLaunchKernelGGL((lmcs_kernel<BD>), dim3(gx, n_jobs), dim3(256), 0, (hipStream_t)stream, jobs_dev));
    HIP_CHECK(hipGetLastError());
}

void vvc355_lmcs_filter(int bd, uint8_t *dst, ptrdiff_t dst_stride, int width, int height, const uint8_t *lut)
{
    if (width <= 0 || height <= 0) return;
    const int px = bd > 8 ? 2 : 1;
    SlotCall call;
    const Staged d = call.rect(dst, dst_stride, 0, width * px, 0, height, true, true);
    vvc355_blend_job job = {};
    job.dst = (uint64_t)d.dev; job.dst_stride = (int32_t)d.pitch;
    job.src0 = (uint64_t)call.linear(lut, (size_t)px << bd, true, false);
    job.w = (int16_t)width; job.h = (int16_t)height;
    vvc355_lmcs_batch(call.stream(), bd, call.upload(&job, 1), 1, width, height);
}

void vvc355_sao_band_filter(int bd, uint8_t *dst, const uint8_t *src, ptrdiff_t dst_stride, ptrdiff_t src_stride,
                            const int16_t *sao_offset_val, int sao_left_class, int width, int height)
{
    if (width <= 0 || height <= 0) return;
    const int px = bd > 8 ? 2 : 1;
    SlotCall call;
    const Staged s = call.rect(src, src_stride, 0, width * px, 0, height, true, false);
    const Staged d = call.rect(dst, dst_stride, 0, width * px, 0, height, false, true);
    vvc355_sao_job job = {};
    job.dst = (uint64_t)d.dev; job.src = (uint64_t)s.dev; job.dst_stride = (int32_t)d.pitch; job.src_stride = (int32_t)s.pitch;
    job.w = (int16_t)width; job.h = (int16_t)height; job.type = 1; job.band_position = (uint8_t)sao_left_class;
    for (int k = 0; k < 5; k++) job.offset_val[k] = sao_offset_val[k];
    launch_sao(bd, call.upload(&job, 1), 1, width, height, call.stream());
}

void vvc355_sao_edge_filter(int bd, uint8_t *dst, const uint8_t *src, ptrdiff_t dst_stride,
                            const int16_t *sao_offset_val, int eo, int width, int height)
{
    if (width <= 0 || height <= 0) return;
    const int px = bd > 8 ? 2 : 1;
    const ptrdiff_t src_stride = 2 * VVC355_PB + 64;      // implicit: 2*MAX_PB_SIZE + AV_INPUT_BUFFER_PADDING_SIZE (vvcdsp.h:140)
    SlotCall call;
    const Staged s = call.rect(src, src_stride, -px, (width + 1) * px, -1, height + 1, true, false);
    const Staged d = call.rect(dst, dst_stride, 0, width * px, 0, height, false, true);
    vvc355_sao_job job = {};
    job.dst = (uint64_t)d.dev; job.src = (uint64_t)s.dev; job.dst_stride = (int32_t)d.pitch; job.src_stride = (int32_t)s.pitch;
    job.w = (int16_t)width; job.h = (int16_t)height; job.type = 2; job.eo = (uint8_t)eo;
    for (int k = 0; k < 5; k++) job.offset_val[k] = sao_offset_val[k];
    launch_sao(bd, call.upload(&job, 1), 1, width, height, call.stream());
}

void vvc355_sao_edge_restore(int bd, int variant, uint8_t *dst, const uint8_t *src, ptrdiff_t dst_stride, ptrdiff_t src_stride,
                             const int16_t *offset_val, int eo_class, const int *borders, int width, int height,
                             const uint8_t *vert_edge, const uint8_t *horiz_edge, const uint8_t *diag_edge)
{
    if (width <= 0 || height <= 0) return;
    const int px = bd > 8 ? 2 : 1;
    SlotCall call;
    const Staged s = call.rect(src, src_stride, 0, width * px, 0, height, true, false);
    const Staged d = call.rect(dst, dst_stride, 0, width * px, 0, height, true, true);
    vvc355_sao_job job = {};
    job.dst = (uint64_t)d.dev; job.src = (uint64_t)s.dev; job.dst_stride = (int32_t)d.pitch; job.src_stride = (int32_t)s.pitch;
    job.w = (int16_t)width; job.h = (int16_t)height; job.type = 4; job.eo = (uint8_t)eo_class; job.restore = (uint8_t)!!variant;
    for (int k = 0; k < 5; k++) job.offset_val[k] = offset_val[k];
    for (int k = 0; k < 4; k++) { job.borders[k] = (uint8_t)!!borders[k]; job.diag_edge[k] = variant ? diag_edge[k] : 0; }
    for (int k = 0; k < 2; k++) { job.vert_edge[k] = variant ? vert_edge[k] : 0; job.horiz_edge[k] = variant ? horiz_edge[k] : 0; }
    launch_sao(bd, call.upload(&job, 1), 1, width, height, call.stream());
}

void vvc355_lf_filter_luma(int bd, int dir, uint8_t *pix, ptrdiff_t stride, const int32_t *beta, const int32_t *tc,
                           const uint8_t *no_p, const uint8_t *no_q, const uint8_t *max_len_p, const uint8_t *max_len_q, int hor_ctu_edge)
{
    slot_deblock(bd, dir, 0, pix, stride, beta, tc, no_p, no_q, max_len_p, max_len_q, hor_ctu_edge);
}

void vvc355_lf_filter_chroma(int bd, int dir, uint8_t *pix, ptrdiff_t stride, const int32_t *beta, const int32_t *tc,
                             const uint8_t *no_p, const uint8_t *no_q, const uint8_t *max_len_p, const uint8_t *max_len_q, int shift)
{
    slot_deblock(bd, dir, 1, pix, stride, beta, tc, no_p, no_q, max_len_p, max_len_q, shift);
}

int vvc355_lf_ladf_level(int bd, int dir, const uint8_t *pix, ptrdiff_t stride)
{
    const int px = bd > 8 ? 2 : 1;
    int result = 0;
    {
        SlotCall call;
        const Staged s = dir == 0 ? call.rect(pix, stride, 0, 4 * px, -1, 1, true, false)
                                  : call.rect(pix, stride, -px, px, 0, 4, true, false);
        int *out = (int *)call.linear(&result, sizeof(int), false, true);
        const ptrdiff_t ps = s.pitch / px;
        const ptrdiff_t xs = dir == 0 ? ps : 1, ys = dir == 0 ? 1 : ps;
        VVC355_BD_DISPATCH(bd, hipLaunchKernelGGL((ladf_kernel<BD>), dim3(1), dim3(1), 0, call.stream(), (const uint8_t *)s.dev, xs, ys, out));
        HIP_CHECK(hipGetLastError());
    }
    return result;
}

} // extern "C"
