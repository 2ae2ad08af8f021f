// Intra-prediction kernels for gfx950: planar / DC / vertical / horizontal / angular (4-tap fC/fG luma, 2-tap chroma,
// angular PDPC) / MIP leaf predictors, and the whole intra_pred slot (reference-sample preparation, [1 2 1] smoothing,
// projected side references, predictor dispatch, planar/DC/V/H PDPC) with the decoder context flattened into a job.
//
// Reference behaviour: libavcodec/vvc/vvc_intra_template.c:450-1001 and the helpers in libavcodec/vvc/vvc_intra.c:529-690.
// Leaf predictors keep the reference's convention that `stride` counts PIXELS (POS(), vvc_intra_template.c:27).
//
// One workgroup per block: reference samples live in LDS as uint16 with the reference's origin offset (MAX_TB_SIZE + 3),
// lanes sweep the block row-major so that every row store is contiguous.
#include <type_traits>
#include "common.hpp"
#include "runtime.hpp"
#include "../../include/vvc_mi355.h"

namespace vvc355 {

#define VVC355_TABLE(type, name, count) __device__ static const type i_tab_##name[count] __attribute__((aligned(16)))
#include "tables.inc"
#undef VVC355_TABLE

#ifdef VVC355_RECON_PROF
// profiling build only (tools/dbg): 100 MHz wall-clock ticks per phase, accumulated per workgroup in LDS (every lane adds the same
// wave-uniform value, so no branch and no atomic sits inside the walk) and flushed to the device array when the workgroup ends
__device__ unsigned long long vvc355_recon_prof[64];
__device__ __forceinline__ unsigned long long *rprof_lds() { __shared__ unsigned long long p[64]; return p; }
#define RPROF_NOW() wall_clock64()
#define RPROF_ADD(slot, t0) do { unsigned long long *p_ = rprof_lds(); p_[slot] = p_[slot] + (unsigned long long)(wall_clock64() - (t0)); } while (0)
#define RPROF_INC(slot) do { unsigned long long *p_ = rprof_lds(); p_[slot] = p_[slot] + 1ull; } while (0)
// per-CTU timeline: [rs][8] wall-clock stamps (0 ticket, 1 luma wait over, 2 luma flag up, 3 chroma wait over, 4 chroma walk over, 5 done flag up, 6 flags | n_cmd << 8, 7 ticket)
__device__ unsigned long long vvc355_recon_trace[4096 * 8];
#define RTRACE(rs, slot, v) do { vvc355_recon_trace[((rs) & 4095) * 8 + (slot)] = (unsigned long long)(v); } while (0)
#else
#define RTRACE(rs, slot, v) do { } while (0)
#define RPROF_NOW() 0ull
#define RPROF_ADD(slot, t0) do { (void)(t0); } while (0)
#define RPROF_INC(slot) do { } while (0)
#endif

static constexpr int kEdgeOrg = 64 + 3;          // MAX_TB_SIZE + 3
static constexpr int kEdgeLen = 6 * 64 + 5;

// ------------------------------------------------------------------------------------------------ mode helpers (host + device)

// intraPredAngle and invAngle = round(16384 / intraPredAngle) of a directional mode, one packed table entry (angle | inv << 16).
// The reference rounds a float quotient (vvc_intra.c:683-690); no quotient of the 30 table angles is within float error of a half,
// so the integer form is exact (tests/test_oracle_cpu.py checks all of them).
// the reference's inline tables as compile-time constants (tables_small.inc, generated from its initialisers; tables.cpp exports the
// same text for the table check): the packed / arithmetic forms below are proven equal to them by static_assert
#define VVC355_TABLE(type, name, count) static constexpr type c_##name[count]
#include "tables_small.inc"
#undef VVC355_TABLE

__host__ __device__ constexpr uint32_t intra_angle_entry(int aidx)
{
#define VVC355_AI(a) ((uint32_t)(a) | (uint32_t)((16384 + (a) / 2) / (a)) << 16)
    const uint32_t tab[32] = { 0, VVC355_AI(1), VVC355_AI(2), VVC355_AI(3), VVC355_AI(4), VVC355_AI(6), VVC355_AI(8), VVC355_AI(10), VVC355_AI(12),
                               VVC355_AI(14), VVC355_AI(16), VVC355_AI(18), VVC355_AI(20), VVC355_AI(23), VVC355_AI(26), VVC355_AI(29), VVC355_AI(32),
                               VVC355_AI(35), VVC355_AI(39), VVC355_AI(45), VVC355_AI(51), VVC355_AI(57), VVC355_AI(64), VVC355_AI(73), VVC355_AI(86),
                               VVC355_AI(102), VVC355_AI(128), VVC355_AI(171), VVC355_AI(256), VVC355_AI(341), VVC355_AI(512), 0 };
#undef VVC355_AI
    return tab[aidx];
}
constexpr bool intra_angles_match()
{
    for (int i = 0; i < 31; i++)
        if ((int)(intra_angle_entry(i) & 0xffff) != c_intra_angles[i])
            return false;
    return true;
}
static_assert(intra_angles_match(), "intra_angle_entry() != angles[] of ff_vvc_intra_pred_angle_derive (vvc_intra.c:667-670)");
__host__ __device__ inline int intra_angle_index(int mode) { return mode > 34 ? mode - 50 : mode > 0 ? 18 - mode : 16 - mode; }
__host__ __device__ inline void intra_angle_split(int idx, uint32_t e, int *angle, int *inv)
{
    const int a = (int)(e & 0xffff), r = (int)(e >> 16);
    *angle = idx < 0 ? -a : a;
    *inv = idx < 0 ? -r : r;
}
__host__ __device__ inline void intra_angle_inv(int mode, int *angle, int *inv)
{
    const int idx = intra_angle_index(mode);
    intra_angle_split(idx, intra_angle_entry(idx < 0 ? -idx : idx), angle, inv);
}
__host__ __device__ inline int intra_pred_angle(int mode) { int a, r; intra_angle_inv(mode, &a, &r); return a; }
__host__ __device__ inline int intra_inv_angle_of_mode(int mode) { int a, r; intra_angle_inv(mode, &a, &r); return r; }
__host__ __device__ inline int ilog2i(int v)
{
#ifdef __HIP_DEVICE_COMPILE__
    return v <= 1 ? 0 : 31 - __clz(v);
#else
    int r = 0; while (v > 1) { v >>= 1; r++; } return r;
#endif
}
__host__ __device__ inline int intra_nscale(int w, int h, int mode)
{
    if (mode == 0 || mode == 1 || mode == 18 || mode == 50)
        return (ilog2i(w) + ilog2i(h) - 2) >> 2;
    const int inv = intra_inv_angle_of_mode(mode);
    const int side = mode >= 50 ? h : w;
    const int v = ilog2i(side) - ilog2i(3 * inv - 2) + 8;
    return v < 2 ? v : 2;
}
__host__ __device__ inline int intra_need_pdpc(int w, int h, int bdpcm_flag, int mode, int ref_idx)
{
    if (w >= 4 && h >= 4 && !ref_idx && !bdpcm_flag) {
        if (mode == 0 || mode == 1 || mode == 18 || mode == 50) return 1;
        if (mode > 18 && mode < 50) return 0;
        return intra_nscale(w, h, mode) >= 0;
    }
    return 0;
}
// the modes whose reference samples get the [1 2 1] filter (:450): -14 -12 -10 -6 0 2 34 66 72 76 78 80, as a bit test on mode + 14
__host__ __device__ constexpr bool ref_filter_mode(int mode)
{
    const unsigned m = (unsigned)(mode + 14);
    const unsigned long long lo = (1ull << 0) | (1ull << 2) | (1ull << 4) | (1ull << 8) | (1ull << 14) | (1ull << 16) | (1ull << 48);
    const unsigned long long hi = (1ull << (80 - 64)) | (1ull << (86 - 64)) | (1ull << (90 - 64)) | (1ull << (92 - 64)) | (1ull << (94 - 64));
    return m < 64 ? (lo >> m) & 1 : m < 95 ? (hi >> (m - 64)) & 1 : false;
}
constexpr bool ref_filter_modes_match()
{
    for (int mode = -14; mode <= 80; mode++) {
        bool listed = false;
        for (int k = 0; k < 12; k++) listed = listed || c_ref_filter_modes[k] == mode;
        if (ref_filter_mode(mode) != listed)
            return false;
    }
    return true;
}
static_assert(ref_filter_modes_match(), "ref_filter_mode() != modes[] of ff_vvc_ref_filter_flag_derive (vvc_intra.c:657)");
// distance threshold of the 4-tap filter choice (intra_hor_ver_dist_thres, vvc_intra_template.c:559) and CCLM's divSigTable (:261)
__host__ __device__ constexpr int intra_filter_thres(int ti) { return ti == 0 ? 24 : ti == 1 ? 14 : ti == 2 ? 2 : 0; }
__host__ __device__ constexpr int cclm_div_sig(int norm) { return (int)((0x0111122334455670ull >> (4 * norm)) & 15); }
constexpr bool intra_small_tables_match()
{
    for (int i = 0; i < 5; i++) if (intra_filter_thres(i) != c_intra_filter_thres[i]) return false;
    for (int i = 0; i < 16; i++) if (cclm_div_sig(i) != c_cclm_div_sig[i]) return false;
    return true;
}
static_assert(intra_small_tables_match(), "filter thresholds / div_sig differ from vvc_intra_template.c:559,:261");

// reference-sample accessors: pixel-typed global arrays (leaf slots) or uint16 LDS arrays (flattened intra_pred)
template <int BD> struct GRef {
    const uint8_t *p;
    __device__ __forceinline__ int operator()(int i) const { return ld_px<BD>(p, i); }
};
struct LRef {
    const uint16_t *p;
    __device__ __forceinline__ int operator()(int i) const { return p[i]; }
};

// sample-plane accessors, offsets in PIXELS from the accessor's origin: a component plane in HBM, or the RECON stage driver's CTU
// tile in LDS (uint16 samples); at(off) moves the origin
template <int BD> struct GPix {
    uint8_t *p;
    __device__ __forceinline__ int ld(ptrdiff_t off) const { return ld_px<BD>(p, off); }
    __device__ __forceinline__ void st(ptrdiff_t off, int v) const { st_px<BD>(p, off, v); }
    __device__ __forceinline__ GPix at(ptrdiff_t off) const { return GPix{ p + off * (ptrdiff_t)sizeof(typename Px<BD>::type) }; }
    __device__ __forceinline__ void st4(ptrdiff_t off, int a, int b, int c, int d) const { st(off, a); st(off + 1, b); st(off + 2, c); st(off + 3, d); }
    __device__ __forceinline__ void ld4(ptrdiff_t off, int *v) const { v[0] = ld(off); v[1] = ld(off + 1); v[2] = ld(off + 2); v[3] = ld(off + 3); }
};
// The rows above a CTU as the RECON stage driver keeps them in LDS (kApr rows of their own, reaching one CTU to the right for the
// above-right references), or nothing.  `on`: this block's top edge is the CTU's top edge, so every sample above it comes from here.
static constexpr int kApr = 4;          // apron rows above / columns to the left (reference line 3 is line -4)
struct StripRef {
    const uint16_t *p;
    int pitch, x0s;
    bool on;
    __device__ __forceinline__ int at(int x_abs, int r) const { return p[(kApr + r) * pitch + x_abs - x0s]; }
};
static constexpr StripRef kNoStrip = { nullptr, 0, 0, false };

struct LPix {
    uint16_t *p;
    int base;
    __device__ __forceinline__ int ld(int off) const { return p[base + off]; }
    __device__ __forceinline__ void st(int off, int v) const { p[base + off] = (uint16_t)v; }
    __device__ __forceinline__ LPix at(int off) const { return LPix{ p, base + off }; }
    // four samples of one row; the tile keeps every block's first column on an even sample index (4:2:0 chroma blocks start on even
    // columns), so two dword stores
    __device__ __forceinline__ void st4(int off, int a, int b, int c, int d) const
    {
        uint32_t *q = (uint32_t *)(p + base + off);
        q[0] = (uint32_t)a | ((uint32_t)b << 16);
        q[1] = (uint32_t)c | ((uint32_t)d << 16);
    }
    __device__ __forceinline__ void ld4(int off, int *v) const
    {
        const uint32_t *q = (const uint32_t *)(p + base + off);
        const uint32_t a = q[0], b = q[1];
        v[0] = (int)(a & 0xffff); v[1] = (int)(a >> 16); v[2] = (int)(b & 0xffff); v[3] = (int)(b >> 16);
    }
};

// component c's accessor out of three, chosen field by field (a conditional expression on the whole struct would be taken through
// memory — scratch — and lose the pointer's address space)
template <int BD> __device__ __forceinline__ GPix<BD> pick3(int c, GPix<BD> a, GPix<BD> b, GPix<BD> d) { return GPix<BD>{ c == 0 ? a.p : c == 1 ? b.p : d.p }; }
__device__ __forceinline__ LPix pick3(int c, LPix a, LPix b, LPix d) { return LPix{ a.p, c == 0 ? a.base : c == 1 ? b.base : d.base }; }      // one LDS array
__device__ __forceinline__ StripRef pick3(int c, StripRef a, StripRef b, StripRef d)
{
    return StripRef{ a.p + (c == 0 ? 0 : c == 1 ? (int)(b.p - a.p) : (int)(d.p - a.p)), c == 0 ? a.pitch : c == 1 ? b.pitch : d.pitch, c == 0 ? a.x0s : c == 1 ? b.x0s : d.x0s, false };
}

// constant tables the predictors index per sample: straight from the device's constant arrays (batched kernels: many waves hide the
// latency), or from a copy the RECON stage driver keeps in LDS (one wave walks a dependent chain there: every global round trip
// is on the critical path)
struct IntraTabsLds {
    uint32_t angle_inv[32];              // intra_angle_entry
    uint32_t luma_filter[64];            // fC[32] then fG[32], four int8 taps per entry
    uint8_t mip4[1024], mip8[1024], mip16[2688];
};
struct GTabs {
    __device__ __forceinline__ uint32_t angle_inv(int aidx) const { return intra_angle_entry(aidx); }
    __device__ __forceinline__ uint32_t filt4(int e) const { uint32_t d; __builtin_memcpy(&d, i_tab_intra_luma_filter + e * 4, 4); return d; }
    __device__ __forceinline__ const uint8_t *mip(int size_id) const
    {
        return size_id == 0 ? i_tab_mip_matrix_4x4 : size_id == 1 ? i_tab_mip_matrix_8x8 : i_tab_mip_matrix_16x16;
    }
};
struct LTabs {
    const IntraTabsLds *t;
    __device__ __forceinline__ uint32_t angle_inv(int aidx) const { return (uint32_t)__builtin_amdgcn_readfirstlane((int)t->angle_inv[aidx]); }
    __device__ __forceinline__ uint32_t filt4(int e) const { return t->luma_filter[e]; }
    __device__ __forceinline__ const uint8_t *mip(int size_id) const { return size_id == 0 ? t->mip4 : size_id == 1 ? t->mip8 : t->mip16; }
};

// ------------------------------------------------------------------------------------------------ leaf predictors
// Executed by a group of NT lanes (a wave or the whole workgroup); `tid` is the lane's index in its group.

template <int NT> __device__ __forceinline__ void group_sync()
{
    if (NT <= 64) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }     // the group sits inside one wave
    else __syncthreads();
}

template <int BD, int NT, typename R, typename PX>
__device__ void pred_planar(int tid, PX src, int stride, R top, R left, int w, int h)
{
    const int lw = ilog2i(w), lh = ilog2i(h);
    for (int i = tid; i < w * h; i += NT) {
        const int y = i >> lw, x = i & (w - 1);           // block sides are powers of two
        const int pv = ((h - 1 - y) * top(x) + (y + 1) * left(h)) << lw;
        const int ph = ((w - 1 - x) * left(y) + (x + 1) * top(w)) << lh;
        src.st(x + __mul24(stride, y), (pv + ph + w * h) >> (lw + lh + 1));
    }
}

// sum of v over the group's lanes, returned to all of them.  Groups inside a wave: prefix sums along the rows of 16 (DPP), then the
// row totals through readlane; the workgroup: an LDS accumulator.
template <int NT> __device__ __forceinline__ int group_sum(int v, int tid, int *scratch)
{
    if (NT <= 64) {
        v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);      // row_shr:1
        v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);      // row_shr:2
        v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);      // row_shr:4
        v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);      // row_shr:8
        const int r0 = __builtin_amdgcn_readlane(v, 15), r1 = __builtin_amdgcn_readlane(v, 31);
        const int r2 = __builtin_amdgcn_readlane(v, 47), r3 = __builtin_amdgcn_readlane(v, 63);
        if (NT == 64)
            return r0 + r1 + r2 + r3;
        return (__lane_id() & 32) ? r2 + r3 : r0 + r1;
    } else {
        if (tid == 0) *scratch = 0;
        __syncthreads();
        if (v) atomicAdd(scratch, v);
        __syncthreads();
        return *scratch;
    }
}

template <int BD, int NT, typename R, typename PX>
__device__ void pred_dc(int tid, PX src, int stride, R top, R left, int w, int h, int *scratch)
{
    int part = 0;
    if (w >= h) for (int i = tid; i < w; i += NT) part += top(i);
    if (w <= h) for (int i = tid; i < h; i += NT) part += left(i);
    const unsigned offset = w == h ? (unsigned)w << 1 : (unsigned)max(w, h);
    const int dc = (group_sum<NT>(part, tid, scratch) + (int)(offset >> 1)) >> ilog2i((int)offset);
    const int w4 = (w + 3) & ~3;                          // stores cover whole groups of 4 (:856)
    const int lw4 = ilog2i(w4);
    for (int i = tid; i < w4 * h; i += NT) {
        const int y = i >> lw4, x = i & (w4 - 1);
        src.st(x + __mul24(stride, y), dc);
    }
}

template <int BD, int NT, typename R, typename PX>
__device__ void pred_vh(int tid, PX src, int stride, R ref, int w, int h, bool vertical)
{
    const int ww = vertical ? w : (w + 3) & ~3;           // pred_h stores whole groups of 4 (:885)
    const int lww = ilog2i(ww);
    for (int i = tid; i < ww * h; i += NT) {
        const int y = i >> lww, x = i & (ww - 1);
        src.st(x + __mul24(stride, y), vertical ? ref(x) : ref(y));
    }
}

template <int BD, typename R, typename TB>
__device__ __forceinline__ int angular_sample(R ref, int i, int fact, int c_idx, int filter_flag, TB tabs)
{
    if (!fact && (c_idx || !filter_flag))
        return ref(i + 1);
    if (!c_idx) {
        const int f = (int)tabs.filt4(filter_flag * 32 + fact);
        return clip_px<BD>((ref(i) * (int)(int8_t)f + ref(i + 1) * (int)(int8_t)(f >> 8) + ref(i + 2) * (int)(int8_t)(f >> 16) + ref(i + 3) * (f >> 24) + 32) >> 6);
    }
    return ((32 - fact) * ref(i + 1) + fact * ref(i + 2) + 16) >> 5;
}

template <int BD, int NT, typename R, typename PX, typename TB = GTabs>
__device__ void pred_angular(int tid, PX src, int stride, R top, R left, int w, int h, bool vertical,
                             int c_idx, int mode, int ref_idx, int filter_flag, int need_pdpc, TB tabs = TB{})
{
    const int angle = intra_pred_angle(mode);
    int inv = 0, nscale = 0;
    if (need_pdpc) {
        inv = intra_inv_angle_of_mode(mode);
        nscale = intra_nscale(w, h, mode);
    }
    const int base = -(1 + ref_idx), lw = ilog2i(w);
    for (int i = tid; i < w * h; i += NT) {
        const int y = i >> lw, x = i & (w - 1);
        const int along = vertical ? x : y, across = vertical ? y : x;
        const int pos = (1 + ref_idx + across) * angle;
        const int idx = (pos >> 5) + ref_idx, fact = pos & 31;
        int pred = vertical ? angular_sample<BD>(top, base + along + idx, fact, c_idx, filter_flag, tabs)
                            : angular_sample<BD>(left, base + along + idx, fact, c_idx, filter_flag, tabs);
        if (need_pdpc) {
            if (vertical) {
                if (x < min(w, 3 << nscale)) {
                    const int l = left(y + ((256 + (x + 1) * inv) >> 9));
                    pred = clip_px<BD>(pred + (((l - pred) * (32 >> ((x << 1) >> nscale)) + 32) >> 6));
                }
            } else if (y < (3 << nscale)) {
                const int t = top(x + ((256 + (y + 1) * inv) >> 9));
                pred = clip_px<BD>(pred + (((t - pred) * (32 >> min(31, (y * 2) >> nscale)) + 32) >> 6));
            }
        }
        src.st(x + __mul24(stride, y), pred);
    }
}

// ---- the same predictors four samples per lane and step (w >= 4): one row's four neighbours share the row's weights and, for the
// directional modes, the reference window (seven samples feed four 4-tap outputs), and the planar / DC / vertical / horizontal
// PDPC (:654-682) is applied in registers before the store instead of in a second pass over the block.  A single wave walking a
// CTU's blocks in order (the RECON stage driver) is bound by its instruction count, not by lanes.
// pdpc_mode: -1 = none, else the prediction mode (0, 1, 18, 50) whose position-dependent weights apply
template <int BD, typename R>
__device__ __forceinline__ int pdpc_simple(int val, int x, int y, int pdpc_mode, int scale, R top, R left, int tl_top, int tl_left)
{
    int l, t, wl, wt;
    if (pdpc_mode == 0 || pdpc_mode == 1) {
        l = left(y); t = top(x);
        wl = 32 >> min((x << 1) >> scale, 31);
        wt = 32 >> min((y << 1) >> scale, 31);
    } else {
        l = left(y) - tl_left + val; t = top(x) - tl_top + val;
        wl = pdpc_mode == 50 ? 32 >> min((x << 1) >> scale, 31) : 0;
        wt = pdpc_mode == 18 ? 32 >> min((y << 1) >> scale, 31) : 0;
    }
    return clip_px<BD>(val + ((wl * (l - val) + wt * (t - val) + 32) >> 6));
}

// KIND: 0 planar, 1 DC (dc = the mean), 2 vertical, 3 horizontal; V = samples per lane and step (4: w >= 4; 1: any block); PDPC: the
// position-dependent weights of the block's own mode apply (mode 0 / 1 / 50 / 18 for KIND 0 / 1 / 2 / 3).  All compile-time: a wave
// that walks blocks one after the other pays for every branch inside these loops.
template <int BD, int NT, int V, int KIND, bool PDPC, typename R, typename PX>
__device__ void pred_simple_k(int tid, PX src, int stride, R top, R left, int w, int h, int dc)
{
    constexpr int LV = V == 4 ? 2 : 0;
    constexpr int pdpc_mode = KIND == 0 ? 0 : KIND == 1 ? 1 : KIND == 2 ? 50 : 18;
    const int lw = ilog2i(w), lh = ilog2i(h), lq = lw - LV, scale = (lw + lh - 2) >> 2;
    const int tw = KIND == 0 ? top(w) : 0, lhh = KIND == 0 ? left(h) : 0;
    const int tl_top = (PDPC && KIND >= 2) ? top(-1) : 0, tl_left = (PDPC && KIND >= 2) ? left(-1) : 0;
    for (int q = tid; q < (h << lq); q += NT) {
        const int y = q >> lq, x0 = (q & ((1 << lq) - 1)) << LV;
        int v[V];
        const int l = (KIND == 0 || KIND == 3) ? left(y) : 0;
#pragma unroll
        for (int e = 0; e < V; e++) {
            const int x = x0 + e;
            if (KIND == 0) {
                const int pv = ((h - 1 - y) * top(x) + (y + 1) * lhh) << lw;
                const int ph = ((w - 1 - x) * l + (x + 1) * tw) << lh;
                v[e] = (pv + ph + w * h) >> (lw + lh + 1);
            } else if (KIND == 1)
                v[e] = dc;
            else if (KIND == 2)
                v[e] = top(x);
            else
                v[e] = l;
            if (PDPC)
                v[e] = pdpc_simple<BD>(v[e], x, y, pdpc_mode, scale, top, left, tl_top, tl_left);
        }
        if constexpr (V == 4)
            src.st4(x0 + __mul24(stride, y), v[0], v[1], v[2], v[3]);
        else
            src.st(x0 + __mul24(stride, y), v[0]);
    }
}
template <int BD, int NT, int V, typename R, typename PX>
__device__ void pred_simple_q(int tid, PX src, int stride, R top, R left, int w, int h, int kind, int dc, bool pdpc)
{
    switch (kind * 2 + (pdpc ? 1 : 0)) {
    case 0: pred_simple_k<BD, NT, V, 0, false>(tid, src, stride, top, left, w, h, dc); break;
    case 1: pred_simple_k<BD, NT, V, 0, true>(tid, src, stride, top, left, w, h, dc); break;
    case 2: pred_simple_k<BD, NT, V, 1, false>(tid, src, stride, top, left, w, h, dc); break;
    case 3: pred_simple_k<BD, NT, V, 1, true>(tid, src, stride, top, left, w, h, dc); break;
    case 4: pred_simple_k<BD, NT, V, 2, false>(tid, src, stride, top, left, w, h, dc); break;
    case 5: pred_simple_k<BD, NT, V, 2, true>(tid, src, stride, top, left, w, h, dc); break;
    case 6: pred_simple_k<BD, NT, V, 3, false>(tid, src, stride, top, left, w, h, dc); break;
    default: pred_simple_k<BD, NT, V, 3, true>(tid, src, stride, top, left, w, h, dc); break;
    }
}

// directional modes: lanes own four samples ALONG the main reference (a row's four for the vertical modes, a column's four for the
// horizontal ones): they share the row's (column's) offset and fraction, and seven reference samples cover their four windows
// VERT / LUMA / PDPC are compile-time.  A zero fraction needs no case of its own where the reference copies ref[i + 1]: the cubic filter's
// entry 0 is {0, 64, 0, 0} and the chroma two-tap form with fact = 0 is (32 ref[i + 1] + 16) >> 5 — both give ref[i + 1] exactly (with
// the smoothing filter fG the reference filters at fact = 0 too).
template <int BD, int NT, int V, bool VERT, bool LUMA, bool PDPC, typename R, typename PX, typename TB>
__device__ void pred_angular_k(int tid, PX src, int stride, R top, R left, int w, int h, int angle, int inv, int nscale, int ref_idx, int filter_flag, TB tabs)
{
    constexpr int LV = V == 4 ? 2 : 0;
    const int base = -(1 + ref_idx);
    const int n_along = VERT ? w : h, n_across = VERT ? h : w;
    const int lq = ilog2i(n_along) - LV, lw = ilog2i(w);
    const R ref = VERT ? top : left, side = VERT ? left : top;
    const int pd_lim = PDPC ? (VERT ? min(w, 3 << nscale) : (3 << nscale)) : 0;
    for (int q = tid; q < (n_across << lq); q += NT) {
        // V = 4, vertical: q walks rows of quads (stores are 4-sample row pieces); horizontal: consecutive lanes take consecutive
        // columns.  V = 1: q is the sample's raster index, whatever the direction (row stores stay contiguous).
        int across, a0;
        if (V == 4) {
            across = VERT ? q >> lq : q & (n_across - 1);
            a0 = VERT ? (q & ((1 << lq) - 1)) << 2 : (q >> ilog2i(n_across)) << 2;
        } else {
            const int y = q >> lw, x = q & (w - 1);
            across = VERT ? y : x;
            a0 = VERT ? x : y;
        }
        const int pos = (1 + ref_idx + across) * angle;
        const int idx = (pos >> 5) + ref_idx, fact = pos & 31;
        const int b = base + a0 + idx;
        int v[V];
        if (LUMA) {
            const int f = (int)tabs.filt4(filter_flag * 32 + fact);
            const int f0 = (int)(int8_t)f, f1 = (int)(int8_t)(f >> 8), f2 = (int)(int8_t)(f >> 16), f3 = f >> 24;
            int r[V + 3];
#pragma unroll
            for (int e = 0; e < V + 3; e++) r[e] = ref(b + e);
#pragma unroll
            for (int e = 0; e < V; e++) v[e] = clip_px<BD>((r[e] * f0 + r[e + 1] * f1 + r[e + 2] * f2 + r[e + 3] * f3 + 32) >> 6);
        } else {
            int r[V + 1];
#pragma unroll
            for (int e = 0; e < V + 1; e++) r[e] = ref(b + 1 + e);
#pragma unroll
            for (int e = 0; e < V; e++) v[e] = ((32 - fact) * r[e] + fact * r[e + 1] + 16) >> 5;
        }
        if (PDPC) {
#pragma unroll
            for (int e = 0; e < V; e++) {
                const int p = a0 + e;               // x for the vertical modes, y for the horizontal ones
                if (p < pd_lim) {
                    const int s_ = side(across + ((256 + (p + 1) * inv) >> 9));
                    const int wgt = VERT ? 32 >> ((p << 1) >> nscale) : 32 >> min(31, (p * 2) >> nscale);
                    v[e] = clip_px<BD>(v[e] + (((s_ - v[e]) * wgt + 32) >> 6));
                }
            }
        }
        if (V == 1)
            src.st(VERT ? a0 + __mul24(stride, across) : across + __mul24(stride, a0), v[0]);
        else if (VERT)
            src.st4(a0 + __mul24(stride, across), v[0], v[1], v[2], v[3]);
        else {
#pragma unroll
            for (int e = 0; e < V; e++) src.st(across + __mul24(stride, a0 + e), v[e]);
        }
    }
}
template <int BD, int NT, int V, typename R, typename PX, typename TB>
__device__ void pred_angular_q(int tid, PX src, int stride, R top, R left, int w, int h, bool vertical,
                               int c_idx, int angle, int inv, int nscale, int ref_idx, int filter_flag, int need_pdpc, TB tabs)
{
#define VVC355_ANG(VERT, LUMA, PDPC) pred_angular_k<BD, NT, V, VERT, LUMA, PDPC>(tid, src, stride, top, left, w, h, angle, inv, nscale, ref_idx, filter_flag, tabs)
    switch ((vertical ? 4 : 0) + (c_idx == 0 ? 2 : 0) + (need_pdpc ? 1 : 0)) {
    case 0: VVC355_ANG(false, false, false); break;
    case 1: VVC355_ANG(false, false, true); break;
    case 2: VVC355_ANG(false, true, false); break;
    case 3: VVC355_ANG(false, true, true); break;
    case 4: VVC355_ANG(true, false, false); break;
    case 5: VVC355_ANG(true, false, true); break;
    case 6: VVC355_ANG(true, true, false); break;
    default: VVC355_ANG(true, true, true); break;
    }
#undef VVC355_ANG
}

// MIP (:708-824).  `red` = 16 ints of LDS scratch.
template <int BD, int NT, typename R, typename PX, typename TB = GTabs>
__device__ void pred_mip(int tid, PX src, int stride, R top, R left, int w, int h, int mode_id, int transposed, int *red, TB tabs = TB{})
{
    const int size_id = (w == 4 && h == 4) ? 0 : ((w == 4 || h == 4) || (w == 8 && h == 8)) ? 1 : 2;
    const int bsize = size_id == 0 ? 2 : 4, psize = size_id == 2 ? 8 : 4;
    const int in_size = 2 * bsize - (size_id == 2);
    const uint8_t *matrix = tabs.mip(size_id) + mode_id * (size_id == 0 ? 16 * 4 : size_id == 1 ? 16 * 8 : 64 * 7);
    const int up_h = w / psize, up_v = h / psize;
    group_sync<NT>();
    for (int k = tid; k < 2 * bsize; k += NT) {
        // boundary down-sampling: first bsize entries from the top row (left column when transposed), then the other side
        const int second = k >= bsize, from_top = second == (transposed != 0);
        const int len = from_top ? w : h, per = len / bsize, i0 = (k - second * bsize) * per;
        int s = 0;
        for (int j = 0; j < per; j++)
            s += from_top ? top(i0 + j) : left(i0 + j);
        red[k] = per == 1 ? s : (s + (per >> 1)) >> ilog2i(per);
    }
    group_sync<NT>();
    if (tid == 0) {
        const int t0 = red[0];
        int ow, off = 1;
        if (size_id != 2) { off = 0; ow = (1 << (BD - 1)) - t0; }
        else ow = red[1] - t0;
        red[0] = ow;
        for (int i = 1; i < in_size; i++) { red[i] = red[i + off] - t0; ow += red[i]; }
        red[14] = 32 - 32 * ow;
        red[15] = t0;
    }
    group_sync<NT>();
    for (int t = tid; t < psize * psize; t += NT) {
        const int y = psize == 8 ? t >> 3 : t >> 2, x = t & (psize - 1);
        int p = 0;
        for (int i = 0; i < in_size; i++)
            p += red[i] * matrix[(y * psize + x) * in_size + i];
        p = clip3(((p + red[14]) >> 6) + red[15], 0, (1 << BD) - 1);
        const int cx = transposed ? y : x, cy = transposed ? x : y;
        src.st((up_h - 1 + cx * up_h) + stride * (up_v - 1 + cy * up_v), p);
    }
    group_sync<NT>();
    // linear interpolation between the reduced samples (:790-824); the factors are powers of two and the operands non-negative, so
    // the reference's divisions are shifts.  Horizontal: one lane per output sample of the rows that hold reduced samples.
    const int lh_ = ilog2i(up_h), lv_ = ilog2i(up_v), lw_ = ilog2i(w);
    for (int t = tid; up_h > 1 && t < (psize << lw_); t += NT) {
        const int x = t & (w - 1), row = up_v - 1 + (t >> lw_) * up_v;
        const int k = (x & (up_h - 1)) + 1, j = x >> lh_;
        if (k < up_h) {
            const int before = j ? src.ld(j * up_h - 1 + stride * row) : left(row), after = src.ld((j + 1) * up_h - 1 + stride * row);
            src.st(x + stride * row, ((up_h - k) * before + k * after + (up_h >> 1)) >> lh_);
        }
    }
    group_sync<NT>();
    for (int x = tid; up_v > 1 && x < w; x += NT) {                // one lane per column
        int before = top(x);
        for (int j = 0; j < psize; j++) {
            const int after = src.ld(x + stride * ((j + 1) * up_v - 1));
            for (int k = 1; k < up_v; k++)
                src.st(x + stride * (j * up_v + k - 1), ((up_v - k) * before + k * after + (up_v >> 1)) >> lv_);
            before = after;
        }
    }
}

// ------------------------------------------------------------------------------------------------ leaf slot kernel

// kind: 0 planar, 1 dc, 2 v, 3 h, 4 angular_v, 5 angular_h, 6 mip; one workgroup, job in kernel arguments
struct LeafArgs {
    uint8_t *src; const uint8_t *top, *left;
    int stride, w, h, kind, c_idx, mode, ref_idx, filter_flag, need_pdpc, mip_mode, mip_transposed;
};

template <int BD>
__global__ __launch_bounds__(256) void intra_leaf_kernel(LeafArgs a)
{
    __shared__ int scratch[16];
    GRef<BD> top{ a.top }, left{ a.left };
    const GPix<BD> dst{ a.src };
    switch (a.kind) {
    case 0: pred_planar<BD, 256>(threadIdx.x, dst, a.stride, top, left, a.w, a.h); break;
    case 1: pred_dc<BD, 256>(threadIdx.x, dst, a.stride, top, left, a.w, a.h, scratch); break;
    case 2: pred_vh<BD, 256>(threadIdx.x, dst, a.stride, top, a.w, a.h, true); break;
    case 3: pred_vh<BD, 256>(threadIdx.x, dst, a.stride, left, a.w, a.h, false); break;
    case 4: pred_angular<BD, 256>(threadIdx.x, dst, a.stride, top, left, a.w, a.h, true, a.c_idx, a.mode, a.ref_idx, a.filter_flag, a.need_pdpc); break;
    case 5: pred_angular<BD, 256>(threadIdx.x, dst, a.stride, top, left, a.w, a.h, false, a.c_idx, a.mode, a.ref_idx, a.filter_flag, a.need_pdpc); break;
    default: pred_mip<BD, 256>(threadIdx.x, dst, a.stride, top, left, a.w, a.h, a.mip_mode, a.mip_transposed, scratch); break;
    }
}

// ------------------------------------------------------------------------------------------------ flattened intra_pred

// vvc_intra_template.c:467-592 (edge preparation) + :595-683 (dispatch, PDPC); one workgroup per job.
// NT = 32: half a wave per block (w*h <= 64), eight blocks per workgroup; NT = 64: one wave per block (w*h <= 256), four per
// workgroup; both with wave-level synchronisation only.  NT = 256: one workgroup per block.
// the whole slot for one block, executed by a group of NT lanes (tid = lane's index in the group); arr = four edge arrays in LDS
// plane = accessor of the component plane's sample (0, 0); stride in pixels
template <int BD, int NT, typename PX, typename TB = GTabs>
__device__ void intra_pred_body(const vvc355_intra_job &j, PX plane, int stride_px, uint16_t (*arr)[kEdgeLen], int *scratch, int tid,
                                const StripRef sr = kNoStrip, TB tabs = TB{})
{
    const int stride = stride_px;
    const int w = j.w, h = j.h, c_idx = j.c_idx, mode = j.mode, ref_idx = j.ref_idx;
    const bool is_mip = j.is_mip, no_isp = !j.isp_split;
    const PX src = plane.at(__mul24(j.y, stride) + j.x);
    const bool directional = !is_mip && mode != 0 && mode != 1 && mode != 50 && mode != 18;
    int angle = 0, inv = 0, nscale = 0, need_pdpc = 0;
    if (directional) {
        const int aidx = intra_angle_index(mode);
        intra_angle_split(aidx, tabs.angle_inv(aidx < 0 ? -aidx : aidx), &angle, &inv);
    }
    if (w >= 4 && h >= 4 && !ref_idx && !j.bdpcm_flag) {        // intra_need_pdpc with the angle already at hand
        if (!directional)
            need_pdpc = 1;
        else if (!(mode > 18 && mode < 50)) {
            nscale = min(2, ilog2i(mode >= 50 ? h : w) - ilog2i(3 * inv - 2) + 8);
            need_pdpc = nscale >= 0;
        }
    }
    uint16_t *left = arr[0] + kEdgeOrg, *top = arr[1] + kEdgeOrg, *fleft = arr[2] + kEdgeOrg, *ftop = arr[3] + kEdgeOrg;

    const bool rff = is_mip ? false : ref_filter_mode(mode);
    const bool smooth = !ref_idx && w * h > 32 && !c_idx && no_isp && rff;
    const int ref_line = ref_idx == 3 ? -4 : -1 - ref_idx;
    int left_size, top_size, uleft, utop, refw = 0, refh = 0;
    if (is_mip || mode == 0)      { left_size = h + 1; top_size = w + 1; uleft = left_size + smooth; utop = top_size + smooth; }
    else if (mode == 1)           { uleft = left_size = h; utop = top_size = w; }
    else if (mode == 50)          { uleft = left_size = need_pdpc ? h : 1; utop = top_size = w; }
    else if (mode == 18)          { uleft = left_size = h; utop = top_size = need_pdpc ? w : 1; }
    else {
        if (no_isp || c_idx) { refw = w * 2; refh = h * 2; } else { refw = j.cb_width + w; refh = j.cb_height + h; }
        utop = top_size = refw; uleft = left_size = refh;
    }
    const int la = min(uleft, (int)j.left_avail), ta = min(utop, (int)j.top_avail);
    const int prof_o = c_idx ? 32 : 0; (void)prof_o;
    unsigned long long t_ph = RPROF_NOW();
#define RPHASE(slot) do { RPROF_ADD((slot) + prof_o, t_ph); t_ph = RPROF_NOW(); } while (0)
    // Reference samples (:467-592) in one pass.  Entry i of the left column (i = ref_line .. uleft - 1) and of the top row
    // (i = ref_line .. utop - 1) each come from exactly one picture sample: the sample itself where it is available, the last
    // available one of its side past the end, the corner for an empty side, and — when the corner is not available — the nearest
    // available sample of the other side (or mid-grey): the substitution process, resolved per entry instead of by passes.
    {
        const int R = -ref_line, n_left = uleft + R, total = n_left + utop + R;
        const int mid = 1 << (BD - 1);
        for (int e = tid; e < total; e += NT) {
            const bool is_top = e >= n_left;
            const int i = (is_top ? e - n_left : e) - R;
            const int avail = is_top ? ta : la;
            const int ie = i >= 0 ? min(i, avail - 1) : i;            // -1 past the end of an empty side: the corner
            int sx, sy;
            bool grey = false;
            if (ie >= 0 || j.cand_up_left) {
                sx = is_top ? ie : ref_line; sy = is_top ? ref_line : ie;
            } else if (la) {
                sx = ref_line; sy = 0;
            } else if (ta) {
                sx = 0; sy = ref_line;
            } else {
                sx = sy = 0; grey = true;
            }
            int v = mid;
            if (!grey)
                v = (sr.on && sy < 0) ? sr.at(j.x + sx, sy) : src.ld(sx + __mul24(stride, sy));
            (is_top ? top : left)[i] = (uint16_t)v;
        }
    }
    group_sync<NT>();
    RPHASE(22);
    if (rff && smooth) {                                  // ref_filter (:450)
        const int keep_last = left_size == uleft;
        if (tid == 0)
            fleft[-1] = ftop[-1] = (uint16_t)((left[0] + 2 * left[-1] + top[0] + 2) >> 2);
        for (int i = tid; i < uleft - keep_last; i += NT) fleft[i] = (uint16_t)((left[i - 1] + 2 * left[i] + left[i + 1] + 2) >> 2);
        for (int i = tid; i < utop - keep_last; i += NT) ftop[i] = (uint16_t)((top[i - 1] + 2 * top[i] + top[i + 1] + 2) >> 2);
        if (keep_last && tid == 0) { ftop[utop - 1] = top[utop - 1]; fleft[uleft - 1] = left[uleft - 1]; }
        group_sync<NT>();
        left = fleft; top = ftop;
    }
    int filter_flag = 0;
    if (directional) {
        if (!(rff || ref_idx || !no_isp)) {
            const int ti = max(0, ((ilog2i(w) + ilog2i(h)) >> 1) - 2);           // thresholds 24, 14, 2, 0, 0
            const int dist = min(abs(mode - 50), abs(mode - 18));
            filter_flag = dist > intra_filter_thres(ti);
        }
        if (mode >= 34) {
            if (angle < 0) {
                uint16_t *p = top - (ref_idx + 1);
                for (int x = -h + tid; x < 0; x += NT)
                    p[x] = left[-1 - ref_idx + min((x * inv + 256) >> 9, h)];
            } else {
                const uint16_t v = top[refw - 1];
                for (int i = refw + tid; i <= refw + max(1, w / h) * ref_idx + 1; i += NT) top[i] = v;
            }
        } else {
            if (angle < 0) {
                uint16_t *p = left - (ref_idx + 1);
                for (int x = -w + tid; x < 0; x += NT)
                    p[x] = top[-1 - ref_idx + min((x * inv + 256) >> 9, w)];
            } else {
                const uint16_t v = left[refh - 1];
                for (int i = refh + tid; i <= refh + max(1, h / w) * ref_idx + 1; i += NT) left[i] = v;
            }
        }
        group_sync<NT>();
    }

    RPHASE(23);
    LRef T{ top }, L{ left };
    if (is_mip) {
        pred_mip<BD, NT>(tid, src, stride, T, L, w, h, j.mip_mode, j.mip_transposed, scratch, tabs);
        RPHASE(24);
        return;
    }
    // four samples per lane once one-per-lane would take three or more steps (the wave's time is its longest lane's instruction count) and the
    // block has whole groups of four along the store direction; one sample per lane otherwise.  PDPC is applied before the store.
    const bool quads = w * h > 2 * NT && w >= 4 && (!directional || mode >= 34 || h >= 4);
    if (directional) {
        if (quads) pred_angular_q<BD, NT, 4>(tid, src, stride, T, L, w, h, mode >= 34, c_idx, angle, inv, nscale, ref_idx, filter_flag, need_pdpc, tabs);
        else       pred_angular_q<BD, NT, 1>(tid, src, stride, T, L, w, h, mode >= 34, c_idx, angle, inv, nscale, ref_idx, filter_flag, need_pdpc, tabs);
    } else {
        int dc = 0;
        if (mode == 1) {
            int part = 0;
            if (w >= h) for (int i = tid; i < w; i += NT) part += T(i);
            if (w <= h) for (int i = tid; i < h; i += NT) part += L(i);
            const unsigned offset = w == h ? (unsigned)w << 1 : (unsigned)max(w, h);
            dc = (group_sum<NT>(part, tid, scratch) + (int)(offset >> 1)) >> ilog2i((int)offset);
        }
        const int kind = mode == 0 ? 0 : mode == 1 ? 1 : mode == 50 ? 2 : 3;
        // DC and horizontal stores cover whole groups of four columns (:856, :885): blocks narrower than four take the leaf routines
        if (w < 4 && (kind == 1 || kind == 3)) {
            if (kind == 1) pred_dc<BD, NT>(tid, src, stride, T, L, w, h, scratch);
            else           pred_vh<BD, NT>(tid, src, stride, L, w, h, false);
        } else if (quads) pred_simple_q<BD, NT, 4>(tid, src, stride, T, L, w, h, kind, dc, need_pdpc != 0);
        else              pred_simple_q<BD, NT, 1>(tid, src, stride, T, L, w, h, kind, dc, need_pdpc != 0);
    }
    RPHASE(24);
#undef RPHASE
}

template <int BD, int NT>
__global__ __launch_bounds__(256) void intra_pred_kernel(const vvc355_intra_job *__restrict__ jobs, int n_jobs)
{
    constexpr int TBS = 256 / NT;
    __shared__ uint16_t arr_all[TBS][4][kEdgeLen];
    __shared__ int scratch_all[TBS][16];
    const int sub = threadIdx.x / NT;
    const int ji = xcd_chunked(blockIdx.x, gridDim.x) * TBS + sub;
    if (ji >= n_jobs)
        return;
    // a wave (or the workgroup) per job: the descriptor comes through the scalar cache; two jobs per wave: per-lane loads
    const vvc355_intra_job j = NT >= 64 ? load_uniform(jobs + __builtin_amdgcn_readfirstlane(ji)) : jobs[ji];
    intra_pred_body<BD, NT>(j, GPix<BD>{ (uint8_t *)j.plane }, j.stride / (int)sizeof(typename Px<BD>::type), arr_all[sub], scratch_all[sub], threadIdx.x % NT);
}

// ------------------------------------------------------------------------------------------------ host side

// A context that claims neighbours where the picture has none would make the kernels read in front of the staged window (a GPU memory
// fault, which the runtime turns into a bare abort of the process): such calls are outside the slots' domain and end here, with a message.
// (Round 2's unexplained abort in tests/test_host_shim.py was exactly this: lmcs_scale_chroma for a CTU in the picture's first row with
// lc->ctb_up_flag = 1, i.e. avail_t at y = 0.)
static void check_neighbours(const char *slot, int x, int y, int left, int top, int up_left)
{
    if (x < 0 || y < 0 || (left && x == 0) || (top && y == 0) || (up_left && (x == 0 || y == 0))) {
        fprintf(stderr, "vvc_mi355: %s at (%d, %d) claims neighbours outside the picture (left %d, top %d, up-left %d): outside the slot's domain\n",
                slot, x, y, left, top, up_left);
        abort();
    }
}

static void check_intra_dims(int w, int h)
{
    if (w <= 0 || h <= 0 || w > 128 || h > 128 || (w & (w - 1)) || (h & (h - 1))) {
        fprintf(stderr, "vvc_mi355: intra block %dx%d outside the slot's domain (powers of two <= 128)\n", w, h);
        abort();
    }
}

// [lo, hi] index range of the main reference that pred_angular reads
static void angular_main_range(int n_along, int n_across, int c_idx, int mode, int ref_idx, int filter_flag, int *lo, int *hi)
{
    const int angle = intra_pred_angle(mode);
    *lo = 1 << 30; *hi = -(1 << 30);
    for (int across = 0; across < n_across; across++) {
        const int pos = (1 + ref_idx + across) * angle;
        const int idx = (pos >> 5) + ref_idx, fact = pos & 31;
        int a, b;
        if (!fact && (c_idx || !filter_flag)) { a = b = 1; }
        else if (!c_idx) { a = 0; b = 3; }
        else { a = 1; b = 2; }
        const int base = -(1 + ref_idx) + idx;
        if (base + a < *lo) *lo = base + a;
        if (base + n_along - 1 + b > *hi) *hi = base + n_along - 1 + b;
    }
}

static void slot_leaf(int bd, int kind, uint8_t *src, const uint8_t *top, const uint8_t *left, int w, int h, ptrdiff_t stride,
                      int c_idx, int mode, int ref_idx, int filter_flag, int need_pdpc, int mip_mode, int mip_transposed)
{
    check_intra_dims(w, h);
    const int px = bd > 8 ? 2 : 1;
    int t_lo = 0, t_hi = -1, l_lo = 0, l_hi = -1;       // inclusive sample ranges read from top / left
    switch (kind) {
    case 0: t_hi = w; l_hi = h; break;
    case 1: if (w >= h) t_hi = w - 1; if (w <= h) l_hi = h - 1; break;
    case 2: t_hi = w - 1; break;
    case 3: l_hi = h - 1; break;
    case 6: t_hi = w - 1; l_hi = h - 1; break;
    default: {
        const bool vertical = kind == 4;
        int lo, hi;
        angular_main_range(vertical ? w : h, vertical ? h : w, c_idx, mode, ref_idx, filter_flag, &lo, &hi);
        if (vertical) { t_lo = lo; t_hi = hi; } else { l_lo = lo; l_hi = hi; }
        if (need_pdpc) {
            const int inv = intra_inv_angle_of_mode(mode), nscale = intra_nscale(w, h, mode);
            int slo = 1 << 30, shi = -(1 << 30);
            const int n_side = vertical ? h : w, n_pd = vertical ? (w < (3 << nscale) ? w : (3 << nscale)) : (h < (3 << nscale) ? h : (3 << nscale));
            for (int k = 0; k < n_pd; k++) {
                const int o = (256 + (k + 1) * inv) >> 9;
                if (o < slo) slo = o;
                if (o + n_side - 1 > shi) shi = o + n_side - 1;
            }
            if (n_pd > 0) { if (vertical) { l_lo = slo; l_hi = shi; } else { t_lo = slo; t_hi = shi; } }
        }
    } }
    SlotCall call;
    LeafArgs a = {};
    const int w_store = (kind == 1 || kind == 3) ? (w + 3) & ~3 : w;
    const Staged d = call.rect(src, stride * px, 0, w_store * px, 0, h, false, true);
    a.src = d.dev; a.stride = (int)(d.pitch / px);
    if (t_hi >= t_lo) a.top = (const uint8_t *)call.linear(top + (ptrdiff_t)t_lo * px, (size_t)(t_hi - t_lo + 1) * px, true, false) - (ptrdiff_t)t_lo * px;
    if (l_hi >= l_lo) a.left = (const uint8_t *)call.linear(left + (ptrdiff_t)l_lo * px, (size_t)(l_hi - l_lo + 1) * px, true, false) - (ptrdiff_t)l_lo * px;
    a.w = w; a.h = h; a.kind = kind; a.c_idx = c_idx; a.mode = mode; a.ref_idx = ref_idx; a.filter_flag = filter_flag;
    a.need_pdpc = need_pdpc; a.mip_mode = mip_mode; a.mip_transposed = mip_transposed;
    VVC355_BD_DISPATCH(bd, hipLaunchKernelGGL((intra_leaf_kernel<BD>), dim3(1), dim3(256), 0, call.stream(), a));
    HIP_CHECK(hipGetLastError());
}

} // namespace vvc355

using namespace vvc355;

extern "C" {

void vvc355_intra_pred_batch(void *stream, int bd, const vvc355_intra_job *jobs_dev, int n_jobs, int max_log2_area)
{
    if (n_jobs <= 0) return;
    VVC355_BD_DISPATCH(bd, {
        if (max_log2_area <= 6)      hipLaunchKernelGGL((intra_pred_kernel<BD, 32>), dim3((n_jobs + 7) / 8), dim3(256), 0, (hipStream_t)stream, jobs_dev, n_jobs);
        else if (max_log2_area <= 8) hipLaunchKernelGGL((intra_pred_kernel<BD, 64>), dim3((n_jobs + 3) / 4), dim3(256), 0, (hipStream_t)stream, jobs_dev, n_jobs);
        else                    hipLaunchKernelGGL((intra_pred_kernel<BD, 256>), dim3(n_jobs), dim3(256), 0, (hipStream_t)stream, jobs_dev, n_jobs);
    });
    HIP_CHECK(hipGetLastError());
}

void vvc355_pred_planar(int bd, uint8_t *src, const uint8_t *top, const uint8_t *left, int w, int h, ptrdiff_t stride)
{
    slot_leaf(bd, 0, src, top, left, w, h, stride, 0, 0, 0, 0, 0, 0, 0);
}
void vvc355_pred_dc(int bd, uint8_t *src, const uint8_t *top, const uint8_t *left, int w, int h, ptrdiff_t stride)
{
    slot_leaf(bd, 1, src, top, left, w, h, stride, 0, 0, 0, 0, 0, 0, 0);
}
void vvc355_pred_v(int bd, uint8_t *src, const uint8_t *top, int w, int h, ptrdiff_t stride)
{
    slot_leaf(bd, 2, src, top, nullptr, w, h, stride, 0, 0, 0, 0, 0, 0, 0);
}
void vvc355_pred_h(int bd, uint8_t *src, const uint8_t *left, int w, int h, ptrdiff_t stride)
{
    slot_leaf(bd, 3, src, nullptr, left, w, h, stride, 0, 0, 0, 0, 0, 0, 0);
}
void vvc355_pred_angular_v(int bd, uint8_t *src, const uint8_t *top, const uint8_t *left, int w, int h, ptrdiff_t stride,
                           int c_idx, int mode, int ref_idx, int filter_flag, int need_pdpc)
{
    slot_leaf(bd, 4, src, top, left, w, h, stride, c_idx, mode, ref_idx, filter_flag, need_pdpc, 0, 0);
}
void vvc355_pred_angular_h(int bd, uint8_t *src, const uint8_t *top, const uint8_t *left, int w, int h, ptrdiff_t stride,
                           int c_idx, int mode, int ref_idx, int filter_flag, int need_pdpc)
{
    slot_leaf(bd, 5, src, top, left, w, h, stride, c_idx, mode, ref_idx, filter_flag, need_pdpc, 0, 0);
}
void vvc355_pred_mip(int bd, uint8_t *src, const uint8_t *top, const uint8_t *left, int w, int h, ptrdiff_t stride,
                     int mode_id, int is_transpose)
{
    slot_leaf(bd, 6, src, top, left, w, h, stride, 0, 0, 0, 0, 0, mode_id, is_transpose);
}

// intra_pred with the context flattened; job->plane is a HOST address here, the touched window is staged
void vvc355_intra_pred_flat(int bd, const vvc355_intra_job *job)
{
    check_intra_dims(job->w, job->h);
    check_neighbours("intra_pred", job->x, job->y, job->left_avail, job->top_avail, job->cand_up_left);
    const int px = bd > 8 ? 2 : 1;
    // window of the plane the slot can touch: 4 reference lines up/left, up to cb + block (ISP) or 2x block samples down/right
    const int reach_x = job->w + (job->isp_split && !job->c_idx ? (job->cb_width > job->w ? job->cb_width : job->w) : job->w) + 4;
    const int reach_y = job->h + (job->isp_split && !job->c_idx ? (job->cb_height > job->h ? job->cb_height : job->h) : job->h) + 4;
    const int x0 = job->x - 4 > 0 ? job->x - 4 : 0, y0 = job->y - 4 > 0 ? job->y - 4 : 0;
    const int x1 = job->x + reach_x < job->plane_w ? job->x + reach_x : job->plane_w;
    const int y1 = job->y + reach_y < job->plane_h ? job->y + reach_y : job->plane_h;
    SlotCall call;
    uint8_t *org = (uint8_t *)(uintptr_t)job->plane + (ptrdiff_t)y0 * job->stride + (ptrdiff_t)x0 * px;
    // the window (reference lines, the coding unit's other ISP partitions) is read only; what comes back is the predicted w x h block:
    // RECON of a CTU runs beside INTER of the CTU to its right (vvc_thread.c:156-184), whose samples may lie in the window
    const Staged s = call.rect(org, job->stride, 0, (ptrdiff_t)(x1 - x0) * px, 0, y1 - y0, true, false);
    // (DC and horizontal blocks are written in groups of four samples like the reference's, vvc_intra_template.c:858,884: a 2-wide
    // chroma block's footprint is 4 wide)
    int fw = (!job->is_mip && (job->mode == 1 || job->mode == 18)) ? (job->w + 3) & ~3 : job->w;
    if (job->x + fw > x1) fw = x1 - job->x;
    call.download(org + (ptrdiff_t)(job->y - y0) * job->stride + (ptrdiff_t)(job->x - x0) * px, job->stride,
                  s.dev + (ptrdiff_t)(job->y - y0) * s.pitch + (ptrdiff_t)(job->x - x0) * px, s.pitch, (size_t)fw * px, job->h);
    vvc355_intra_job dj = *job;
    dj.plane = (uint64_t)(s.dev - (ptrdiff_t)y0 * s.pitch - (ptrdiff_t)x0 * px);
    dj.stride = (int32_t)s.pitch;
    int lg = 0;
    while ((1 << lg) < job->w * job->h) lg++;
    vvc355_intra_pred_batch(call.stream(), bd, call.upload(&dj, 1), 1, lg);
}

} // extern "C"

// ================================================================================================ CCLM + LMCS chroma scaling
// intra_cclm_pred (vvc_intra_template.c:352, helpers :29-349) and lmcs_scale_chroma (:431, :390) with the decoder
// context flattened into job descriptors.

namespace vvc355 {

template <int BD, typename PX>
__device__ __forceinline__ int cclm_ds_luma(const vvc355_cclm_job &j, PX luma, int s, int cx, int cy, const StripRef sl = kNoStrip)
{
    const int hs = j.hs, vs = j.vs;
    const int o = (j.y0 + (cy << vs)) * s + j.x0 + (cx << hs);
#define L(dx, dy) luma.ld(o + (dx) + (dy) * s)
    if (!hs && !vs)
        return L(0, 0);
    const int lx = (cx || j.avail_l) ? -1 : 0;
    if (!vs)
        return (L(lx, 0) + 2 * L(0, 0) + L(1, 0) + 2) >> 2;
    if (j.collocated) {
        const int ty = (cy || j.avail_t) ? -1 : 0;
        const int above = (sl.on && !cy && ty) ? sl.at(j.x0 + (cx << hs), -1) : L(0, ty);
        return (L(lx, 0) + above + 4 * L(0, 0) + L(1, 0) + L(0, 1) + 4) >> 3;
    }
    return (L(lx, 0) + L(lx, 1) + 2 * L(0, 0) + 2 * L(0, 1) + L(1, 0) + L(1, 1) + 4) >> 3;
#undef L
}

// the whole slot for one block, executed by the workgroup; prm = six ints in LDS (a[2], b[2], k[2])
// luma / cb / cr = accessors of the planes' sample (0, 0), strides in pixels (the job's addresses and strides are not used)
template <int BD, int NT, typename PX>
__device__ void cclm_body(const vvc355_cclm_job &j, PX luma, int ls, PX cb, PX cr, int cs0, int cs1, int *prm, int tid,
                          const StripRef sl = kNoStrip, const StripRef sb = kNoStrip, const StripRef sr = kNoStrip)
{
    const int hs = j.hs, vs = j.vs;
    const int x = j.x0 >> hs, y = j.y0 >> vs, w = j.width >> hs, h = j.height >> vs;
    const int avail_t = j.avail_t, avail_l = j.avail_l;
    const PX cpl[2] = { cb, cr };
    const int cs[2] = { cs0, cs1 };

    if (!avail_t && !avail_l) {
        const int lw0 = ilog2(w);
        for (int i = tid; i < w * h; i += NT) {
            const int yy = i >> lw0, xx = i & (w - 1);
            cb.st((y + yy) * cs0 + x + xx, 1 << (BD - 1));
            cr.st((y + yy) * cs1 + x + xx, 1 << (BD - 1));
        }
        return;
    }
    // cclm_get_params (:29-349): the (at most four) selected neighbour positions are fetched by four lanes of the group's first
    // wave — luma down-sampled the way the position asks for, the two chroma samples beside it — and the min / max pairing and the
    // line fit run on wave-uniform values (readlane), so nothing is indexed dynamically in registers
    if (tid < 64) {
        int a[2] = { 0, 0 }, b[2] = { 1 << (BD - 1), 1 << (BD - 1) }, k[2] = { 0, 0 };
        const int lt = j.mode == 81;
        const int is4 = !avail_t || !avail_l || !lt;
        int num[2];
        if (lt) { num[0] = avail_t ? w : 0; num[1] = avail_l ? h : 0; }
        else {
            num[0] = (avail_t && j.mode == 83) ? min(w + min(w, h), (int)j.top_avail_c) : 0;
            num[1] = (avail_l && j.mode == 82) ? min(h + min(w, h), (int)j.left_avail_c) : 0;
        }
        if (num[0] || num[1]) {
            int cnt[2], start[2], step[2];
            for (int i = 0; i < 2; i++) {
                start[i] = num[i] >> (2 + is4);
                step[i] = max(1, num[i] >> (1 + is4));
                cnt[i] = min(num[i], (1 + is4) << 1);
            }
            const int total = cnt[0] + cnt[1];
            // lane q holds entry q of the reference's four-entry arrays (two samples: entries 1, 0, 1, 0)
            const int q = tid & 3, si = total == 2 ? (~q & 1) : q;
            const bool from_top = si < cnt[0];
            const int i = from_top ? si : si - cnt[0];
            int s0 = 0, s1 = 0, s2 = 0;
            const int lo = j.y0 * ls + j.x0;
#define LP(off) luma.ld((int)(off))
            if (si < total && from_top) {
                const int p = start[0] + i * step[0];
                if (!hs && !vs)
                    s0 = LP(lo - avail_t * ls + p);
                else {
                    const int xx = p << hs;
                    const int has_left = xx || avail_l;
                    if (vs && !j.ctu_boundary) {
                        const int o = lo - 2 * ls + xx;
                        const int l = has_left ? LP(o - 1) : LP(o);
                        if (j.collocated)
                            s0 = (LP(o - ls) + l + 4 * LP(o) + LP(o + 1) + LP(o + ls) + 4) >> 3;
                        else {
                            const int l1 = has_left ? LP(o - 1 + ls) : LP(o + ls);
                            s0 = (l + l1 + 2 * (LP(o) + LP(o + ls)) + LP(o + 1) + LP(o + 1 + ls) + 4) >> 3;
                        }
                    } else {
                        const int o = lo - ls + xx;
                        if (sl.on) {            // the row above is the CTU above: it lives in the strip
                            const int xa = j.x0 + xx;
                            const int l = has_left ? sl.at(xa - 1, -1) : sl.at(xa, -1);
                            s0 = (l + 2 * sl.at(xa, -1) + sl.at(xa + 1, -1) + 2) >> 2;
                        } else {
                            const int l = has_left ? LP(o - 1) : LP(o);
                            s0 = (l + 2 * LP(o) + LP(o + 1) + 2) >> 2;
                        }
                    }
                }
                s1 = sb.on ? sb.at(x + p, -1) : cpl[0].ld((y - 1) * cs[0] + x + p);
                s2 = sb.on ? sr.at(x + p, -1) : cpl[1].ld((y - 1) * cs[1] + x + p);
            } else if (si < total) {
                const int p = start[1] + i * step[1];
                if (!hs && !vs)
                    s0 = LP(lo - avail_l + p * ls);
                else {
                    const int yy = p << vs;
                    const int o = lo - (1 + hs) * avail_l + yy * ls, l = o - avail_l;
                    if (!vs)
                        s0 = (LP(l) + 2 * LP(o) + LP(o + 1) + 2) >> 2;
                    else if (j.collocated) {
                        const int t = (yy || avail_t) ? ((sl.on && !yy) ? sl.at(j.x0 - (1 + hs) * avail_l, -1) : LP(o - ls)) : LP(o);
                        s0 = (LP(l) + t + 4 * LP(o) + LP(o + 1) + LP(o + ls) + 4) >> 3;
                    } else
                        s0 = (LP(l) + LP(l + ls) + 2 * LP(o) + 2 * LP(o + ls) + LP(o + 1) + LP(o + 1 + ls) + 4) >> 3;
                }
                s1 = cpl[0].ld((y + p) * cs[0] + x - 1);
                s2 = cpl[1].ld((y + p) * cs[1] + x - 1);
            }
#undef LP
            const int l0 = __builtin_amdgcn_readlane(s0, 0), l1 = __builtin_amdgcn_readlane(s0, 1);
            const int l2 = __builtin_amdgcn_readlane(s0, 2), l3 = __builtin_amdgcn_readlane(s0, 3);
            const int lv[4] = { l0, l1, l2, l3 };
            int mn0 = 0, mn1 = 2, mx0 = 1, mx1 = 3, t;
#define SWAP(p, q) do { t = p; p = q; q = t; } while (0)
#define LV(i) ((i) == 0 ? lv[0] : (i) == 1 ? lv[1] : (i) == 2 ? lv[2] : lv[3])
            if (LV(mn0) > LV(mn1)) SWAP(mn0, mn1);
            if (LV(mx0) > LV(mx1)) SWAP(mx0, mx1);
            if (LV(mn0) > LV(mx1)) { SWAP(mn0, mx0); SWAP(mn1, mx1); }
            if (LV(mn1) > LV(mx0)) SWAP(mn1, mx0);
#undef LV
#undef SWAP
            const int sv[3] = { s0, s1, s2 };
            int vmax[3], vmin[3];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                vmax[c] = (__builtin_amdgcn_readlane(sv[c], mx0) + __builtin_amdgcn_readlane(sv[c], mx1) + 1) >> 1;
                vmin[c] = (__builtin_amdgcn_readlane(sv[c], mn0) + __builtin_amdgcn_readlane(sv[c], mn1) + 1) >> 1;
            }
            const int diff = vmax[0] - vmin[0];
#pragma unroll
            for (int i2 = 0; i2 < 2; i2++) {
                if (!diff) { a[i2] = k[i2] = 0; b[i2] = vmin[i2 + 1]; continue; }
                // div_sig[] = { 0, 7, 6, 5, 5, 4, 4, 3, 3, 2, 2, 1, 1, 1, 1, 0 } as nibbles
                const int diffc = vmax[i2 + 1] - vmin[i2 + 1];
                int xl = ilog2(diff);
                const int norm = ((diff << 4) >> xl) & 15;
                xl += norm ? 1 : 0;
                const int yl = abs(diffc) > 0 ? ilog2(abs(diffc)) + 1 : 0;
                const int v = cclm_div_sig(norm) | 8;
                a[i2] = (diffc * v + ((1 << yl) >> 1)) >> yl;
                k[i2] = max(1, 3 + xl - yl);
                if (3 + xl - yl < 1)
                    a[i2] = sign_of(a[i2]) * 15;
                b[i2] = vmin[i2 + 1] - ((a[i2] * vmin[0]) >> k[i2]);
            }
        }
        if (tid == 0) {
            prm[0] = a[0]; prm[1] = a[1]; prm[2] = b[0]; prm[3] = b[1]; prm[4] = k[0]; prm[5] = k[1];
        }
    }
    group_sync<NT>();
    const int lw = ilog2(w);                    // block sides are powers of two
    if (hs == 1 && vs == 1 && w >= 4) {
        // 4:2:0: four chroma samples of a row per lane.  Their down-sampling windows (cclm_ds_luma) cover luma columns -1 .. 7 of the
        // row pair: one single load and two four-sample loads per luma row instead of five or six loads per chroma sample.
        const int lq = lw - 2;
        const int a0 = prm[0], a1 = prm[1], b0 = prm[2], b1 = prm[3], k0 = prm[4], k1 = prm[5];
        for (int q = tid; q < (h << lq); q += NT) {
            const int yy = q >> lq, xx0 = (q & ((1 << lq) - 1)) << 2;
            const int o = (j.y0 + 2 * yy) * ls + j.x0 + 2 * xx0;       // luma sample (2 xx0, 2 yy) of the block
            int r0[9], r1[9], v[4];                                    // columns -1 .. 7
            luma.ld4(o, r0 + 1); luma.ld4(o + 4, r0 + 5);
            luma.ld4(o + ls, r1 + 1); luma.ld4(o + ls + 4, r1 + 5);
            const bool has_left = xx0 || avail_l;
            r0[0] = has_left ? luma.ld(o - 1) : r0[1];
            if (j.collocated) {
                const int ty = (yy || avail_t) ? -1 : 0;
                int up[8];
                if (ty == 0) {
#pragma unroll
                    for (int e = 0; e < 8; e++) up[e] = r0[1 + e];
                } else if (sl.on && !yy) {
#pragma unroll
                    for (int e = 0; e < 4; e++) up[2 * e] = sl.at(j.x0 + 2 * xx0 + 2 * e, -1);
                } else {
                    luma.ld4(o - ls, up); luma.ld4(o - ls + 4, up + 4);
                }
#pragma unroll
                for (int e = 0; e < 4; e++)
                    v[e] = (r0[2 * e] + up[2 * e] + 4 * r0[2 * e + 1] + r0[2 * e + 2] + r1[2 * e + 1] + 4) >> 3;
            } else {
                r1[0] = has_left ? luma.ld(o + ls - 1) : r1[1];
#pragma unroll
                for (int e = 0; e < 4; e++)
                    v[e] = (r0[2 * e] + r1[2 * e] + 2 * (r0[2 * e + 1] + r1[2 * e + 1]) + r0[2 * e + 2] + r1[2 * e + 2] + 4) >> 3;
            }
            int u[4], t[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                u[e] = clip_px<BD>(((v[e] * a0) >> k0) + b0);
                t[e] = clip_px<BD>(((v[e] * a1) >> k1) + b1);
            }
            cb.st4((y + yy) * cs0 + x + xx0, u[0], u[1], u[2], u[3]);
            cr.st4((y + yy) * cs1 + x + xx0, t[0], t[1], t[2], t[3]);
        }
        return;
    }
    for (int i = tid; i < w * h; i += NT) {
        const int yy = i >> lw, xx = i & (w - 1);
        const int dsy = cclm_ds_luma<BD>(j, luma, ls, xx, yy, sl);
#pragma unroll
        for (int c = 0; c < 2; c++)
            cpl[c].st((y + yy) * cs[c] + x + xx, clip_px<BD>(((dsy * prm[c]) >> prm[4 + c]) + prm[2 + c]));
    }
}

template <int BD>
__global__ __launch_bounds__(256) void cclm_kernel(const vvc355_cclm_job *__restrict__ jobs)
{
    __shared__ int prm[6];
    const vvc355_cclm_job j = load_uniform(jobs + (blockIdx.x));
    constexpr int PXS = (int)sizeof(typename Px<BD>::type);
    cclm_body<BD, 256>(j, GPix<BD>{ (uint8_t *)j.luma }, j.luma_stride / PXS, GPix<BD>{ (uint8_t *)j.cb }, GPix<BD>{ (uint8_t *)j.cr },
                       j.cb_stride / PXS, j.cr_stride / PXS, prm, threadIdx.x);
}

template <int BD>
__device__ int lmcs_chroma_scale(const vvc355_lmcs_scale_job &j)
{
    const uint8_t *luma = (const uint8_t *)j.luma;
    const ptrdiff_t s = j.luma_stride / (ptrdiff_t)sizeof(typename Px<BD>::type);
    const int size = j.size_y, x = j.x_vpdu, y = j.y_vpdu;
    int cnt = 0, sum = 0;
    if (j.avail_l) {
        const int n = min(j.pic_h - y, size);
        for (int i = 0; i < n; i++) sum += ld_px<BD>(luma, (ptrdiff_t)(y + i) * s + x - 1);
        sum += ld_px<BD>(luma, (ptrdiff_t)(y + n - 1) * s + x - 1) * (size - n);
        cnt = size;
    }
    if (j.avail_t) {
        const int n = min(j.pic_w - x, size);
        for (int i = 0; i < n; i++) sum += ld_px<BD>(luma, (ptrdiff_t)(y - 1) * s + x + i);
        sum += ld_px<BD>(luma, (ptrdiff_t)(y - 1) * s + x + n - 1) * (size - n);
        cnt += size;
    }
    const int avg = cnt ? (sum + (cnt >> 1)) >> ilog2(cnt) : 1 << (BD - 1);
    int i;
    for (i = j.min_bin_idx; i <= j.max_bin_idx; i++)
        if (avg < j.pivot[i + 1])
            break;
    return j.chroma_scale_coeff[min(i, 15)];
}

// one workgroup: lane 0 derives the scale, then all lanes scale the residual block
template <int BD>
__global__ __launch_bounds__(256) void lmcs_scale_kernel(const vvc355_lmcs_scale_job *job, int *dst, const int *coeff, int n)
{
    __shared__ int scale;
    if (threadIdx.x == 0)
        scale = lmcs_chroma_scale<BD>(*job);
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int c = clip_intp2(coeff[i], BD);
        dst[i] = c > 0 ? (c * scale + (1 << 10)) >> 11 : -((-c * scale + (1 << 10)) >> 11);
    }
}

// lmcs_derive_chroma_scale (vvc_intra_template.c:390-429) of the 64x64 unit at (x, y) from a luma plane in HBM, one wave: lanes
// 0 .. size - 1 fetch one left and one upper neighbour each (lmcs_sum_samples' replication beyond the picture = a clamped index)
// The bin of lmcs_derive_chroma_scale's search (vvc_intra_template.c:421-426: from min_bin_idx up while avg >= pivot[bin + 1], at most to
// max_bin_idx + 1) without its chain of dependent loads: lane l compares against pivot[l + 1], the first lane at or after min_bin_idx that
// would stop the loop is the bin.  Then chroma_scale_coeff[min(bin, 15)].  Wave-uniform arguments and result.
__device__ __forceinline__ int lmcs_scale_of_avg(const vvc355_lmcs_model *model, int avg, int lane)
{
    const int first = gld<uint8_t>(&model->min_bin_idx), last = gld<uint8_t>(&model->max_bin_idx);
    const int pv = lane < 16 ? (int)gld<uint16_t>(&model->pivot[lane + 1]) : 0;
    const bool stop = lane >= first && (lane > last || avg < pv);          // lane 16 stops at the latest (last <= 15)
    const int bin = __builtin_ctzll(__builtin_amdgcn_ballot_w64(stop));
    return __builtin_amdgcn_readfirstlane((int)gld<uint16_t>(&model->chroma_scale_coeff[min(bin, 15)]));
}

template <int BD>
__device__ __forceinline__ int lmcs_scale_from_plane(const vvc355_lmcs_model *model, const uint8_t *luma, int ls, int x, int y, int size, bool avail_l, bool avail_t,
                                                     int pic_w, int pic_h, int lane)
{
    int v = 0;
    if (lane < size) {
        if (avail_l)
            v += ld_px<BD>(luma, (ptrdiff_t)(y + min(lane, min(pic_h - y, size) - 1)) * ls + x - 1);
        if (avail_t)
            v += ld_px<BD>(luma, (ptrdiff_t)(y - 1) * ls + x + min(lane, min(pic_w - x, size) - 1));
    }
#pragma unroll
    for (int sft = 32; sft; sft >>= 1)
        v += __shfl_xor(v, sft, 64);
    const int cnt = (avail_l ? size : 0) + (avail_t ? size : 0);
    const int avg = cnt ? (v + (cnt >> 1)) >> ilog2i(cnt) : 1 << (BD - 1);
    return lmcs_scale_of_avg(model, avg, lane);
}

// the tail of itransform for one block in HBM, one wave: joint sign / shift (pred_residual_joint), lmcs_scale_chroma when joint bit 3 is
// set, add_residual; four samples of a row per lane and step (one at a time for blocks narrower than four)
template <int BD, int NL = 64>
__device__ __forceinline__ void resid_block_add(uint8_t *dst, int dst_stride, const int *res, int w, int h, int joint, int scale, int lane)
{
    auto resid_of = [&](int r) {
        if (joint & 1)
            r = (r * ((joint & 2) ? -1 : 1)) >> ((joint >> 2) & 1);
        if (joint & 8) {
            const int c = clip_intp2(r, BD);
            r = c > 0 ? (c * scale + (1 << 10)) >> 11 : -((-c * scale + (1 << 10)) >> 11);
        }
        return r;
    };
    const int n = w * h, lw = ilog2i(w);
    if (w < 4) {
        for (int i = lane; i < n; i += NL) {
            uint8_t *row = dst + row_off(i >> lw, dst_stride);
            st_px<BD>(row, i & (w - 1), clip_px<BD>(ld_px<BD>(row, i & (w - 1)) + resid_of(gld<int>(res + i))));
        }
        return;
    }
    for (int i = lane * 4; i < n; i += NL * 4) {
        const int4 r4 = gld<int4>(res + i);
        uint8_t *row = dst + row_off(i >> lw, dst_stride);
        const int xo = i & (w - 1);
        const int r[4] = { r4.x, r4.y, r4.z, r4.w };
#pragma unroll
        for (int q = 0; q < 4; q++)
            st_px<BD>(row, xo + q, clip_px<BD>(ld_px<BD>(row, xo + q) + resid_of(r[q])));
    }
}

// every 64x64 unit's scale: one wave per unit (vvc355_lmcs_vpdu_scale_pass)
template <int BD>
__global__ __launch_bounds__(256) void lmcs_vpdu_scale_kernel(const vvc355_lmcs_scale_frame *__restrict__ fp)
{
    using px_t = typename Px<BD>::type;
    const vvc355_lmcs_scale_frame f = load_uniform(fp);
    const int size = f.size_y, ux = (f.width + size - 1) / size, uy = (f.height + size - 1) / size;
    const int u = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (u >= ux * uy)
        return;
    const int vy = u / ux, vx = u - vy * ux, x = vx * size, y = vy * size;
    // ff_vvc_get_left / top_available(lc, x, y, 1, 0) for a unit's first sample: inside its CTU the neighbour unit precedes it; at the CTU's
    // edge the neighbour CTU must exist in the same tile (and, above, the same slice): ff_vvc_decode_neighbour, vvc_ctu.c:2468-2495
    const int ctb = 1 << f.ctb_log2, rx = x >> f.ctb_log2, ry = y >> f.ctb_log2, rs = ry * f.ctb_width + rx;
    const int16_t *slice_idx = (const int16_t *)f.slice_idx, *col_bd = (const int16_t *)f.ctb_to_col_bd, *row_bd = (const int16_t *)f.ctb_to_row_bd;
    bool avail_l = true, avail_t = true;
    if ((x & (ctb - 1)) == 0)
        avail_l = rx > 0 && gld<int16_t>(col_bd + rx) == gld<int16_t>(col_bd + rx - 1);
    if ((y & (ctb - 1)) == 0)
        avail_t = ry > 0 && gld<int16_t>(row_bd + ry) == gld<int16_t>(row_bd + ry - 1) && gld<int16_t>(slice_idx + rs) == gld<int16_t>(slice_idx + rs - f.ctb_width);
    const int s = lmcs_scale_from_plane<BD>((const vvc355_lmcs_model *)f.model, (const uint8_t *)f.luma, f.luma_stride / (int)sizeof(px_t), x, y, size, avail_l, avail_t,
                                            f.width, f.height, lane);
    if (lane == 0)
        gst<int16_t>((int16_t *)f.scale + u, (int16_t)s);
}

// Sixteen lanes per chroma block, four blocks per wave (a typical block of an inter coding unit is 8x8: one step of four samples per lane;
// a wave per block left three quarters of the lanes idle and the launch was bound by its wave count — one slot per transform block of the
// picture, most of them empty).  The block's scale comes from the picture's table (joint bit 4); a job that wants it derived from the luma
// plane (bit 3 alone) gets the whole wave for that, one such job after the other, before the groups add their blocks.
template <int BD>
__global__ __launch_bounds__(256) void lmcs_chroma_resid_kernel(const vvc355_lmcs_resid_job *__restrict__ jobs, int n_jobs, const vvc355_lmcs_model *__restrict__ model)
{
    using px_t = typename Px<BD>::type;
    const int lane = threadIdx.x & 63, g = lane >> 4, l16 = lane & 15;
    const int first = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;             // the wave's four slots
    if (first >= n_jobs)
        return;
    const int ji = first + g;
    vvc355_lmcs_resid_job j = {};
    if (ji < n_jobs) {
        // 56 bytes: three 16-byte pieces and one of 8 (the struct is 8-byte aligned; the lanes of a group read the same addresses)
        const uint2 *p = (const uint2 *)(jobs + ji);
        uint2 q[7];
#pragma unroll
        for (int i = 0; i < 7; i++) q[i] = gld<uint2>(p + i);
        __builtin_memcpy(&j, q, sizeof(j));
    }
    int scale = 0;
    const bool derive = j.w > 0 && (j.joint & 8) && !(j.joint & 16);
    unsigned long long todo = __builtin_amdgcn_ballot_w64(derive && l16 == 0);
    while (todo) {                                   // wave-uniform: the group's parameters through its first lane
        const int src = __builtin_ctzll(todo);
        todo &= todo - 1;
        auto u32 = [&](uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)v, src); };
        const uint64_t luma = (uint64_t)u32((uint32_t)j.luma) | ((uint64_t)u32((uint32_t)(j.luma >> 32)) << 32);
        const int s_ = lmcs_scale_from_plane<BD>(model, (const uint8_t *)luma, (int)u32((uint32_t)j.luma_stride) / (int)sizeof(px_t), (int16_t)u32((uint16_t)j.x_vpdu),
                                                 (int16_t)u32((uint16_t)j.y_vpdu), (int16_t)u32((uint16_t)j.size_y), u32(j.avail_l) != 0, u32(j.avail_t) != 0,
                                                 (int16_t)u32((uint16_t)j.pic_w), (int16_t)u32((uint16_t)j.pic_h), lane);
        if (g == (src >> 4))
            scale = s_;
    }
    if (j.w <= 0)
        return;                 // an empty slot of a job array the transform-block builder wrote (vvc355_itx_frame.resid_jobs), or past the end
    if (j.joint & 16)
        scale = (int)gld<int16_t>((const int16_t *)j.luma);
    resid_block_add<BD, 16>((uint8_t *)j.dst, j.dst_stride, (const int *)j.resid, j.w, j.h, j.joint, scale, l16);
}

} // namespace vvc355

extern "C" {

void vvc355_lmcs_vpdu_scale_pass(void *stream, int bd, const vvc355_lmcs_scale_frame *frame_dev, const vvc355_lmcs_scale_frame *frame_host)
{
    const int size = frame_host->size_y;
    if (size != 32 && size != 64) {
        fprintf(stderr, "vvc_mi355: vvc355_lmcs_vpdu_scale_pass: unit size %d (min(CtbSizeY, 64) is 32 or 64)\n", size);
        abort();
    }
    const int n = ((frame_host->width + size - 1) / size) * ((frame_host->height + size - 1) / size);
    if (n <= 0) return;
    VVC355_BD_DISPATCH(bd, hipLaunchKernelGGL((vvc355::lmcs_vpdu_scale_kernel<BD>), dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, frame_dev));
    HIP_CHECK(hipGetLastError());
}

void vvc355_lmcs_chroma_resid_batch(void *stream, int bd, const vvc355_lmcs_resid_job *jobs_dev, int n_jobs, const vvc355_lmcs_model *model_dev)
{
    if (n_jobs <= 0) return;
    VVC355_BD_DISPATCH(bd, hipLaunchKernelGGL((vvc355::lmcs_chroma_resid_kernel<BD>), dim3((n_jobs + 15) / 16), dim3(256), 0, (hipStream_t)stream, jobs_dev, n_jobs, model_dev));
    HIP_CHECK(hipGetLastError());
}

void vvc355_cclm_batch(void *stream, int bd, const vvc355_cclm_job *jobs_dev, int n_jobs)
{
    if (n_jobs <= 0) return;
    VVC355_BD_DISPATCH(bd, hipLaunchKernelGGL((vvc355::cclm_kernel<BD>), dim3(n_jobs), dim3(256), 0, (hipStream_t)stream, jobs_dev));
    HIP_CHECK(hipGetLastError());
}

// job->luma/cb/cr are HOST addresses of sample (0,0) of each plane; pic_* bound what is staged
void vvc355_intra_cclm_pred_flat(int bd, const vvc355_cclm_job *job, int pic_w, int pic_h)
{
    using namespace vvc355;
    check_neighbours("intra_cclm_pred", job->x0, job->y0, job->left_avail_c || job->avail_l, job->top_avail_c || job->avail_t, 0);
    const int px = bd > 8 ? 2 : 1;
    const int hs = job->hs, vs = job->vs;
    // luma window: block + 2 rows above, 3 columns left, 1 extra row/column, and the T/L extension (2x the block)
    const int lx0 = job->x0 - 4 > 0 ? job->x0 - 4 : 0, ly0 = job->y0 - 4 > 0 ? job->y0 - 4 : 0;
    const int lx1 = job->x0 + 2 * job->width + 4 < pic_w ? job->x0 + 2 * job->width + 4 : pic_w;
    const int ly1 = job->y0 + 2 * job->height + 4 < pic_h ? job->y0 + 2 * job->height + 4 : pic_h;
    const int cw = pic_w >> hs, ch = pic_h >> vs;
    const int cx0 = lx0 >> hs, cy0 = ly0 >> vs;
    const int cx1 = ((lx1 + (1 << hs) - 1) >> hs) < cw ? ((lx1 + (1 << hs) - 1) >> hs) : cw;
    const int cy1 = ((ly1 + (1 << vs) - 1) >> vs) < ch ? ((ly1 + (1 << vs) - 1) >> vs) : ch;
    SlotCall call;
    vvc355_cclm_job dj = *job;
    const Staged l = call.rect((uint8_t *)(uintptr_t)job->luma + (ptrdiff_t)ly0 * job->luma_stride + (ptrdiff_t)lx0 * px,
                               job->luma_stride, 0, (ptrdiff_t)(lx1 - lx0) * px, 0, ly1 - ly0, true, false);
    dj.luma = (uint64_t)(l.dev - (ptrdiff_t)ly0 * l.pitch - (ptrdiff_t)lx0 * px); dj.luma_stride = (int32_t)l.pitch;
    const Staged b = call.rect((uint8_t *)(uintptr_t)job->cb + (ptrdiff_t)cy0 * job->cb_stride + (ptrdiff_t)cx0 * px,
                               job->cb_stride, 0, (ptrdiff_t)(cx1 - cx0) * px, 0, cy1 - cy0, true, false);
    dj.cb = (uint64_t)(b.dev - (ptrdiff_t)cy0 * b.pitch - (ptrdiff_t)cx0 * px); dj.cb_stride = (int32_t)b.pitch;
    const Staged r = call.rect((uint8_t *)(uintptr_t)job->cr + (ptrdiff_t)cy0 * job->cr_stride + (ptrdiff_t)cx0 * px,
                               job->cr_stride, 0, (ptrdiff_t)(cx1 - cx0) * px, 0, cy1 - cy0, true, false);
    dj.cr = (uint64_t)(r.dev - (ptrdiff_t)cy0 * r.pitch - (ptrdiff_t)cx0 * px); dj.cr_stride = (int32_t)r.pitch;
    // only the predicted chroma blocks come back (the windows hold neighbours that other decoder threads may be writing)
    {
        const int bx = job->x0 >> hs, by = job->y0 >> vs, bw = job->width >> hs, bh = job->height >> vs;
        call.download((uint8_t *)(uintptr_t)job->cb + (ptrdiff_t)by * job->cb_stride + (ptrdiff_t)bx * px, job->cb_stride,
                      b.dev + (ptrdiff_t)(by - cy0) * b.pitch + (ptrdiff_t)(bx - cx0) * px, b.pitch, (size_t)bw * px, bh);
        call.download((uint8_t *)(uintptr_t)job->cr + (ptrdiff_t)by * job->cr_stride + (ptrdiff_t)bx * px, job->cr_stride,
                      r.dev + (ptrdiff_t)(by - cy0) * r.pitch + (ptrdiff_t)(bx - cx0) * px, r.pitch, (size_t)bw * px, bh);
    }
    vvc355_cclm_batch(call.stream(), bd, call.upload(&dj, 1), 1);
}

// job->luma is a HOST address; dst/coeff are host int arrays of width*height
void vvc355_lmcs_scale_chroma_flat(int bd, const vvc355_lmcs_scale_job *job, int *dst, const int *coeff, int width, int height)
{
    using namespace vvc355;
    if (width <= 0 || height <= 0) return;
    check_neighbours("lmcs_scale_chroma", job->x_vpdu, job->y_vpdu, job->avail_l, job->avail_t, 0);
    const int px = bd > 8 ? 2 : 1, n = width * height;
    const int x = job->x_vpdu, y = job->y_vpdu;
    const int x0 = x > 0 ? x - 1 : 0, y0 = y > 0 ? y - 1 : 0;
    const int x1 = x + job->size_y < job->pic_w ? x + job->size_y : job->pic_w;
    const int y1 = y + job->size_y < job->pic_h ? y + job->size_y : job->pic_h;
    SlotCall call;
    vvc355_lmcs_scale_job dj = *job;
    const Staged l = call.rect((uint8_t *)(uintptr_t)job->luma + (ptrdiff_t)y0 * job->luma_stride + (ptrdiff_t)x0 * px,
                               job->luma_stride, 0, (ptrdiff_t)(x1 - x0) * px, 0, y1 - y0, true, false);
    dj.luma = (uint64_t)(l.dev - (ptrdiff_t)y0 * l.pitch - (ptrdiff_t)x0 * px); dj.luma_stride = (int32_t)l.pitch;
    int *d_dst = (int *)call.linear(dst, (size_t)n * sizeof(int), false, true);
    const int *d_coeff = (const int *)call.linear(coeff, (size_t)n * sizeof(int), true, false);
    const vvc355_lmcs_scale_job *jd = call.upload(&dj, 1);
    VVC355_BD_DISPATCH(bd, hipLaunchKernelGGL((lmcs_scale_kernel<BD>), dim3(1), dim3(256), 0, call.stream(), jd, d_dst, d_coeff, n));
    HIP_CHECK(hipGetLastError());
}

} // extern "C"

// ================================================================================================ RECON stage driver
// ff_vvc_reconstruct (vvc_intra.c:498-527) for a whole picture: one workgroup per CTU walks the CTU's command list in decoding
// order (intra prediction reading what earlier blocks wrote, then the transform unit's residual), CTUs are released in wavefront
// order.  See include/vvc_mi355.h (vvc355_recon_frame) for the command set and for what stays in the batched transform stage.
//
// Scheduling: workgroups (one wave each) take tickets (one atomic per workgroup) and process the CTUs that have commands in raster order of their
// tickets, so every CTU a workgroup waits for belongs to a workgroup that is already running or done: no deadlock whatever the
// dispatch order.  A CTU waits for its left, upper-left, upper and upper-right neighbours (those with commands; the others carry
// only inter prediction + residuals finished by earlier launches).  Hand-off between workgroups: all stores of the CTU, every
// wave's s_waitcnt vmcnt(0), workgroup barrier, then one lane's agent-scope release and the flag store; the consumer polls with
// relaxed agent-scope loads, one agent-scope acquire, s_waitcnt, workgroup barrier, then plain loads (MI355X_MICROARCH.md,
// "Valid forms").  Inside a workgroup a block's stores reach the later blocks' loads through the workgroup barrier (same CU, same L1).

namespace vvc355 {

struct ReconLds {
    uint16_t arr[3][4][kEdgeLen];     // edge arrays and scratch: luma wave, chroma wave (third set: Cr when Cb and Cr are predicted together)
    int scratch[3][16];
    int prm[2][8];
    uint32_t rmap[2][2][2][32];       // per wave (the availability pre-pass of recon_one_ctu): reconstructed areas of this CTU per channel type as
                                      // bitmaps of 4x4-luma-sample units: [0] bit b of word u = unit (b, u), [1] its transpose (see recon_top_available)
    uint32_t cmdbuf[2][64 * 10];      // per wave: the window of 64 commands it is walking
    int luma_done;                    // commands the luma wave has passed (the chroma wave waits on it before CCLM)
    int bc[8];
    IntraTabsLds tabs;                // the predictors' constant tables (filled while the CTU waits for its neighbours)
};
struct ReconCtx { int ctb_up, ctb_left, ctb_up_left, end_of_tiles_x, ox, oy; };       // ox, oy: the CTU's origin (luma samples)

// The reference keeps the CTU's reconstructed areas as a list (add_reconstructed_area, vvc_intra.c:188-206, restarted per CTU :508)
// and answers "how many samples above / left of this block are reconstructed" by walking it (get_reconstructed_area :574-589:
// newest area first, stop at an area wholly up-left of the point).  For the areas a decoder produces — disjoint blocks of one
// partitioning, in coding order, positions and sizes multiples of four luma samples except the 1- and 2-row intra sub-partitions,
// which only ever ask about their own earlier rows or about finished coding units — the walk's answer is the length of the run of
// covered 4x4 units starting at the block: here one LDS word and a count of trailing ones.  (An area up-left of a point is never
// coded after an area containing it, so the walk's early stop never hides a hit.)  The oracle keeps the literal list walk.
__device__ __forceinline__ int recon_run(uint32_t line, int from)
{
    const uint32_t m = (uint32_t)__builtin_amdgcn_readfirstlane((int)line) >> from;
    return m == ~0u ? 32 : __builtin_ctz(~m);
}
// ff_vvc_get_top_available (vvc_intra.c:591-620), one wave (uniform arguments, uniform result)
typedef uint32_t ReconMaps[2][2][32];
__device__ int recon_top_available(const vvc355_recon_frame &f, const ReconCtx &cx, const ReconMaps &rmap, int cu_x0, int x, int y, int target, int c_idx)
{
    const int hs = c_idx ? f.hs : 0, vs = c_idx ? f.vs : 0;
    const int end_of_ctb_x = ((cu_x0 >> f.ctb_log2) + 1) << f.ctb_log2;
    const int y0b = y & ((1 << (f.ctb_log2 - vs)) - 1);
    const int max_x = min(f.width, end_of_ctb_x) >> hs;
    if (!y0b) {
        if (!cx.ctb_up)
            return 0;
        target = min(target, (cx.end_of_tiles_x >> hs) - x);
        if (f.wpp)
            target = min(target, (end_of_ctb_x >> hs) - x);
        return target;
    }
    target = max(0, min(target, max_x - x));
    const int ux = ((x << hs) - cx.ox) >> 2, uy = (((y - 1) << vs) - cx.oy) >> 2;
    const int end = (cx.ox + ((ux + recon_run(rmap[c_idx > 0][0][uy], ux)) << 2)) >> hs;
    return max(0, min(target, end - x));
}
// ff_vvc_get_left_available (vvc_intra.c:622-648), one wave
__device__ int recon_left_available(const vvc355_recon_frame &f, const ReconCtx &cx, const ReconMaps &rmap, int cu_y0, int x, int y, int target, int c_idx)
{
    const int hs = c_idx ? f.hs : 0, vs = c_idx ? f.vs : 0;
    const int x0b = x & ((1 << (f.ctb_log2 - hs)) - 1);
    const int end_of_ctb_y = ((cu_y0 >> f.ctb_log2) + 1) << f.ctb_log2;
    const int max_y = min(f.height, end_of_ctb_y) >> vs;
    if (!x0b && !cx.ctb_left)
        return 0;
    target = max(0, min(target, max_y - y));
    if (!x0b)
        return target;
    const int ux = (((x - 1) << hs) - cx.ox) >> 2, uy = ((y << vs) - cx.oy) >> 2;
    const int end = (cx.oy + ((uy + recon_run(rmap[c_idx > 0][1][ux], uy)) << 2)) >> vs;
    return max(0, min(target, end - y));
}
// ff_vvc_wide_angle_mode_mapping (vvc_intra.c:693-714)
__host__ __device__ inline int wide_angle_mode(int isp_split, int c_idx, int tb_w, int tb_h, int cb_w, int cb_h, int mode)
{
    const int nw = (!isp_split || c_idx) ? tb_w : cb_w, nh = (!isp_split || c_idx) ? tb_h : cb_h;
    const int d = ilog2i(nw) - ilog2i(nh), wh_ratio = d < 0 ? -d : d;
    const int mx = wh_ratio > 1 ? 8 + 2 * wh_ratio : 8, mn = wh_ratio > 1 ? 60 - 2 * wh_ratio : 60;
    if (nw > nh && mode >= 2 && mode < mx)
        return mode + 65;
    if (nh > nw && mode <= 66 && mode > mn)
        return mode - 67;
    return mode;
}


static constexpr int kReconFlags = 16;
#define VVC355_LDS __attribute__((address_space(3)))
__device__ __forceinline__ void recon_luma_done_set(ReconLds &L, int v)
{
    __hip_atomic_store((VVC355_LDS int *)&L.luma_done, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ int recon_luma_done_get(ReconLds &L)
{
    return __hip_atomic_load((VVC355_LDS int *)&L.luma_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}        // state[0] = ticket counter, state[kReconFlags + rs] = CTU rs done, state[kReconFlags + n_ctus + rs] = its luma is in the planes

// one wave per CTU and channel type: a CTU's blocks are a dependent chain (each reads what the previous ones wrote), so more lanes
// per block would only add workgroup barriers to every link.  TILE (4:2:0, CTUs up to 128x128): the CTU's three component blocks live in LDS for the whole
// walk, together with the four rows above (reaching one CTU to the right: above-right references) and the four columns to the
// left that reference-line selection can address — loaded once after the neighbours' flags, written back once before this CTU's
// flag.  Every link of the chain then costs LDS latency instead of an HBM / L2 round trip.  Other chroma formats walk on the
// planes in HBM (the wave's own stores are made visible to its later loads by s_waitcnt vmcnt(0)).
__device__ __forceinline__ void recon_sync_mem()
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// CTU body tiles: rows 0 .. ctb - 1, columns -kApr .. ctb - 1 (the left apron holds the left neighbour's last columns); strips: the
// kApr rows above, columns -kApr .. 2 ctb - 1.  Luma 128 x 136 + 4 x 264, each chroma 64 x 68 + 4 x 136: 56.5 KB with the walk's own
// state below it — two CTUs per CU.
static constexpr int kBodyLumaP = kApr + 128 + 4, kBodyChromaP = kApr + 64;                    // 136, 68
static constexpr int kStripLumaP = kApr + 2 * 128 + 4, kStripChromaP = kApr + 2 * 64 + 4;      // 264, 136
static constexpr int kBodyLuma = 128 * kBodyLumaP, kBodyChroma = 64 * kBodyChromaP;
static constexpr int kStripLuma = kApr * kStripLumaP, kStripChroma = kApr * kStripChromaP;
static constexpr int kTileSamples = kBodyLuma + 2 * kBodyChroma + kStripLuma + 2 * kStripChroma;

// four samples of a plane row as the tile holds them (uint16 each); x may hang over the picture's left / right edge (samples there
// are never referenced: the availability process stops at the picture; they are clamped so that no load leaves the plane)
template <int BD>
__device__ __forceinline__ uint2 recon_chunk_load(const uint8_t *plane, int stride, int x, int y, int pic_w, int pic_h)
{
    using px_t = typename Px<BD>::type;
    y = clip3(y, 0, pic_h - 1);
    if (x >= 0 && x + 3 < pic_w) {
        const uint8_t *p = plane + row_off(y, stride) + x * (int)sizeof(px_t);
        if (BD > 8)
            return gld<uint2>(p);
        const uint32_t q = gld<uint32_t>(p);
        return make_uint2(__builtin_amdgcn_perm(0, q, 0x0c010c00u), __builtin_amdgcn_perm(0, q, 0x0c030c02u));
    }
    uint32_t e[4];
#pragma unroll
    for (int i = 0; i < 4; i++)
        e[i] = (uint32_t)ld_px<BD>(plane + row_off(y, stride), clip3(x + i, 0, pic_w - 1));
    return make_uint2(e[0] | (e[1] << 16), e[2] | (e[3] << 16));
}
// The CTU's own cw x ch samples -> the body tile (columns kApr ..).  Nothing here depends on the neighbours, so this runs BEFORE the
// wait on their flags.  LK = log2 of the chunk slots per row (5: luma, 4: chroma): 64 >> LK rows per step, eight steps' loads in flight.
template <int BD, int LK>
__device__ void recon_body_load(uint16_t *tile, int pitch, const uint8_t *plane, int stride, int ox, int oy, int cw, int ch, int pic_w, int pic_h, int lane)
{
    constexpr int RPS = 64 >> LK;
    const int k = lane & ((1 << LK) - 1), r0 = lane >> LK;
    if (4 * k >= cw)
        return;
    uint16_t *d = tile + r0 * pitch + kApr + 4 * k;
    for (int r = r0; r < ch; r += RPS * 8) {
        uint2 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (r + u * RPS < ch)
                v[u] = recon_chunk_load<BD>(plane, stride, ox + 4 * k, oy + r + u * RPS, pic_w, pic_h);
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (r + u * RPS < ch)
                *(uint2 *)(d + (r - r0 + u * RPS) * pitch) = v[u];
    }
}
// What the neighbours wrote: the kApr columns left of the CTU (-> the body tile's apron) and the kApr rows above it, reaching one
// CTU to the right (-> the strip); after the wait.  All of a lane's loads are issued before its first LDS store.
template <int BD>
__device__ void recon_edges_load(uint16_t *tile, int pitch, uint16_t *strip, int spitch, const uint8_t *plane, int stride, int ox, int oy, int ctbc, int ch,
                                 int pic_w, int pic_h, int lane)
{
    uint2 va[2], vs[5];
    const int n4 = (kApr + 2 * ctbc) / 4;            // chunks per strip row (at most 65)
    if (ox > 0) {
#pragma unroll
        for (int u = 0; u < 2; u++)
            if (lane + 64 * u < ch)
                va[u] = recon_chunk_load<BD>(plane, stride, ox - kApr, oy + lane + 64 * u, pic_w, pic_h);
    }
    if (oy > 0) {
#pragma unroll
        for (int u = 0; u < 5; u++) {
            const int i = lane + 64 * u;
            if (i < kApr * n4) {
                const int r = (i >= n4) + (i >= 2 * n4) + (i >= 3 * n4), k = i - r * n4;
                vs[u] = recon_chunk_load<BD>(plane, stride, ox - kApr + 4 * k, oy - kApr + r, pic_w, pic_h);
            }
        }
    }
    if (ox > 0) {
#pragma unroll
        for (int u = 0; u < 2; u++)
            if (lane + 64 * u < ch)
                *(uint2 *)(tile + (lane + 64 * u) * pitch) = va[u];
    }
    if (oy > 0) {
#pragma unroll
        for (int u = 0; u < 5; u++) {
            const int i = lane + 64 * u;
            if (i < kApr * n4) {
                const int r = (i >= n4) + (i >= 2 * n4) + (i >= 3 * n4), k = i - r * n4;
                *(uint2 *)(strip + r * spitch + 4 * k) = vs[u];
            }
        }
    }
}
// rows [r_lo, r_hi) x chunks [k_lo, k_hi) of the CTU's own samples back to the plane
template <int BD, int LK>
__device__ void recon_tile_store(const uint16_t *tile, int pitch, uint8_t *plane, int stride, int ox, int oy, int r_lo, int r_hi, int k_lo, int k_hi, int lane)
{
    using px_t = typename Px<BD>::type;
    constexpr int RPS = 64 >> LK;
    const int k = k_lo + (lane & ((1 << LK) - 1));
    if (k >= k_hi)
        return;
    for (int r = r_lo + (lane >> LK); r < r_hi; r += RPS) {
        const uint2 v = *(const uint2 *)(tile + r * pitch + kApr + 4 * k);
        uint8_t *p = plane + row_off(oy + r, stride) + (ox + 4 * k) * (int)sizeof(px_t);
        if (BD > 8)
            gst<uint2>(p, v);
        else
            gst<uint32_t>(p, __builtin_amdgcn_perm(v.y, v.x, 0x06040200u));
    }
}

// A LIGHT CTU (vvc355_recon_ctu.flags): MARK and RESID commands only, no luma written — the chroma residuals of inter coding units that
// chroma residual scaling keeps in the walk.  Nothing is staged: groups of sixteen lanes take the residual blocks and add them straight on the
// planes; a block's scale comes from the luma plane (its own CTU's luma is final; a neighbour's once that neighbour's luma flag is up).
// Inside the CTU every neighbour of a 64x64 unit belongs to an earlier unit of the same CTU (the MARKs of a conforming list say so), at
// the CTU's edge the neighbour CTU decides (ctb_left / ctb_up), which is what ff_vvc_get_left / top_available answer for one sample.
template <int BD>
__device__ __forceinline__ void recon_light_ctu(const vvc355_recon_frame &f, ReconLds &L, const vvc355_recon_ctu &ctu, const ReconCtx &cx, VVC355_GLOBAL int *state,
                                                const int rs, const int rx, const int ry, const int role, const int tid)
{
    using px_t = typename Px<BD>::type;
    const int ncx = f.ctb_width, n_ctus = ncx * f.ctb_height, ctb = 1 << f.ctb_log2, size_y = min(ctb, 64);
    if (role == 0) {
        RTRACE(rs, 0, wall_clock64());
        const int dep[2] = { ((ctu.flags & VVC355_RECON_CTU_LUMA_LEFT) && rx > 0) ? rs - 1 : -1, ((ctu.flags & VVC355_RECON_CTU_LUMA_UP) && ry > 0) ? rs - ncx : -1 };
#pragma unroll
        for (int d = 0; d < 2; d++) {
            if (dep[d] < 0 || __builtin_amdgcn_readfirstlane(gld<uint32_t>(&((const vvc355_recon_ctu *)f.ctus)[dep[d]].n_cmd)) == 0)
                continue;                         // (a neighbour without commands never raises a flag: its luma was final before the pass)
            while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&state[kReconFlags + n_ctus + dep[d]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0)
                __builtin_amdgcn_s_sleep(4);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        RTRACE(rs, 1, wall_clock64()); RTRACE(rs, 3, wall_clock64());
    }
    __syncthreads();
    const vvc355_recon_cmd *cmds = (const vvc355_recon_cmd *)f.cmds + ctu.first_cmd;
    const uint8_t *luma = (const uint8_t *)f.plane[0];
    const int ls = f.stride[0] / (int)sizeof(px_t);
    // The scales of the CTU's 64x64 units first (at most four; the luma they average is final now), two per wave, into LDS.  Inside the CTU
    // every neighbour of a unit belongs to an earlier unit of the same CTU, at the CTU's edge the neighbour CTU decides (ctb_left / ctb_up).
    const int upr = ctb / size_y;                      // units per CTU row: 1 or 2
    if (f.lmcs_model) {
        for (int u = role; u < upr * upr; u += 2) {
            const int xv = cx.ox + (u % upr) * size_y, yv = cx.oy + (u / upr) * size_y;
            if (xv >= f.width || yv >= f.height)
                continue;
            const bool avail_l = (xv & (ctb - 1)) ? true : cx.ctb_left != 0;
            const bool avail_t = (yv & (ctb - 1)) ? true : cx.ctb_up != 0;
            const int sc = lmcs_scale_from_plane<BD>((const vvc355_lmcs_model *)f.lmcs_model, luma, ls, xv, yv, size_y, avail_l, avail_t, f.width, f.height, tid);
            L.prm[0][u] = sc;
        }
    }
    __syncthreads();
    // Then the residual blocks: they do not depend on each other, so sixteen lanes take a block and the workgroup's eight groups run eight
    // blocks at a time.  Windows of 64 commands: lane i looks at the kind of command win + i (dword 6: mode, kind, c_idx, ref_idx), the
    // residual blocks of the window are ranked through LDS, group g takes the blocks ranked g, g + 8, ... and fetches only those commands.
    VVC355_LDS uint32_t *list = (VVC355_LDS uint32_t *)L.cmdbuf[role];
    const int grp = role * 4 + (tid >> 4), l16 = tid & 15;
    for (uint32_t win = 0; win < ctu.n_cmd; win += 64) {
        const uint32_t kind = win + tid < ctu.n_cmd ? (gld<uint32_t>((const uint32_t *)(cmds + win + tid) + 6) >> 8) & 0xff : (uint32_t)VVC355_RECON_MARK;
        if (__builtin_amdgcn_ballot_w64(kind != VVC355_RECON_MARK && kind != VVC355_RECON_RESID))
            __builtin_trap();                      // a LIGHT CTU holds nothing else
        const unsigned long long todo = __builtin_amdgcn_ballot_w64(kind == VVC355_RECON_RESID);
        const int n_res = __builtin_popcountll(todo);
        if (kind == VVC355_RECON_RESID)
            list[__builtin_popcountll(todo & ((1ull << tid) - 1))] = win + (uint32_t)tid;
        group_sync<64>();
        for (int r = grp; r < n_res; r += 8) {
            const uint32_t k = list[r];
            vvc355_recon_cmd c;
            {
                const uint2 *p = (const uint2 *)(cmds + k);
                uint2 q[5];
#pragma unroll
                for (int i = 0; i < 5; i++) q[i] = gld<uint2>(p + i);
                __builtin_memcpy(&c, q, sizeof(c));
            }
            const int c_idx = c.c_idx, hs = c_idx ? f.hs : 0, vs = c_idx ? f.vs : 0;
            const int unit = (((c.cu_y0 - cx.oy) & (ctb - 1)) / size_y) * upr + ((c.cu_x0 - cx.ox) & (ctb - 1)) / size_y;
            const int scale = (c.joint & 8) ? L.prm[0][unit] : 0;
            uint8_t *dst = (uint8_t *)f.plane[c_idx] + row_off(c.y0 >> vs, f.stride[c_idx]) + (c.x0 >> hs) * (int)sizeof(px_t);
            resid_block_add<BD, 16>(dst, f.stride[c_idx], (const int *)c.resid, c.w, c.h, c.joint, scale, l16);
        }
        group_sync<64>();                          // the list is rewritten by the next window
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (role == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(&state[kReconFlags + n_ctus + rs], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&state[kReconFlags + rs], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        RTRACE(rs, 2, wall_clock64()); RTRACE(rs, 4, wall_clock64()); RTRACE(rs, 5, wall_clock64());
        RTRACE(rs, 6, ctu.flags | ((unsigned long long)ctu.n_cmd << 8)); 
    }
}

// one CTU: takes the next ticket, returns false when none is left
template <int BD, bool TILE>
__device__ __forceinline__ bool recon_one_ctu(const vvc355_recon_frame &f, ReconLds &L, uint16_t *tiles, const int role, const int tid)
{
    using px_t = typename Px<BD>::type;
    using PX = typename std::conditional<TILE, LPix, GPix<BD>>::type;
    VVC355_GLOBAL int *state = (VVC355_GLOBAL int *)f.state;
    // Every branch around a workgroup barrier in this loop is wave-uniform (whole waves take it, lanes write identical values where
    // one lane would do): a lane-0-only block next to the loop's back edge lets the compiler's control-flow structurizer move that
    // lane's code across the barrier of the next iteration.
    if (role == 0) {
        // lane 0 adds 1, the other lanes 0: lane 0's return value is the ticket
        const int t = __hip_atomic_fetch_add(&state[0], tid == 0 ? 1 : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        L.bc[0] = __builtin_amdgcn_readfirstlane(t);
        recon_luma_done_set(L, 0);
    }
    __syncthreads();
    const int ticket = __builtin_amdgcn_readfirstlane(L.bc[0]);
    if (ticket >= f.n_work)
        return false;
    const int rs = __builtin_amdgcn_readfirstlane(gld<int>((const int *)f.order + ticket));
    const vvc355_recon_ctu *ctus = (const vvc355_recon_ctu *)f.ctus;
    const vvc355_recon_ctu ctu = load_uniform(ctus + rs);
    const int ncx = f.ctb_width, ry = rs / ncx, rx = rs - ry * ncx;
    const int ctb = 1 << f.ctb_log2;
    ReconCtx cx;
    {
        // ff_vvc_decode_neighbour (vvc_ctu.c:2468-2495)
        const int16_t *slice_idx = (const int16_t *)f.slice_idx, *col_bd = (const int16_t *)f.ctb_to_col_bd, *row_bd = (const int16_t *)f.ctb_to_row_bd;
        const bool left_tile = rx > 0 && gld<int16_t>(col_bd + rx) != gld<int16_t>(col_bd + rx - 1);
        const bool upper_tile = ry > 0 && gld<int16_t>(row_bd + ry) != gld<int16_t>(row_bd + ry - 1);
        const bool upper_slice = ry > 0 && gld<int16_t>(slice_idx + rs) != gld<int16_t>(slice_idx + rs - ncx);
        cx.end_of_tiles_x = f.width;
        if (gld<int16_t>(col_bd + rx) != gld<int16_t>(col_bd + rx + 1))
            cx.end_of_tiles_x = min(rx * ctb + ctb, f.width);
        cx.ctb_left = rx > 0 && !left_tile;
        cx.ctb_up = ry > 0 && !upper_tile && !upper_slice;
        cx.ctb_up_left = cx.ctb_left && cx.ctb_up;
        cx.ox = rx * ctb; cx.oy = ry * ctb;
    }
    RTRACE(rs, 7, ticket | ((unsigned long long)blockIdx.x << 32));
    if (ctu.flags & VVC355_RECON_CTU_LIGHT) {          // workgroup-uniform
        recon_light_ctu<BD>(f, L, ctu, cx, state, rs, rx, ry, role, tid);
        return true;
    }
    // the three planes as the walk sees them: accessor of sample (0, 0) + stride in pixels
    PX pl0, pl1, pl2;
    int ps0, ps1;                       // strides in pixels: luma, chroma
    constexpr int kTileOff[3] = { 0, kBodyLuma, kBodyLuma + kBodyChroma };
    constexpr int kStripOff[3] = { kBodyLuma + 2 * kBodyChroma, kBodyLuma + 2 * kBodyChroma + kStripLuma, kBodyLuma + 2 * kBodyChroma + kStripLuma + kStripChroma };
    StripRef st0 = kNoStrip, st1 = kNoStrip, st2 = kNoStrip;
    const int cw0 = min(ctb, f.width - rx * ctb), ch0 = min(ctb, f.height - ry * ctb);
    // (TILE) this CTU's own samples -> LDS: they do not depend on the neighbours, so the loads overlap the wait
    if constexpr (TILE) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int sh = c ? 1 : 0, pitch = c ? kBodyChromaP : kBodyLumaP, spitch = c ? kStripChromaP : kStripLumaP;
            const int ox = (rx * ctb) >> sh, oy = (ry * ctb) >> sh;
            if ((c == 0) == (role == 0)) {     // the luma wave brings in the luma tile, the chroma wave both chroma tiles
                if (c == 0)
                    recon_body_load<BD, 5>(tiles + kTileOff[c], pitch, (const uint8_t *)f.plane[c], f.stride[c], ox, oy, cw0, ch0, f.width, f.height, tid);
                else
                    recon_body_load<BD, 4>(tiles + kTileOff[c], pitch, (const uint8_t *)f.plane[c], f.stride[c], ox, oy, cw0 >> 1, ch0 >> 1, f.width >> 1, f.height >> 1, tid);
            }
            const LPix px{ tiles, kTileOff[c] - oy * pitch + (kApr - ox) };
            const StripRef sx{ tiles + kStripOff[c], spitch, ox - kApr, false };
            if (c == 0) { pl0 = px; st0 = sx; } else if (c == 1) { pl1 = px; st1 = sx; } else { pl2 = px; st2 = sx; }
        }
        ps0 = kBodyLumaP; ps1 = kBodyChromaP;
    }
    const vvc355_recon_cmd *cmds = (const vvc355_recon_cmd *)f.cmds + ctu.first_cmd;
    const int ctb_mask = ctb - 1;
    constexpr int CMD_DW = (int)sizeof(vvc355_recon_cmd) / 4;
    // Each wave walks only its own commands, and no MARKs (see the availability pre-pass below).  The list is taken in windows of 64 commands: the whole window (64 x 10 dwords, contiguous) goes
    // from HBM into the wave's LDS buffer in one round trip, a ballot over the kinds (dword 6: mode, kind, c_idx, ref_idx) gives the wave's
    // commands in the window, and every command is then an LDS read away — the global round trip is paid once per window, not per command
    // (a register pipeline of per-command global loads makes every rotation wait for the load issued last).
    static_assert(CMD_DW == 10, "cmdbuf holds windows of 10-dword commands");
    const uint32_t n_cmd = ctu.n_cmd;
    uint16_t (*arr)[kEdgeLen] = L.arr[role];
    const LTabs tabs{ &L.tabs };
    VVC355_LDS uint32_t *cbuf = (VVC355_LDS uint32_t *)L.cmdbuf[role];
    auto load_window = [&](uint32_t base, uint32_t *all_kinds) -> unsigned long long {
        const uint32_t n_dw = min(64u, n_cmd - base) * CMD_DW;
        const uint32_t *g = (const uint32_t *)(cmds + base);
        uint32_t v[CMD_DW];
#pragma unroll
        for (int u = 0; u < CMD_DW; u++)
            v[u] = (uint32_t)tid + 64u * u < n_dw ? gld<uint32_t>(g + tid + 64 * u) : 0u;
#pragma unroll
        for (int u = 0; u < CMD_DW; u++)
            cbuf[tid + 64 * u] = v[u];
        group_sync<64>();
        const uint32_t kinds = cbuf[tid * CMD_DW + 6];
        *all_kinds = kinds;
        return __ballot(base + (uint32_t)tid < n_cmd && ((((kinds >> 16) & 0xff) > 0) == (role == 1)) && ((kinds >> 8) & 0xff) != VVC355_RECON_MARK);
    };
    // Availability pre-pass.  What ff_vvc_get_top_available / _left_available return for a block depends on the command list alone (the areas
    // the MARKs before it record), not on any sample — so it does not belong on the chain of dependent blocks.  Before waiting for the
    // neighbours each wave runs through the whole list once: MARKs update its bitmaps of reconstructed 4x4 units (the chroma wave keeps the
    // luma bitmap as well: CCLM and the chroma residual scale ask about luma neighbours), and for each of its own PRED / CCLM / scaled RESID
    // commands the answers go into the command's pad bytes in HBM (dword 9 = left | top << 16 in the component's samples; pad_[0] bits
    // 0 / 1 = the 1-sample luma questions).  The walk below reads them back with the command and never sees a MARK.  A workgroup that
    // took its ticket ahead of its neighbours — the usual case on a dependency chain — does this while it would be waiting anyway.
    {
        ReconMaps &rmap = L.rmap[role];
        ((uint32_t *)rmap)[tid] = 0;
        ((uint32_t *)rmap)[tid + 64] = 0;
        uint32_t *cmd_dw = (uint32_t *)cmds;
        for (uint32_t base = 0; base < n_cmd; base += 64) {
            uint32_t kinds;
            (void)load_window(base, &kinds);
            const uint32_t kind_l = (kinds >> 8) & 0xff, type_l = ((kinds >> 16) & 0xff) > 0;
            const bool in_l = base + (uint32_t)tid < n_cmd;
            const bool mine = type_l == (uint32_t)role;
            // MARKs of the types this wave tracks; its own PREDs and CCLMs; its own RESIDs with chroma residual scaling (joint bit 3: dword 8, byte 1)
            const bool scaled_l = in_l && mine && kind_l == VVC355_RECON_RESID && ((cbuf[tid * CMD_DW + 8] >> 8) & 8);
            unsigned long long todo = __ballot(in_l && ((kind_l == VVC355_RECON_MARK && (role == 1 || type_l == 0)) ||
                                                        (mine && (kind_l == VVC355_RECON_PRED || kind_l == VVC355_RECON_CCLM)) || scaled_l));
            while (todo) {
                const int k = __builtin_ctzll(todo);
                todo &= todo - 1;
                const uint32_t q2 = __builtin_amdgcn_readfirstlane(cbuf[k * CMD_DW + 2]), q3 = __builtin_amdgcn_readfirstlane(cbuf[k * CMD_DW + 3]);
                const uint32_t q4 = __builtin_amdgcn_readfirstlane(cbuf[k * CMD_DW + 4]), q6 = __builtin_amdgcn_readfirstlane(cbuf[k * CMD_DW + 6]);
                const int x0 = (int16_t)(q2 & 0xffff), y0 = (int16_t)(q2 >> 16), cw_ = (int16_t)(q3 & 0xffff), ch_ = (int16_t)(q3 >> 16);
                const int cu_x0 = (int16_t)(q4 & 0xffff), cu_y0 = (int16_t)(q4 >> 16);
                const int kind = (q6 >> 8) & 0xff, c_idx = (q6 >> 16) & 0xff;
                if (kind == VVC355_RECON_MARK) {
                    // add_reconstructed_area (vvc_intra.c:188-206): lanes 0-31 set the unit rows, lanes 32-63 the unit columns
                    const int ux = (x0 - cx.ox) >> 2, uy = (y0 - cx.oy) >> 2, uw = max(1, cw_ >> 2), uh = max(1, ch_ >> 2);
                    const int i = tid & 31, t = tid >> 5;
                    const int span = t ? uw : uh, first = t ? ux : uy, len = t ? uh : uw, at = t ? uy : ux;
                    if (i < span)
                        rmap[c_idx > 0][t][first + i] |= (len >= 32 ? ~0u : ((1u << len) - 1)) << at;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // LDS accesses of a wave stay in order; this only pins the compiler
                    continue;
                }
                uint32_t dw9 = 0, bits = 0;
                if (kind == VVC355_RECON_PRED) {
                    const int hs = c_idx ? f.hs : 0, vs = c_idx ? f.vs : 0;
                    dw9 = (uint32_t)(uint16_t)__builtin_amdgcn_readfirstlane(recon_left_available(f, cx, rmap, cu_y0, x0 >> hs, y0 >> vs, 16384, c_idx)) |
                          (uint32_t)(uint16_t)__builtin_amdgcn_readfirstlane(recon_top_available(f, cx, rmap, cu_x0, x0 >> hs, y0 >> vs, 16384, c_idx)) << 16;
                } else if (kind == VVC355_RECON_CCLM) {
                    dw9 = (uint32_t)(uint16_t)__builtin_amdgcn_readfirstlane(recon_left_available(f, cx, rmap, cu_y0, x0 >> f.hs, y0 >> f.vs, 16384, 1)) |
                          (uint32_t)(uint16_t)__builtin_amdgcn_readfirstlane(recon_top_available(f, cx, rmap, cu_x0, x0 >> f.hs, y0 >> f.vs, 16384, 1)) << 16;
                    bits = (__builtin_amdgcn_readfirstlane(recon_left_available(f, cx, rmap, cu_y0, x0, y0, 1, 0)) != 0 ? 1u : 0u) |
                           (__builtin_amdgcn_readfirstlane(recon_top_available(f, cx, rmap, cu_x0, x0, y0, 1, 0)) != 0 ? 2u : 0u);
                } else {
                    // the 64x64 unit of the coding unit (lmcs_derive_chroma_scale, vvc_intra_template.c:399-404)
                    const int size_y = min(ctb, 64), xv = cu_x0 & ~(size_y - 1), yv = cu_y0 & ~(size_y - 1);
                    bits = (__builtin_amdgcn_readfirstlane(recon_left_available(f, cx, rmap, cu_y0, xv, yv, 1, 0)) != 0 ? 1u : 0u) |
                           (__builtin_amdgcn_readfirstlane(recon_top_available(f, cx, rmap, cu_x0, xv, yv, 1, 0)) != 0 ? 2u : 0u);
                }
                if (tid == 0) {
                    uint32_t *g = cmd_dw + (size_t)(base + k) * CMD_DW;
                    gst<uint32_t>(g + 9, dw9);
                    gst<uint8_t>((uint8_t *)(g + 8) + 2, (uint8_t)bits);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the answers are in L2 before this wave fetches its windows again (after the acquire below)
    }
    // Wait for the neighbours this wave reads — left, upper-left, upper, upper-right, those that have commands.  The luma wave needs their
    // LUMA only (flag raised as soon as a neighbour's luma commands are done; a LIGHT neighbour's luma was final before the pass), the
    // chroma wave the whole neighbour.  From here to the join at the end the two waves run on their own: each waits, loads its own edges
    // and walks its commands; where the chroma wave reads luma (CCLM, the chroma residual scale) it synchronises on luma_done.
    const unsigned long long t_wait = RPROF_NOW();
    if (role == 0) { RTRACE(rs, 0, wall_clock64()); RTRACE(rs, 6, ctu.flags | ((unsigned long long)ctu.n_cmd << 8)); }
    {
        const int n_ctus = ncx * f.ctb_height;
        const int dep[4] = { rx > 0 ? rs - 1 : -1, (rx > 0 && ry > 0) ? rs - ncx - 1 : -1, ry > 0 ? rs - ncx : -1, (ry > 0 && rx + 1 < ncx) ? rs - ncx + 1 : -1 };
#pragma unroll
        for (int d = 0; d < 4; d++) {
            if (dep[d] < 0 || __builtin_amdgcn_readfirstlane(gld<uint32_t>(&ctus[dep[d]].n_cmd)) == 0)
                continue;
            if (role == 0 && (__builtin_amdgcn_readfirstlane(gld<uint32_t>(&ctus[dep[d]].flags)) & VVC355_RECON_CTU_LIGHT))
                continue;
            VVC355_GLOBAL int *flag = &state[kReconFlags + (role == 0 ? n_ctus : 0) + dep[d]];
            while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0)
                __builtin_amdgcn_s_sleep(4);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    RPROF_ADD(1 + 32 * role, t_wait);
    RPROF_INC(0 + 32 * role);
    RTRACE(rs, role ? 3 : 1, wall_clock64());
    const unsigned long long t_load = RPROF_NOW();
    if constexpr (TILE) {
        // what the neighbours wrote: left apron and the rows above
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int sh = c ? 1 : 0;
            if ((c == 0) == (role == 0))
                recon_edges_load<BD>(tiles + kTileOff[c], c ? kBodyChromaP : kBodyLumaP, tiles + kStripOff[c], c ? kStripChromaP : kStripLumaP, (const uint8_t *)f.plane[c], f.stride[c],
                                     (rx * ctb) >> sh, (ry * ctb) >> sh, ctb >> sh, ch0 >> sh, f.width >> sh, f.height >> sh, tid);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        group_sync<64>();
    } else {
        pl0 = GPix<BD>{ (uint8_t *)f.plane[0] }; pl1 = GPix<BD>{ (uint8_t *)f.plane[1] }; pl2 = GPix<BD>{ (uint8_t *)f.plane[2] };
        ps0 = f.stride[0] / (int)sizeof(px_t); ps1 = f.stride[1] / (int)sizeof(px_t);
    }

    RPROF_ADD(2 + 32 * role, t_load);
    const unsigned long long t_loop = RPROF_NOW();
#ifdef VVC355_RECON_PROF
    const long long c_loop = clock64();
#endif
    uint32_t win = 0, kinds_unused;
    unsigned long long mask = load_window(0, &kinds_unused);
    auto advance = [&]() -> int {         // index of the wave's next command, -1 when its list is exhausted
        while (!mask) {
            if (win + 64 >= n_cmd)
                return -1;
            win += 64;
            mask = load_window(win, &kinds_unused);
        }
        const int bit = __builtin_ctzll(mask);
        mask &= mask - 1;
        return (int)win + bit;
    };
    auto load_cmd = [&](int idx) -> uint32_t { return cbuf[(idx - (int)win) * CMD_DW + min(tid, CMD_DW - 1)]; };      // lane i holds dword i
    int i0 = advance();
    // luma_done = every luma command below this index is finished (the chroma wave waits on it before CCLM)
    if (role == 0)
        recon_luma_done_set(L, i0 >= 0 ? i0 : (int)n_cmd);
    auto shift = [&]() {
        i0 = advance();
        if (role == 0)
            recon_luma_done_set(L, i0 >= 0 ? i0 : (int)n_cmd);          // issued after the finished command's stores (LDS: in order; planes: after s_waitcnt vmcnt(0))
    };
    int4 pre[4] = {};                    // residuals fetched one command ahead (the first 1024 of the block)
    int pre_k = -1;
    int lmcs_xv = -1, lmcs_yv = -1, lmcs_scale = 0;      // lc->lmcs: the 64x64 unit whose chroma residual scale is known (reset per CTU, vvc_intra.c:509-510)
    while (i0 >= 0) {
        const uint32_t k = (uint32_t)i0;
        const int i1 = mask ? (int)win + __builtin_ctzll(mask) : -1;        // the wave's next command, when this window holds it
        const uint32_t d0 = load_cmd(i0), d1 = i1 >= 0 ? load_cmd(i1) : 0u;
        vvc355_recon_cmd c;
        bool pair_next = false;
        uint32_t avail_dw;                   // dword 9: the pre-pass's answers (left | top << 16)
        {
            uint32_t w[CMD_DW];
#pragma unroll
            for (int i = 0; i < CMD_DW; i++) w[i] = (uint32_t)__builtin_amdgcn_readlane((int)d0, i);
            __builtin_memcpy(&c, w, sizeof(c));
            avail_dw = w[9];
            if (i1 >= 0) {
                const uint32_t n6 = (uint32_t)__builtin_amdgcn_readlane((int)d1, 6);
                // the wave's next command is a residual block: start its first loads now, so that they travel while this command
                // (usually that block's prediction) runs — unless this command is about to use the registers they go to
                if (((n6 >> 8) & 0xff) == VVC355_RECON_RESID && pre_k != i0) {
                    const uint64_t ptr = (uint32_t)__builtin_amdgcn_readlane((int)d1, 0) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)d1, 1) << 32);
                    const uint32_t wh = (uint32_t)__builtin_amdgcn_readlane((int)d1, 3);
                    const int nn = (int)(wh & 0xffff) * (int)(wh >> 16);
#pragma unroll
                    for (int u = 0; u < 4; u++)
                        if (tid * 4 + 256 * u < nn)
                            pre[u] = gld<int4>((const int *)ptr + tid * 4 + 256 * u);
                    pre_k = i1;
                }
                // Cb and Cr of a coding unit are predicted by two consecutive commands that differ in c_idx only (predict_intra,
                // vvc_intra.c:263-264): when the next command is that twin, both go through one pass, half a wave each
                if (TILE && c.kind == VVC355_RECON_PRED && c.c_idx == 1 && i1 == i0 + 1) {
                    bool same = n6 == ((w[6] & ~0xff0000u) | 0x020000u);
#pragma unroll
                    for (int i = 2; i < 9; i++)
                        if (i != 6)
                            same = same && (uint32_t)__builtin_amdgcn_readlane((int)d1, i) == w[i];
                    pair_next = same;
                }
            }
        }
        const unsigned long long t_cmd = RPROF_NOW();
        if (c.kind == VVC355_RECON_PRED) {
            const int c_idx = c.c_idx, hs = c_idx ? f.hs : 0, vs = c_idx ? f.vs : 0;
            const int x = c.x0 >> hs, y = c.y0 >> vs, w = c.w >> hs, h = c.h >> vs;
            vvc355_intra_job j = {};
            j.x = (int16_t)x; j.y = (int16_t)y; j.w = (int16_t)w; j.h = (int16_t)h;
            j.mode = (int16_t)wide_angle_mode(c.isp_split, c_idx, w, h, c.cb_width, c.cb_height, c.mode);
            j.cb_width = c.cb_width; j.cb_height = c.cb_height;
            j.left_avail = (int16_t)(avail_dw & 0xffff);
            j.top_avail = (int16_t)(avail_dw >> 16);
            j.c_idx = (uint8_t)c_idx; j.ref_idx = c_idx ? 0 : c.ref_idx;
            j.is_mip = c.is_mip; j.mip_mode = c.mip_mode; j.mip_transposed = c.mip_transposed;
            j.isp_split = c.isp_split; j.bdpcm_flag = c.bdpcm_flag;
            {   // ff_vvc_set_neighbour_available (vvc_ctu.c:2497-2510), luma coordinates
                const int x0b = c.x0 & ctb_mask, y0b = c.y0 & ctb_mask;
                const bool cand_up = cx.ctb_up || y0b, cand_left = cx.ctb_left || x0b;
                j.cand_up_left = (x0b || y0b) ? (cand_left && cand_up) : cx.ctb_up_left;
            }
            const PX plane = pick3(c_idx, pl0, pl1, pl2);
            StripRef sr = pick3(c_idx, st0, st1, st2);
            sr.on = TILE && (y & ((ctb >> vs) - 1)) == 0;          // the block sits on the CTU's top edge: the rows above it are the strip
            RPROF_ADD(6 + 32 * role, t_cmd);
            if constexpr (TILE) {
                if (pair_next) {
                    // lanes 0-31: Cb, lanes 32-63: Cr — same job, each half on its own tile, strip, edge arrays and scratch
                    const int half = tid >> 5;
                    const LPix plane2{ pl1.p, half ? pl2.base : pl1.base };
                    StripRef sr2{ st1.p + (half ? (int)(st2.p - st1.p) : 0), st1.pitch, st1.x0s, sr.on };
                    intra_pred_body<BD, 32>(j, plane2, ps1, L.arr[1 + half], L.scratch[1 + half], tid & 31, sr2, tabs);
                    group_sync<64>();
                    RPROF_ADD(9 + (int)c.kind + 32 * role, t_cmd);
                    RPROF_INC(17 + (int)c.kind + 32 * role);
                    shift();             // the twin is done as well
                    shift();
                    continue;
                }
            }
            intra_pred_body<BD, 64>(j, plane, c_idx ? ps1 : ps0, arr, L.scratch[role], tid, sr, tabs);
            if (TILE) group_sync<64>(); else recon_sync_mem();
        } else if (c.kind == VVC355_RECON_CCLM) {
            // the coding unit's luma (every luma command before this one) must be reconstructed
            while (recon_luma_done_get(L) < (int)k)
                __builtin_amdgcn_s_sleep(1);
            group_sync<64>();
            vvc355_cclm_job j = {};
            j.x0 = c.x0; j.y0 = c.y0; j.width = c.w; j.height = c.h;
            j.top_avail_c = (int16_t)(avail_dw >> 16);
            j.left_avail_c = (int16_t)(avail_dw & 0xffff);
            j.mode = (uint8_t)c.mode; j.hs = f.hs; j.vs = f.vs;
            j.avail_t = (uint8_t)((c.pad_[0] >> 1) & 1);
            j.avail_l = (uint8_t)(c.pad_[0] & 1);
            j.collocated = f.collocated;
            j.ctu_boundary = (c.y0 & ctb_mask) == 0;
            StripRef s0 = st0, s1 = st1, s2 = st2;
            s0.on = s1.on = s2.on = TILE && j.ctu_boundary;
            cclm_body<BD, 64>(j, pl0, ps0, pl1, pl2, ps1, ps1, L.prm[1], tid, s0, s1, s2);
            if (TILE) group_sync<64>(); else recon_sync_mem();
        } else if (c.kind == VVC355_RECON_CIIP) {
            // inter.put_ciip (vvc_inter_template.c:60) of a combined inter / intra block (pred_regular_luma / _chroma, vvc_inter.c:570-575,
            // :632-638): the intra prediction the previous command left in the picture, weighted against the inter prediction of the
            // batched stage (c.resid: w x h pixels, packed rows); c.joint = ciip_derive_intra_weight (:530-548)
            const int c_idx = c.c_idx, hs = c_idx ? f.hs : 0, vs = c_idx ? f.vs : 0, w = c.w >> hs, n = w * (c.h >> vs);
            const int stride = c_idx ? ps1 : ps0, iw = c.joint;
            const PX dst = pick3(c_idx, pl0, pl1, pl2).at((c.y0 >> vs) * stride + (c.x0 >> hs));
            const uint8_t *inter = (const uint8_t *)c.resid;
            const int lw = ilog2i(w);
            for (int i = tid; i < n; i += 64) {
                const int o = (i >> lw) * stride + (i & (w - 1));
                dst.st(o, (dst.ld(o) * iw + ld_px<BD>(inter, i) * (4 - iw) + 2) >> 2);
            }
            if (TILE) group_sync<64>(); else recon_sync_mem();
        } else {
            // RESID: itx.add_residual / add_residual_joint (vvcdsp_template.c:32,48) of the block the transform stage left in c.resid
            const int c_idx = c.c_idx, hs = c_idx ? f.hs : 0, vs = c_idx ? f.vs : 0, w = c.w, n = w * c.h;
            const int stride = c_idx ? ps1 : ps0;
            const PX dst = pick3(c_idx, pl0, pl1, pl2).at((c.y0 >> vs) * stride + (c.x0 >> hs));
            const int *res = (const int *)c.resid;
            const int lw = ilog2i(w);
            const bool have = pre_k == (int)k;
            const bool scaled = (c.joint & 8) != 0;
            if (scaled && (lmcs_xv != (c.cu_x0 & ~(min(ctb, 64) - 1)) || lmcs_yv != (c.cu_y0 & ~(min(ctb, 64) - 1)))) {
                // lmcs_derive_chroma_scale (vvc_intra_template.c:390-429): average of the reconstructed luma left of and above the 64x64 unit of
                // the coding unit.  Every luma command before this one must be done (the unit's neighbours may be this CTU's own blocks).
                while (recon_luma_done_get(L) < (int)k)
                    __builtin_amdgcn_s_sleep(1);
                group_sync<64>();
                const int size_y = min(ctb, 64);
                lmcs_xv = c.cu_x0 & ~(size_y - 1); lmcs_yv = c.cu_y0 & ~(size_y - 1);
                const bool avail_t = (c.pad_[0] >> 1) & 1, avail_l = c.pad_[0] & 1;
                StripRef s0 = st0;
                s0.on = TILE && (lmcs_yv & ctb_mask) == 0;
                int v = 0;
                if (tid < size_y) {
                    if (avail_l)
                        v += pl0.ld((lmcs_yv + min(tid, min(f.height - lmcs_yv, size_y) - 1)) * ps0 + lmcs_xv - 1);        // lmcs_sum_samples: the last sample stands in beyond the picture
                    if (avail_t) {
                        const int xx = lmcs_xv + min(tid, min(f.width - lmcs_xv, size_y) - 1);
                        v += s0.on ? s0.at(xx, -1) : pl0.ld((lmcs_yv - 1) * ps0 + xx);
                    }
                }
#pragma unroll
                for (int sft = 32; sft; sft >>= 1)
                    v += __shfl_xor(v, sft, 64);
                const int cnt = (avail_l ? size_y : 0) + (avail_t ? size_y : 0);
                const int avg = cnt ? (v + (cnt >> 1)) >> ilog2i(cnt) : 1 << (BD - 1);
                lmcs_scale = lmcs_scale_of_avg((const vvc355_lmcs_model *)f.lmcs_model, avg, tid);
            }
            auto resid_of = [&](int r) {                        // the joint sign / shift (pred_residual_joint), then lmcs_scale_chroma
                if (c.joint & 1)
                    r = (r * ((c.joint & 2) ? -1 : 1)) >> ((c.joint >> 2) & 1);
                if (scaled) {
                    const int v = clip_intp2(r, BD);
                    r = v > 0 ? (v * lmcs_scale + (1 << 10)) >> 11 : -((-v * lmcs_scale + (1 << 10)) >> 11);
                }
                return r;
            };
            auto add4 = [&](int i, const int4 r4) {             // w >= 4: four samples of one row per lane and step
                const int r[4] = { r4.x, r4.y, r4.z, r4.w };
                const int o = (i >> lw) * stride + (i & (w - 1));
                int d[4];
                dst.ld4(o, d);
#pragma unroll
                for (int q = 0; q < 4; q++)
                    d[q] = clip_px<BD>(d[q] + resid_of(r[q]));
                dst.st4(o, d[0], d[1], d[2], d[3]);
            };
            if (w < 4) {
                // the 1- and 2-sample-wide transform blocks of a vertically split ISP coding unit (get_luma_predict_unit, vvc_intra.c:216-226:
                // predicted 4 wide, residuals added per sub-partition): one sample per lane and step, any column parity
                for (int i = tid; i < n; i += 64) {
                    const int o = (i >> lw) * stride + (i & (w - 1));
                    dst.st(o, clip_px<BD>(dst.ld(o) + resid_of(gld<int>(res + i))));
                }
            } else {
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int i = tid * 4 + 256 * u;
                    if (i < n)
                        add4(i, have ? pre[u] : gld<int4>(res + i));
                }
                for (int i = tid * 4 + 1024; i < n; i += 256)
                    add4(i, gld<int4>(res + i));
            }
            if (TILE) group_sync<64>(); else recon_sync_mem();
        }
        RPROF_ADD(9 + (int)c.kind + 32 * role, t_cmd);
        RPROF_INC(17 + (int)c.kind + 32 * role);
        shift();
    }
    RPROF_ADD(3 + 32 * role, t_loop);
#ifdef VVC355_RECON_PROF
    { unsigned long long *p_ = rprof_lds(); p_[7 + 32 * role] = p_[7 + 32 * role] + (unsigned long long)(clock64() - c_loop); }
#endif
    if (role == 0) {
        // the luma wave is through: what other CTUs read of this CTU's luma (its last kApr rows and columns) goes out now and the luma flag
        // goes up, ahead of the chroma wave — LIGHT CTUs next to this one wait for nothing else
        if constexpr (TILE) {
            const int nk = cw0 >> 2, nr = ch0;
            recon_tile_store<BD, 5>(tiles + kTileOff[0], kBodyLumaP, (uint8_t *)f.plane[0], f.stride[0], rx * ctb, ry * ctb, max(nr - kApr, 0), nr, 0, nk, tid);
            recon_tile_store<BD, 0>(tiles + kTileOff[0], kBodyLumaP, (uint8_t *)f.plane[0], f.stride[0], rx * ctb, ry * ctb, 0, max(nr - kApr, 0), nk - 1, nk, tid);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(&state[kReconFlags + ncx * f.ctb_height + rs], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        RTRACE(rs, 2, wall_clock64());
    } else {
        RTRACE(rs, 4, wall_clock64());
    }
    const unsigned long long t_join = RPROF_NOW();
    __syncthreads();
    RPROF_ADD(4 + 32 * role, t_join);
    const unsigned long long t_store = RPROF_NOW();
    // publish: first what the neighbours read (the CTU's last kApr rows and columns), both waves' stores out of the CU, then one lane
    // releases and raises the flag; the rest of the tile goes out behind the flag (no other CTU reads it in this pass)
    if constexpr (TILE) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int sh = c ? 1 : 0, pitch = c ? kBodyChromaP : kBodyLumaP, nk = cw0 >> (2 + sh), nr = ch0 >> sh;
            if ((c == 0) != (role == 0))
                continue;
            uint8_t *plane = (uint8_t *)f.plane[c];
            if (c == 0) {
                // (the luma wave stored its rows and columns before the join)
            } else {
                recon_tile_store<BD, 4>(tiles + kTileOff[c], pitch, plane, f.stride[c], (rx * ctb) >> sh, (ry * ctb) >> sh, max(nr - kApr, 0), nr, 0, nk, tid);
                recon_tile_store<BD, 0>(tiles + kTileOff[c], pitch, plane, f.stride[c], (rx * ctb) >> sh, (ry * ctb) >> sh, 0, max(nr - kApr, 0), nk - 1, nk, tid);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (role == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(&state[kReconFlags + rs], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        RTRACE(rs, 5, wall_clock64());
    }
    if constexpr (TILE) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int sh = c ? 1 : 0, pitch = c ? kBodyChromaP : kBodyLumaP, nk = cw0 >> (2 + sh), nr = ch0 >> sh;
            if ((c == 0) != (role == 0))
                continue;
            if (c == 0)
                recon_tile_store<BD, 5>(tiles + kTileOff[c], pitch, (uint8_t *)f.plane[c], f.stride[c], (rx * ctb) >> sh, (ry * ctb) >> sh, 0, max(nr - kApr, 0), 0, nk - 1, tid);
            else
                recon_tile_store<BD, 4>(tiles + kTileOff[c], pitch, (uint8_t *)f.plane[c], f.stride[c], (rx * ctb) >> sh, (ry * ctb) >> sh, 0, max(nr - kApr, 0), 0, nk - 1, tid);
        }
    }
    RPROF_ADD(5 + 32 * role, t_store);
    return true;
}

// A persistent grid: each workgroup (two waves) takes CTU tickets until none is left, so only as many CTUs as can run at once hold
// LDS and wave slots — the rest of the device stays free for the other frames' kernels.  Any grid size is deadlock-free (tickets
// follow the order in which CTUs become ready: every CTU a workgroup waits for has a smaller ticket, held by a running workgroup).
template <int BD, bool TILE>
__global__ __launch_bounds__(128) void recon_wavefront_kernel(const vvc355_recon_frame *__restrict__ fp)
{
    __shared__ ReconLds L;
    __shared__ __attribute__((aligned(16))) uint16_t tiles[TILE ? kTileSamples : 8];
    const vvc355_recon_frame f = load_uniform(fp);
    // two waves per CTU: wave 0 walks the luma commands, wave 1 the chroma commands.  The two chains only meet at CCLM (chroma
    // predicted from the coding unit's reconstructed luma): the chroma wave waits there until the luma wave has passed that command.
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), tid = threadIdx.x & 63;
    // the predictors' tables -> LDS (visible after the first CTU's barriers)
    {
        uint32_t *t32 = (uint32_t *)&L.tabs.luma_filter[0];
        constexpr int N0 = 64, N1 = N0 + 256, N2 = N1 + 256, N3 = N2 + 672;
        static_assert(sizeof(IntraTabsLds) == (32 + N3) * 4, "table copy covers the whole struct");
        for (int i = threadIdx.x; i < N3; i += 128) {
            const uint8_t *g = i < N0 ? (const uint8_t *)i_tab_intra_luma_filter + 4 * i
                             : i < N1 ? i_tab_mip_matrix_4x4 + 4 * (i - N0)
                             : i < N2 ? i_tab_mip_matrix_8x8 + 4 * (i - N1) : i_tab_mip_matrix_16x16 + 4 * (i - N2);
            t32[i] = gld<uint32_t>(g);
        }
        if (threadIdx.x < 32)
            L.tabs.angle_inv[threadIdx.x] = intra_angle_entry(threadIdx.x);
    }
#ifdef VVC355_RECON_PROF
    rprof_lds()[threadIdx.x & 63] = 0;
    __syncthreads();
#endif
    while (recon_one_ctu<BD, TILE>(f, L, tiles, role, tid)) {
    }
#ifdef VVC355_RECON_PROF
    __syncthreads();
    if (role == 0)
        atomicAdd(&vvc355_recon_prof[tid], rprof_lds()[tid]);
#endif
}

} // namespace vvc355

extern "C" {

#ifdef VVC355_RECON_PROF
void vvc355_recon_prof_read(unsigned long long *out, int reset)
{
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(vvc355::vvc355_recon_prof), sizeof(unsigned long long) * 64));
    int dev = 0, khz = 0, ckhz = 0;
    HIP_CHECK(hipGetDevice(&dev));
    HIP_CHECK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev));
    HIP_CHECK(hipDeviceGetAttribute(&ckhz, hipDeviceAttributeClockRate, dev));
    out[63] = (unsigned long long)khz;
    out[62] = (unsigned long long)ckhz;
    if (reset == 2) {       // out: 4096 * 8 stamps
        HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(vvc355::vvc355_recon_trace), sizeof(unsigned long long) * 4096 * 8));
        return;
    }
    if (reset) {
        unsigned long long z[64] = {};
        HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(vvc355::vvc355_recon_prof), z, sizeof(z)));
    }
}
#endif

size_t vvc355_recon_state_bytes(int n_ctus) { return sizeof(int) * (size_t)(vvc355::kReconFlags + 2 * (n_ctus > 0 ? n_ctus : 0)); }

void vvc355_recon_frame_pass(void *stream, int bd, const vvc355_recon_frame *frame_dev, const vvc355_recon_frame *frame_host)
{
    using namespace vvc355;
    if (frame_host->n_work <= 0) return;
    HIP_CHECK(hipMemsetAsync((void *)frame_host->state, 0, vvc355_recon_state_bytes(frame_host->ctb_width * frame_host->ctb_height), (hipStream_t)stream));
    // 4:2:0 with CTUs up to 128x128 walks on LDS tiles (picture widths are multiples of 8: whole 4-sample chunks); other formats on the planes
    const bool tile = frame_host->hs == 1 && frame_host->vs == 1 && frame_host->ctb_log2 <= 7 && frame_host->width % 8 == 0 && frame_host->height % 2 == 0;
    // enough workgroups for the CTUs that can run at once (independent intra clusters of an inter picture; half a CTU row of an intra
    // picture), few enough that their LDS leaves room for the other frames' kernels.  VVC355_RECON_GRID overrides (tuning aid).
    static const int grid_env = [] { const char *e = getenv("VVC355_RECON_GRID"); return e ? atoi(e) : 0; }();
    const int grid = std::min(frame_host->n_work, grid_env > 0 ? grid_env : frame_host->workgroups ? (int)frame_host->workgroups : 192);
    if (tile)
        VVC355_BD_DISPATCH(bd, hipLaunchKernelGGL((recon_wavefront_kernel<BD, true>), dim3(grid), dim3(128), 0, (hipStream_t)stream, frame_dev));
    else
        VVC355_BD_DISPATCH(bd, hipLaunchKernelGGL((recon_wavefront_kernel<BD, false>), dim3(grid), dim3(128), 0, (hipStream_t)stream, frame_dev));
    HIP_CHECK(hipGetLastError());
}

} // extern "C"
