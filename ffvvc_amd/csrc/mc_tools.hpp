// Regular bi-prediction with the decoder-side tools (DMVR and / or BDOF) for gfx950 — the shapes those tools run on:
// sub-blocks of 8 or 16 samples on a side (vvc_inter.c:772-822, pred_regular_blk: derive_sb_mv -> dmvr_mv_refine :685-748,
// luma_mc_bi :253-296, apply_bdof vvc_inter_template.c:288).  Included by mc_fused.hip (inside namespace vvc355).
//
// One wave per sub-block, W and H compile-time, every loop unrolled.  What bounds this stage is VALU issue, not HBM, so the
// design goal is instruction count:
//
//  * ONE fetch per reference.  Everything the block reads lies in the (W + 7) x (H + 7) window of the UNREFINED motion:
//    the DMVR bilinear planes ((W + 5) x (H + 5) inside it), the 8-tap windows at the refined motion (emulated_edge_dmvr,
//    vvc_inter.c:61-88, clamps them to exactly this window) and the BDOF ring (bdof_fetch_samples reads inside the 8-tap
//    window).  The window goes to LDS once with 8-byte vector loads; a refined read at window index u + d is LDS index
//    clamp(u + d, 0, W + 6) — columns through three replicated pad columns, rows through a clamped row index.
//  * DMVR: packed 16-bit bilinear stage, one natural copy of each plane; the 25 costs come from (dy, row) units that hold
//    both plane rows in registers and form all five dx with v_alignbit + v_sad_u16, reduced inside 8-lane groups by DPP.
//  * 8-tap interpolation: the zero-fraction filter {0,0,0,64,0,0,0,0} makes copy / h / v / hv one formula (exact: 64 s >>
//    (bd - 8) = s << (14 - bd), (64 t) >> 6 = t), so both references go through one unrolled h pass (each half-wave one
//    reference, two rows x two columns per item, v_dot2 on aligned sample pairs, window parity folded into the tap
//    registers) and a v pass whose lane owns column x, rows 4 by .. 4 by + 3.
//  * BDOF: row pairs packed in 16-bit halves (gradients, TH / TV / D, sign products: v_pk_*), the 6 x 6 window sums as
//    separable box sums — vertical from the lane's own four rows plus the neighbours' edge rows through LDS, horizontal by
//    DPP adds inside the 16-lane row; ring replication = doubled edge weights (columns) / own-row reads (rows);
//    the output is one v_dot2 per sample on (ghd, gvd) x (vx, vy).
#pragma once

typedef short pk_i16 __attribute__((ext_vector_type(2)));
typedef unsigned short pk_u16 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ pk_i16 as_pk(uint32_t v) { return __builtin_bit_cast(pk_i16, v); }
__device__ __forceinline__ uint32_t as_u32(pk_i16 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ pk_i16 pk_splat(int v) { return pk_i16{ (short)v, (short)v }; }
__device__ __forceinline__ pk_i16 pk_max(pk_i16 a, pk_i16 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ pk_i16 pk_min(pk_i16 a, pk_i16 b) { return __builtin_elementwise_min(a, b); }

// DPP move with zero fill: lane <- the lane CTRL selects, 0 when that lane does not exist
template <int CTRL> __device__ __forceinline__ int dpp0(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
static constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppHalfMirror = 0x141;          // quad_perm [1,0,3,2], [2,3,0,1]
static constexpr int kDppShr1 = 0x111, kDppShr2 = 0x112, kDppShl1 = 0x101, kDppQuad2 = 0xAA;   // quad_perm [2,2,2,2]

static constexpr int kTwP = 40;         // window pitch in samples: 20 dwords, so 4 consecutive row pairs x 8 dwords tile the 32 banks
static constexpr int kTwC0 = 5;         // LDS column of window column 0 (3 pad columns + alignment: window column 1 sits on an even column)
static constexpr int kTtP = 28;         // transposed intermediate: [column][row], 24 rows used, 14 dwords (2 * 14 = -4 mod 32)
static constexpr int kSmP = 20;         // BDOF sample planes: [column 0..17][element = row + 2]

template <int W, int H> struct ToolsLds {
    static constexpr int NR = H + 7, NRA = (NR + 7) / 8 * 8;
    uint16_t win[2][NRA * kTwP];                        // pre-clamped windows of the unrefined motion
    union {
        int16_t bil[2][(H + 4) * (W + 4)];              // DMVR bilinear planes
        int16_t tmpT[2][16 * kTtP];                     // h-pass output, transposed
        int16_t smp[2][18 * kSmP];                      // BDOF: the two predictions + ring, column-major
    };
    int sad[32];
};
// the vertical exchange of the BDOF box sums overlays the (by then dead) windows: [quantity][column][by][pair]
static constexpr int kXchBytes = 5 * 16 * 4 * 2 * 4;

struct Taps8 { uint32_t e[4], o[5]; };
__device__ __forceinline__ Taps8 make_taps8(uint32_t lo, uint32_t hi)
{
    int f[8];
#pragma unroll
    for (int k = 0; k < 8; k++) f[k] = tap_of(lo, hi, k);
    Taps8 t;
#pragma unroll
    for (int m = 0; m < 4; m++) t.e[m] = pack16(f[2 * m], f[2 * m + 1]);
    t.o[0] = pack16(0, f[0]);
#pragma unroll
    for (int m = 1; m < 4; m++) t.o[m] = pack16(f[2 * m - 1], f[2 * m]);
    t.o[4] = pack16(f[7], 0);
    return t;
}

template <int BD, int W, int H>
__device__ __forceinline__ void bipred_tools(const vvc355_bipred_job *job, ToolsLds<W, H> &L, int lane)
{
    using px_t = typename Px<BD>::type;
    constexpr int NR = H + 7, NC = W + 7, ISZ = (int)sizeof(px_t);
    constexpr int NV4 = (W + 8) / 4;                    // 8-byte vectors per window row: columns -1 .. W + 6
    const int dmvr = job->dmvr, pic_w = job->pic_w, pic_h = job->pic_h;
    int mv[4] = { job->mv[0], job->mv[1], job->mv[2], job->mv[3] };
    int bdof = job->bdof;
    const uint8_t *const ref[2] = { (const uint8_t *)job->ref0, (const uint8_t *)job->ref1 };
    const int rs[2] = { job->ref0_stride, job->ref1_stride };
    int xs[2], ys[2];                                   // plane coordinates of window sample (0, 0)
#pragma unroll
    for (int i = 0; i < 2; i++) {
        xs[i] = job->x + (mv[2 * i] >> 4) - 3;
        ys[i] = job->y + (mv[2 * i + 1] >> 4) - 3;
    }

    // ------------------------------------------------------------------ F: both windows -> LDS
    bool inside = true;
#pragma unroll
    for (int i = 0; i < 2; i++)
        inside = inside && xs[i] >= 1 && xs[i] - 1 + 4 * NV4 <= pic_w && ys[i] >= 0 && ys[i] + NR <= pic_h;
    if (inside) {
        constexpr int NIT = (NR + 7) / 8;
        const int k = lane & 7, rsub = lane >> 3;
        if (k < NV4) {
            typename std::conditional<(BD > 8), uint2, uint32_t>::type q[2][NIT];
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const uint8_t *org = ref[i] + row_off(ys[i], rs[i]) + (xs[i] - 1) * ISZ;        // wave-uniform
#pragma unroll
                for (int it = 0; it < NIT; it++) {
                    const int r = 8 * it + 8 <= NR ? rsub + 8 * it : min(rsub + 8 * it, NR - 1);
                    const uint32_t off = (uint32_t)(__mul24(r, rs[i]) + k * 4 * ISZ);
                    if constexpr (BD > 8) q[i][it] = gld_at<uint2>(org, off);
                    else                  q[i][it] = gld_at<uint32_t>(org, off);
                }
            }
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int it = 0; it < NIT; it++) {
                    uint2 v;
                    if constexpr (BD > 8) v = q[i][it];
                    else v = make_uint2(__builtin_amdgcn_perm(0, q[i][it], 0x0c010c00u), __builtin_amdgcn_perm(0, q[i][it], 0x0c030c02u));
                    *(uint2 *)&L.win[i][(rsub + 8 * it) * kTwP + 4 + 4 * k] = v;             // column -1 lands on LDS column 4
                }
        }
        wave_sync();
        // pad columns: three replicas of the first and of the last window column
        {
            const int i = lane >> 5, r = lane & 31;
            if (r < NR) {
                uint16_t *row = &L.win[i][r * kTwP];
                const uint32_t a = row[kTwC0], b = row[kTwC0 + NC - 1];
                *(uint32_t *)&row[2] = a * 0x10001u;
                row[4] = (uint16_t)a;
                *(uint32_t *)&row[kTwC0 + NC] = b * 0x10001u;
            }
        }
    } else {
        // windows that leave the picture: one clamped sample per lane and step (edge emulation, vvc_inter.c:33-110), pad
        // columns included.  With DMVR the readable rectangle is window-of-the-unrefined-motion x picture: the same clamp.
        const int c = 2 + (lane & 31), j = c - kTwC0;
        if (c <= kTwC0 + NC + 1) {
#pragma unroll 1
            for (int it = 0; it < (NR + 1) / 2; it++) {
                const int r = (lane >> 5) + 2 * it;
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    // pad columns replicate the window's own edge column (not the plane beyond it)
                    const int xa = clip3(xs[i] + clip3(j, 0, NC - 1), 0, pic_w - 1), ya = clip3(ys[i] + r, 0, pic_h - 1);
                    const uint16_t s = (uint16_t)gld_at<px_t>(ref[i], (uint32_t)(__mul24(ya, rs[i]) + xa * ISZ));
                    if (r < NR)
                        L.win[i][r * kTwP + c] = s;
                }
            }
        }
    }
    wave_sync();

    // ------------------------------------------------------------------ D: dmvr_mv_refine (vvc_inter.c:685-748)
    int min_sad = 0, searched = 0;
    if (dmvr) {
        constexpr int PW = W + 4, PH = H + 4, NPAIR = PW / 2, NSEG = NPAIR == 10 ? 3 : 5, RPS = (PH + NSEG - 1) / NSEG;
        constexpr int PER_REF = NPAIR * NSEG;
        // inter.dmvr[!!my][!!mx] (vvc_inter_template.c:324-413): lane -> (reference, pair of columns, segment of rows); bilinear
        // sample (r, c) reads window rows r + 1, r + 2 and columns c + 1, c + 2
        {
            const int i = lane >= PER_REF, id = lane - i * PER_REF;
            int seg = 0;
#pragma unroll
            for (int s = 1; s < NSEG; s++) seg += id >= s * NPAIR;
            const int cp = id - seg * NPAIR;
            if (lane < 2 * PER_REF) {
                const int mx = (i ? mv[2] : mv[0]) & 15, my = (i ? mv[3] : mv[1]) & 15;
                const uint16_t *win = L.win[i] + kTwP + kTwC0 + 1 + 2 * cp;
                int16_t *out = L.bil[i] + 2 * cp;
                const int r0 = seg * RPS;
                constexpr int sh1 = BD - 6, off1 = 1 << (sh1 - 1);
                if constexpr (BD <= 10) {
                    // up to 10 bits the four variants are one formula (a zero fraction gives 16 x sample, (16 s + off1) >> sh1 =
                    // s << (10 - bd), (16 t + 8) >> 4 = t), and every intermediate fits 16 bits: packed arithmetic on the pair
                    auto PK = [](uint32_t v) { return __builtin_bit_cast(pk_u16, v); };
                    auto SP = [](int v) { return pk_u16{ (unsigned short)v, (unsigned short)v }; };
                    const pk_u16 MX = SP(mx), MX16 = SP(16 - mx), MY = SP(my), MY16 = SP(16 - my), OFF1 = SP(off1), EIGHT = SP(8);
                    auto hstage = [&](int r) -> pk_u16 {
                        const uint32_t p0 = *(const uint32_t *)(win + r * kTwP), p1 = *(const uint32_t *)(win + r * kTwP + 2);
                        return (PK(p0) * MX16 + PK(__builtin_amdgcn_alignbit(p1, p0, 16)) * MX + OFF1) >> SP(sh1);
                    };
                    pk_u16 a = hstage(r0);
#pragma unroll
                    for (int rr = 0; rr < RPS; rr++) {
                        const int r = r0 + rr;
                        if (r < PH) {
                            const pk_u16 b = hstage(r + 1);
                            const pk_u16 v = (a * MY16 + b * MY + EIGHT) >> SP(4);
                            *(uint32_t *)&out[r * PW] = __builtin_bit_cast(uint32_t, v);
                            a = b;
                        }
                    }
                } else {
                    // 12 bits: the v-only variant rounds once where the general form rounds twice — case analysis
                    const uint32_t hc = pack16(16 - mx, mx);
                    auto hstage = [&](int r, int &t0, int &t1) {
                        const uint32_t p0 = *(const uint32_t *)(win + r * kTwP), p1 = *(const uint32_t *)(win + r * kTwP + 2);
                        if (mx) {
                            t0 = (dot2(p0, hc, 0) + off1) >> sh1;
                            t1 = (dot2(__builtin_amdgcn_alignbit(p1, p0, 16), hc, 0) + off1) >> sh1;
                        } else {
                            t0 = p0 & 0xffff; t1 = p0 >> 16;
                        }
                    };
                    int a0, a1;
                    hstage(r0, a0, a1);
#pragma unroll 1
                    for (int rr = 0; rr < RPS; rr++) {
                        const int r = r0 + rr;
                        if (r >= PH)
                            break;
                        int v0, v1, b0 = 0, b1 = 0;
                        if (my)
                            hstage(r + 1, b0, b1);
                        if (mx && my)      { v0 = ((16 - my) * a0 + my * b0 + 8) >> 4;         v1 = ((16 - my) * a1 + my * b1 + 8) >> 4; }
                        else if (mx)       { v0 = a0;                                           v1 = a1; }
                        else if (my)       { v0 = ((16 - my) * a0 + my * b0 + off1) >> sh1;     v1 = ((16 - my) * a1 + my * b1 + off1) >> sh1; }
                        else               { v0 = (a0 + (1 << (BD - 11))) >> (BD - 10);         v1 = (a1 + (1 << (BD - 11))) >> (BD - 10); }
                        *(uint32_t *)&out[r * PW] = pack16(v0, v1);
                        if (my) { a0 = b0; a1 = b1; }
                        else hstage(min(r + 1, PH), a0, a1);
                    }
                }
            }
        }
        wave_sync();
        // inter.sad (vvcdsp.c:49) for all 25 offsets: lane -> (dy, even row 2 yr); the two plane rows sit in registers and the five
        // dx are whole-dword (even dx) or v_alignbit (odd dx) views of them
        constexpr int HR = H / 2, ND = PW / 2;
        const int sdy = lane / HR, syr = lane % HR;
        uint32_t cost[5] = { 0, 0, 0, 0, 0 };
        if (lane < 5 * HR) {
            const uint2 *pa = (const uint2 *)(L.bil[0] + (2 * syr + sdy) * PW), *pb = (const uint2 *)(L.bil[1] + (2 * syr + 4 - sdy) * PW);
            uint32_t a[ND], b[ND], a1[ND - 1], b1[ND - 1];
#pragma unroll
            for (int m = 0; m < ND / 2; m++) {
                const uint2 va = pa[m], vb = pb[m];
                a[2 * m] = va.x; a[2 * m + 1] = va.y; b[2 * m] = vb.x; b[2 * m + 1] = vb.y;
            }
#pragma unroll
            for (int m = 0; m < ND - 1; m++) {
                a1[m] = __builtin_amdgcn_alignbit(a[m + 1], a[m], 16);
                b1[m] = __builtin_amdgcn_alignbit(b[m + 1], b[m], 16);
            }
#pragma unroll
            for (int m = 0; m < W / 2; m++) {
                cost[0] = __builtin_amdgcn_sad_u16(a[m], b[m + 2], cost[0]);
                cost[1] = __builtin_amdgcn_sad_u16(a1[m], b1[m + 1], cost[1]);
                cost[2] = __builtin_amdgcn_sad_u16(a[m + 1], b[m + 1], cost[2]);
                cost[3] = __builtin_amdgcn_sad_u16(a1[m + 1], b1[m], cost[3]);
                cost[4] = __builtin_amdgcn_sad_u16(a[m + 2], b[m], cost[4]);
            }
        }
        // sum over the rows of a dy: butterflies inside the group of HR lanes (all of them end up with the total)
#pragma unroll
        for (int dx = 0; dx < 5; dx++) {
            int c = (int)cost[dx];
            c += dpp0<kDppXor1>(c);
            c += dpp0<kDppXor2>(c);
            if (HR == 8)
                c += dpp0<kDppHalfMirror>(c);
            cost[dx] = (uint32_t)c;
        }
        if (sdy == 2)
            cost[2] -= cost[2] >> 2;                     // the centre is compared at 3/4 of its cost (:712-713)
        // 8.5.3.4 array entry selection: the centre wins ties, then the earliest offset in scan order (dy outer, dx inner) =
        // the minimum of (cost, priority) keys
        uint32_t key = 0xffffffffu;
        if (lane < 5 * HR) {
#pragma unroll
            for (int dx = 0; dx < 5; dx++) {
                const uint32_t prio = (uint32_t)(sdy * 5 + dx + 1);
                key = min(key, (cost[dx] << 5) | ((dx == 2 && sdy == 2) ? 0u : prio));
            }
            if (syr == 0) {
#pragma unroll
                for (int dx = 0; dx < 5; dx++) L.sad[sdy * 5 + dx] = (int)cost[dx];
            }
        }
        uint32_t kmin = (uint32_t)__builtin_amdgcn_readlane((int)key, 0);
#pragma unroll
        for (int g = 1; g < 5; g++) kmin = min(kmin, (uint32_t)__builtin_amdgcn_readlane((int)key, g * HR));
        const int centre = __builtin_amdgcn_readlane((int)cost[2], 2 * HR);
        wave_sync();
        min_sad = centre;
        if (centre >= W * H) {
            searched = 1;
            min_sad = (int)(kmin >> 5);
            const int kk = kmin & 31, k = kk ? kk - 1 : 12;
            const int min_dy = k / 5, min_dx = k - min_dy * 5;
            int dmv0 = (min_dx - 2) * 16, dmv1 = (min_dy - 2) * 16;
            if (min_dx != 0 && min_dx != 4 && min_dy != 0 && min_dy != 4) {
                const int sc = __builtin_amdgcn_readfirstlane(L.sad[k]);
                dmv0 += parametric_mv_refine(__builtin_amdgcn_readfirstlane(L.sad[k - 1]), sc, __builtin_amdgcn_readfirstlane(L.sad[k + 1]));
                dmv1 += parametric_mv_refine(__builtin_amdgcn_readfirstlane(L.sad[k - 5]), sc, __builtin_amdgcn_readfirstlane(L.sad[k + 5]));
            }
            mv[0] = clip3(mv[0] + dmv0, -(1 << 17), (1 << 17) - 1);            // ff_vvc_clip_mv
            mv[1] = clip3(mv[1] + dmv1, -(1 << 17), (1 << 17) - 1);
            mv[2] = clip3(mv[2] - dmv0, -(1 << 17), (1 << 17) - 1);
            mv[3] = clip3(mv[3] - dmv1, -(1 << 17), (1 << 17) - 1);
        }
        if (min_sad < 2 * W * H)
            bdof = 0;
        // every lane holds the same refined motion: say so, and everything derived from it stays on the scalar unit
#pragma unroll
        for (int k = 0; k < 4; k++) mv[k] = __builtin_amdgcn_readfirstlane(mv[k]);
        bdof = __builtin_amdgcn_readfirstlane(bdof);
        min_sad = __builtin_amdgcn_readfirstlane(min_sad);
        wave_sync();                                     // bil / sad are dead from here on
    }
    {
        vvc355_bipred_result *rec = (vvc355_bipred_result *)job->rec;
        if (rec && lane == 0) {
#pragma unroll
            for (int k = 0; k < 4; k++) gst<int>(&rec->mv[k], mv[k]);
            gst<int>(&rec->bdof, bdof);
            gst<int>(&rec->min_sad, min_sad);
            gst<int>(&rec->searched, searched);
        }
    }

    // ------------------------------------------------------------------ I: 8-tap interpolation at the (refined) motion
    // window index of tap 0 of output (x, y) of reference i: (x + dxr[i], y + dyr[i]), clamped to the window
    int dxr[2], dyr[2], fxr[2], fyr[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        dxr[i] = (mv[2 * i] >> 4) - (job->mv[2 * i] >> 4);
        dyr[i] = (mv[2 * i + 1] >> 4) - (job->mv[2 * i + 1] >> 4);
        fxr[i] = mv[2 * i] & 15;
        fyr[i] = mv[2 * i + 1] & 15;
    }
    {
        // h pass: lanes 0..31 reference 0, lanes 32..63 reference 1.  Item = (row pair rp, column pair xp): five aligned dwords
        // q[0..9] of each row from the even LDS column cb = 2 xp + ((d + 5) & ~1).  d odd: tap 0 of output 2 xp is q[0]
        // (out0 = E . d[0..3], out1 = O . d[0..4]); d even: it is q[1] (out0 = O . d[0..4], out1 = E . d[1..4]).
        uint32_t t0s[2][5], t1s[2][5];
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const uint2 hf = load_uniform((const uint2 *)(d_tab_inter_luma_filters + (job->hf_idx * 16 + fxr[i]) * 8));
            const Taps8 t = make_taps8(hf.x, hf.y);
            const bool odd = dxr[i] & 1;
#pragma unroll
            for (int m = 0; m < 5; m++) {
                t0s[i][m] = odd ? (m < 4 ? t.e[m] : 0u) : t.o[m];
                t1s[i][m] = odd ? t.o[m] : (m > 0 ? t.e[m - 1] : 0u);
            }
        }
        const bool hi = lane >= 32;
        uint32_t t0[5], t1[5];
#pragma unroll
        for (int m = 0; m < 5; m++) { t0[m] = hi ? t0s[1][m] : t0s[0][m]; t1[m] = hi ? t1s[1][m] : t1s[0][m]; }
        const int l5 = lane & 31;
        constexpr int XPS = W / 2, NRP = (NR + 1) / 2, RPI = 32 / XPS, NK = (NRP + RPI - 1) / RPI;
        const int xp = l5 & (XPS - 1), rp0 = l5 / XPS;
        const int cb = 2 * xp + (((hi ? dxr[1] : dxr[0]) + 5) & ~1);
        const int dyl = hi ? dyr[1] : dyr[0];
        const uint8_t *wbase = (const uint8_t *)(hi ? L.win[1] : L.win[0]) + cb * 2;
        int16_t *tb = (hi ? L.tmpT[1] : L.tmpT[0]) + 2 * xp * kTtP + 2 * rp0;
#pragma unroll
        for (int k = 0; k < NK; k++) {
            const int rp = rp0 + RPI * k;
            if (NRP % RPI == 0 || rp < NRP) {
                int o[2][2];
#pragma unroll
                for (int rr = 0; rr < 2; rr++) {
                    const int rw = clip3(2 * rp + rr + dyl, 0, NR - 1);
                    const uint32_t *d = (const uint32_t *)(wbase + rw * (kTwP * 2));
                    int s0 = 0, s1 = 0;
#pragma unroll
                    for (int m = 0; m < 5; m++) {
                        const uint32_t q = d[m];
                        s0 = dot2(q, t0[m], s0);
                        s1 = dot2(q, t1[m], s1);
                    }
                    o[rr][0] = s0 >> (BD - 8);
                    o[rr][1] = s1 >> (BD - 8);
                }
                *(uint32_t *)(tb + 2 * RPI * k) = pack16(o[0][0], o[1][0]);
                *(uint32_t *)(tb + kTtP + 2 * RPI * k) = pack16(o[0][1], o[1][1]);
            }
        }
    }
    wave_sync();
    // v pass: lane -> column x = lane & 15, rows 4 by .. 4 by + 3 (by = lane >> 4), both references
    const int x = lane & 15, by = lane >> 4;
    const bool live = x < W && by < H / 4;
    int v[2][4];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const uint2 vf = load_uniform((const uint2 *)(d_tab_inter_luma_filters + (job->vf_idx * 16 + fyr[i]) * 8));
        const Taps8 t = make_taps8(vf.x, vf.y);
        const uint2 *p = (const uint2 *)(L.tmpT[i] + x * kTtP + 4 * by);
        uint32_t d[6];
#pragma unroll
        for (int m = 0; m < 3; m++) { const uint2 q = p[m]; d[2 * m] = q.x; d[2 * m + 1] = q.y; }
        int s[4] = { 0, 0, 0, 0 };
#pragma unroll
        for (int m = 0; m < 4; m++) { s[0] = dot2(d[m], t.e[m], s[0]); s[2] = dot2(d[m + 1], t.e[m], s[2]); }
#pragma unroll
        for (int m = 0; m < 5; m++) { s[1] = dot2(d[m], t.o[m], s[1]); s[3] = dot2(d[m + 1], t.o[m], s[3]); }
#pragma unroll
        for (int j = 0; j < 4; j++) v[i][j] = (int16_t)(s[j] >> 6);                  // put[..] stores int16
    }
    wave_sync();                                         // tmpT is dead: the BDOF planes overlay it

    uint8_t *dst = (uint8_t *)job->dst;
    const int dst_stride = job->dst_stride;
    const uint8_t *lut = (const uint8_t *)job->lmcs_lut;        // the tools path is luma only: an LMCS slice's forward map, or 0
    uint32_t doff = (uint32_t)(__mul24(4 * by, dst_stride) + x * ISZ);
    if (!bdof) {
        int shift, off;
        const int wf = job->weight_flag, w0 = job->w0, w1 = job->w1;
        if (!wf) { shift = max(3, 15 - BD); off = 1 << (shift - 1); }                                                   // avg
        else     { shift = job->denom + max(3, 15 - BD); off = (((job->o0 + job->o1) << (BD - 8)) + 1) << (shift - 1); } // w_avg
        if (live) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int p = wf ? (v[0][j] * w0 + v[1][j] * w1 + off) >> shift : (v[0][j] + v[1][j] + off) >> shift;
                gst_at<px_t>(dst, doff, (px_t)lmcs_fwd<BD>(lut, clip_px<BD>(p)));
                doff += dst_stride;
            }
        }
        return;
    }

    // ------------------------------------------------------------------ B: apply_bdof (vvc_inter_template.c:288)
    // smp[i][X = x + 1][e = y + 2]: row pairs (4 by, 4 by + 1), (4 by + 2, 4 by + 3) are aligned dwords at e = 4 by + 2, 4 by + 4
    uint32_t sp[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        sp[i][0] = pack16(v[i][0], v[i][1]);
        sp[i][1] = pack16(v[i][2], v[i][3]);
        if (live) {
            uint32_t *q = (uint32_t *)(L.smp[i] + (x + 1) * kSmP + 4 * by + 2);
            q[0] = sp[i][0];
            q[1] = sp[i][1];
        }
    }
    // ring (bdof_fetch_samples :101): position (X, y) is the integer sample (X - 1 + (fx >> 3), y + (fy >> 3)) of the refined block,
    // i.e. window index (d + 3 + ..): inside the window for |d| <= 2, no clamp.  Lanes 0..31 reference 0, 32..63 reference 1.
    {
        const int i = lane >> 5, l5 = lane & 31;
        const int u0 = (i ? dxr[1] + (fxr[1] >> 3) : dxr[0] + (fxr[0] >> 3)) + 3 + kTwC0;         // LDS column of block column 0
        const int v0r = (i ? dyr[1] + (fyr[1] >> 3) : dyr[0] + (fyr[0] >> 3)) + 3;                // window row of block row 0
        const uint16_t *wn = L.win[i];
        int16_t *sm = L.smp[i];
        if (l5 < W + 2) {            // top (y = -1) and bottom (y = H) rows: X = l5
            sm[l5 * kSmP + 1] = (int16_t)(wn[(v0r - 1) * kTwP + u0 + l5 - 1] << (14 - BD));
            sm[l5 * kSmP + H + 2] = (int16_t)(wn[(v0r + H) * kTwP + u0 + l5 - 1] << (14 - BD));
        }
        if (l5 < 2 * H) {            // left (X = 0) and right (X = W + 1) columns, H rows each
            const int side = l5 >= H, y = l5 - side * H;
            sm[(side ? W + 1 : 0) * kSmP + y + 2] = (int16_t)(wn[(v0r + y) * kTwP + u0 + (side ? W : -1)] << (14 - BD));
        }
    }
    wave_sync();
    // gradients on packed row pairs (prof_grad_filter :135): gh = (right >> 6) - (left >> 6), gv = (below >> 6) - (above >> 6)
    const pk_i16 S6 = pk_splat(6);
    pk_i16 gh[2][2], gvp[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const uint32_t *cl = (const uint32_t *)(L.smp[i] + x * kSmP + 4 * by);              // column x - 1
        const uint32_t *cc = (const uint32_t *)(L.smp[i] + (x + 1) * kSmP + 4 * by);
        const uint32_t *cr = (const uint32_t *)(L.smp[i] + (x + 2) * kSmP + 4 * by);
        const uint32_t l0 = cl[1], l1 = cl[2], r0 = cr[1], r1 = cr[2], up = cc[0], dn = cc[3];
        gh[i][0] = (as_pk(r0) >> S6) - (as_pk(l0) >> S6);
        gh[i][1] = (as_pk(r1) >> S6) - (as_pk(l1) >> S6);
        const pk_i16 a0 = as_pk(__builtin_amdgcn_alignbit(sp[i][0], up, 16)) >> S6;         // rows 4 by - 1, 4 by
        const pk_i16 a1 = as_pk(__builtin_amdgcn_alignbit(sp[i][1], sp[i][0], 16)) >> S6;   // rows 4 by + 1, 4 by + 2
        const pk_i16 a2 = as_pk(__builtin_amdgcn_alignbit(dn, sp[i][1], 16)) >> S6;         // rows 4 by + 3, 4 by + 4
        gvp[i][0] = a1 - a0;
        gvp[i][1] = a2 - a1;
    }
    wave_sync();                                         // every lane has read the ring: the windows are dead, the exchange overlays them
    // per row pair: D = (s0 >> 4) - (s1 >> 4), TH = (gh0 + gh1) >> 1, TV = (gv0 + gv1) >> 1, GHD = gh0 - gh1, GVD = gv0 - gv1, then the
    // five window terms |TH|, |TV|, sign(TV) TH, -sign(TH) D, -sign(TV) D (derive_bdof_vx_vy :237)
    const pk_i16 ONE = pk_splat(1), MONE = pk_splat(-1), ZERO = pk_splat(0), S4 = pk_splat(4);
    pk_i16 ghd[2], gvd[2], term[5][2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const pk_i16 D = (as_pk(sp[0][k]) >> S4) - (as_pk(sp[1][k]) >> S4);
        const pk_i16 th = (gh[0][k] + gh[1][k]) >> ONE, tv = (gvp[0][k] + gvp[1][k]) >> ONE;
        ghd[k] = gh[0][k] - gh[1][k];
        gvd[k] = gvp[0][k] - gvp[1][k];
        const pk_i16 nth = ZERO - th, ntv = ZERO - tv;
        const pk_i16 nsx = pk_max(pk_min(nth, ONE), MONE), nsy = pk_max(pk_min(ntv, ONE), MONE);     // -sign(th), -sign(tv)
        term[0][k] = pk_max(th, nth);
        term[1][k] = pk_max(tv, ntv);
        term[2][k] = (ZERO - nsy) * th;
        term[3][k] = nsx * D;
        term[4][k] = nsy * D;
    }
    // vertical box sums: rows 4 by - 1 .. 4 by + 4 of column x, rows outside the block replicate the edge row.  Own four rows in
    // registers; the row above is the high half of the upper neighbour's second pair, the row below the low half of the lower
    // neighbour's first pair — or, at the block's top / bottom, the lane's own edge row.
    uint32_t *xch = (uint32_t *)&L.win[0][0];
    static_assert(kXchBytes <= sizeof(L.win), "exchange buffer overlays the windows");
    constexpr int NBY = H / 4;
    if (live) {
#pragma unroll
        for (int q = 0; q < 5; q++)
            *(uint2 *)&xch[((q * 16 + x) * 4 + by) * 2] = make_uint2(as_u32(term[q][0]), as_u32(term[q][1]));
    }
    wave_sync();
    int tot[5];
    {
        const bool top = by == 0, bot = by == NBY - 1;
        const uint32_t *pu = &xch[(x * 4 + by) * 2 + (top ? 0 : -1)], *pd = &xch[(x * 4 + by) * 2 + (bot ? 1 : 2)];
        const uint32_t su = top ? 0x00000001u : 0x00010000u, sd = bot ? 0x00010000u : 0x00000001u;
#pragma unroll
        for (int q = 0; q < 5; q++) {
            const uint32_t own = as_u32(term[q][0] + term[q][1]);
            int t = dot2(own, 0x00010001u, 0);
            t = dot2(pu[q * 128], su, t);
            t = dot2(pd[q * 128], sd, t);
            tot[q] = t;
        }
    }
    // horizontal box sums inside the 16-lane row: the window of sub-block bx is columns 4 bx - 1 .. 4 bx + 4 with the edge
    // columns doubled (ring replication) and nothing from beyond the block; the sum lands in lane 4 bx + 2
    {
        const int e = (x == 0 || x == W - 1) ? 1 : 0;
        // (Round 2 kept this block between two __builtin_amdgcn_sched_barrier(0) after a build in which lane 15's right-looking pair read
        // zero.  Round 3 could not reproduce it: the ISA with and without the fences differs only in where the hazard recogniser puts the
        // VALU-write -> DPP-read wait states (s_nop), present in both, and the unfenced library passes every bi-prediction parity test
        // including test_bdof_rightmost_subblock — so the failure belonged to that intermediate source state, not to the instruction mix.)
#pragma unroll
        for (int q = 0; q < 5; q++) {
            int p = tot[q] << e;
            if (W < 16)
                p = x < W ? p : 0;
            // pairs (4 bx - 1, 4 bx), (4 bx + 1, 4 bx + 2) from the left-looking sums, (4 bx + 3, 4 bx + 4) from the right-looking one:
            // no pair is anchored on a lane outside the row
            const int tl = p + dpp0<kDppShr1>(p), tr = p + dpp0<kDppShl1>(p);
            tot[q] = tl + dpp0<kDppShr2>(tl) + dpp0<kDppShl1>(tr);
        }
    }
    const int sgx2 = tot[0], sgy2 = tot[1], sgxgy = tot[2], sgxdi = tot[3], sgydi = tot[4];
    const int vx = sgx2 > 0 ? clip3((sgxdi * 4) >> ilog2(sgx2), -15, 15) : 0;
    const int vy = sgy2 > 0 ? clip3(((sgydi * 4) - ((vx * sgxgy) >> 1)) >> ilog2(sgy2), -15, 15) : 0;
    const uint32_t vxy = (uint32_t)dpp0<kDppQuad2>((int)pack16(vx, vy));                      // lane 4 bx + 2 -> its quad
    // apply_bdof_min_block (:267): (s0 + s1 + off + vx ghd + vy gvd) >> (15 - bd)
    if (live) {
        constexpr int sh = 15 - BD, off = 1 << (sh - 1);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t hd = as_u32(ghd[j >> 1]), vd = as_u32(gvd[j >> 1]);
            const uint32_t g = (j & 1) ? __builtin_amdgcn_perm(vd, hd, 0x07060302u) : __builtin_amdgcn_perm(vd, hd, 0x05040100u);
            const int p = dot2(g, vxy, v[0][j] + v[1][j] + off) >> sh;
            gst_at<px_t>(dst, doff, (px_t)lmcs_fwd<BD>(lut, clip_px<BD>(p)));
            doff += dst_stride;
        }
    }
}
