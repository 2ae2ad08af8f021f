// Affine motion compensation with PROF for gfx950: the luma work of pred_affine_blk (libavcodec/vvc/vvc_inter.c:864-897) for
// one 4x4 sub-block per job — luma_prof_uni (:369-406) / luma_prof_bi (:408-447): interpolation with the affine filter set
// (ff_vvc_inter_luma_filters[2], h2656_inter_template.c:29-340), edge emulation to the picture (:33-59, by clamped reads),
// and where cb_prof_flag is set fetch_samples (vvc_inter_template.c:130) + apply_prof / apply_prof_uni / apply_prof_uni_w
// (:160-235), then put_uni / put_uni_w rounding or avg / w_avg (:25-58) straight to pixels.
//
// Mapping: 16 lanes per sub-block (one per sample), four sub-blocks per wave, wave-level synchronisation only.  Per reference
// an 11x11 window goes to LDS, the separable 8-tap passes run from there, and the 14-bit prediction with its integer-sample
// ring sits in a 6x6 LDS plane for the PROF gradients.
#include "common.hpp"
#include "runtime.hpp"
#include "../../include/vvc_mi355.h"

namespace vvc355 {

#define VVC355_TABLE(type, name, count) __device__ static const type a_tab_##name[count]
#include "tables.inc"
#undef VVC355_TABLE

static constexpr int kAwP = 12;          // window pitch (11 columns used)

struct AffineLds {
    uint16_t win[11 * kAwP];             // samples (-3 .. 7) x (-3 .. 7) around the block
    int16_t th[11 * 4];                  // horizontal pass output, 11 rows x 4 columns
    int16_t pp[2][6 * 6];                // per list: prediction (14-bit) with its ring, (1, 1) = block origin
};

__device__ __forceinline__ void group16_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <int BD>
__global__ __launch_bounds__(256) void affine_kernel(const vvc355_affine_job *__restrict__ jobs, int n_jobs)
{
    using px_t = typename Px<BD>::type;
    __shared__ __attribute__((aligned(16))) AffineLds lds_all[16];
    const int ji = blockIdx.x * 16 + (threadIdx.x >> 4), l = threadIdx.x & 15;
    if (ji >= n_jobs)
        return;                                          // whole 16-lane groups leave together
    AffineLds &L = lds_all[threadIdx.x >> 4];
    const vvc355_affine_job job = jobs[ji];
    const int x = l & 3, y = l >> 2;
    const int16_t *dmv = (const int16_t *)job.diff_mv;
    const bool bi = job.pred_flag == 3;
    int val[2] = { 0, 0 };

#pragma unroll
    for (int i = 0; i < 2; i++) {
        if (!(job.pred_flag & (1 << i)))
            continue;
        const int mvx = job.mv[2 * i], mvy = job.mv[2 * i + 1];
        const int mx = mvx & 15, my = mvy & 15;
        const int ox = job.x + (mvx >> 4), oy = job.y + (mvy >> 4);
        const uint8_t *plane = (const uint8_t *)(i ? job.ref1 : job.ref0);
        const int stride = i ? job.ref1_stride : job.ref0_stride;
        const bool prof = i ? job.prof1 : job.prof0;
        // ---- 11 x 11 window at clamped coordinates (emulated_edge, vvc_inter.c:33-59)
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const int e = l + 16 * it;
            if (e < 121) {
                const int r = e / 11, c = e - r * 11;
                const int xa = clip3(ox - 3 + c, 0, job.pic_w - 1), ya = clip3(oy - 3 + r, 0, job.pic_h - 1);
                L.win[r * kAwP + c] = (uint16_t)gld<px_t>(plane + (ptrdiff_t)ya * stride + xa * (int)sizeof(px_t));
            }
        }
        group16_sync();
        const int8_t *hf = a_tab_inter_luma_filters + (2 * 16 + mx) * 8, *vf = a_tab_inter_luma_filters + (2 * 16 + my) * 8;
        // ---- put[LUMA][..][!!my][!!mx] (h2656_inter_template.c:29, :97, :112, :127)
        int p;
        if (mx) {
            // horizontal pass on the rows the vertical pass reads (all 11, or the block's 4)
            const int r0 = my ? 0 : 3, nr = my ? 11 : 4;
            for (int e = l; e < nr * 4; e += 16) {
                const int r = r0 + (e >> 2), c = e & 3;
                int s = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) s += hf[k] * (int)L.win[r * kAwP + c + k];
                L.th[r * 4 + c] = (int16_t)(s >> (BD - 8));
            }
            group16_sync();
            if (my) {
                int s = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) s += vf[k] * (int)L.th[(y + k) * 4 + x];
                p = s >> 6;
            } else {
                p = L.th[(y + 3) * 4 + x];
            }
        } else if (my) {
            int s = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) s += vf[k] * (int)L.win[(y + k) * kAwP + x + 3];
            p = s >> (BD - 8);
        } else {
            p = (int)L.win[(y + 3) * kAwP + x + 3] << (14 - BD);
        }
        p = (int16_t)p;                                   // put stores int16
        if (prof) {
            // ---- fetch_samples (vvc_inter_template.c:130): ring position (rx, ry) in -1 .. 4 reads the integer sample at
            // (rx + (mx >> 3), ry + (my >> 3)) of the block; then the PROF gradients and refinement (:135, :160-235)
            int16_t *pp = L.pp[i];
            pp[(y + 1) * 6 + x + 1] = (int16_t)p;
            for (int e = l; e < 20; e += 16) {
                int rx, ry;
                if (e < 6)       { ry = -1; rx = e - 1; }
                else if (e < 12) { ry = 4;  rx = e - 7; }
                else             { const int k = e - 12; ry = k >> 1; rx = (k & 1) ? 4 : -1; }
                const int s = L.win[(ry + (my >> 3) + 3) * kAwP + rx + (mx >> 3) + 3];
                pp[(ry + 1) * 6 + rx + 1] = (int16_t)(s << (14 - BD));
            }
            group16_sync();
            const int o = (y + 1) * 6 + x + 1;
            const int g_h = (int16_t)((pp[o + 1] >> 6) - (pp[o - 1] >> 6));
            const int g_v = (int16_t)((pp[o + 6] >> 6) - (pp[o - 6] >> 6));
            const int limit = 1 << max(13, BD + 1);
            const int di = g_h * (int)dmv[i * 32 + l] + g_v * (int)dmv[i * 32 + 16 + l];
            p = p + clip3(di, -limit, limit - 1);
        }
        val[i] = p;
        group16_sync();                                   // win / th are reused by the second list
        if (!bi) {
            // ---- uni-prediction output: put_uni / put_uni_w (h2656_inter_template.c:44, :60) or apply_prof_uni(_w)
            uint8_t *drow = (uint8_t *)job.dst + (ptrdiff_t)y * job.dst_stride;
            int out;
            if (!job.weight_flag) {
                const int sh = 14 - BD;
                out = (p + (1 << (sh - 1))) >> sh;
            } else {
                const int sh = job.denom + (prof ? max(2, 14 - BD) : 14 - BD);
                out = ((p * job.w0 + (1 << (sh - 1))) >> sh) + job.o0 * (1 << (BD - 8));
            }
            st_px<BD>(drow, x, lmcs_fwd<BD>((const uint8_t *)job.lmcs_lut, clip_px<BD>(out)));
        }
    }
    if (bi) {
        // apply_prof / put leave int16 operands; avg / w_avg (vvc_inter_template.c:25, :42)
        const int a = (int16_t)val[0], b = (int16_t)val[1];
        uint8_t *drow = (uint8_t *)job.dst + (ptrdiff_t)y * job.dst_stride;
        int out;
        if (!job.weight_flag) {
            const int sh = max(3, 15 - BD);
            out = (a + b + (1 << (sh - 1))) >> sh;
        } else {
            const int sh = job.denom + max(3, 15 - BD);
            out = (a * job.w0 + b * job.w1 + ((((job.o0 + job.o1) << (BD - 8)) + 1) << (sh - 1))) >> sh;
        }
        st_px<BD>(drow, x, lmcs_fwd<BD>((const uint8_t *)job.lmcs_lut, clip_px<BD>(out)));
    }
}

} // namespace vvc355

extern "C" void vvc355_affine_batch(void *stream, int bd, const vvc355_affine_job *jobs_dev, int n_jobs)
{
    using namespace vvc355;
    if (n_jobs <= 0) return;
    VVC355_BD_DISPATCH(bd, hipLaunchKernelGGL((affine_kernel<BD>), dim3((n_jobs + 15) / 16), dim3(256), 0, (hipStream_t)stream, jobs_dev, n_jobs));
    HIP_CHECK(hipGetLastError());
}
