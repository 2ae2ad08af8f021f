// The decoder's per-unit side tables, filled on the device from compact per-unit records.
//
// The reference's parser writes VVCFrameContext.tab (vvcdec.h:122-187) one minimum unit at a time on the host: set_cb_pos / set_cb_tab
// per coding unit (vvc_ctu.c:124-140, :1144-1160, :1230-1250), set_tb_pos / set_tb_tab per transform block (:41-75, :395-400, :511),
// ff_vvc_set_mvf and friends per prediction unit or sub-block (vvc_mvs.c).  A frame-resident backend needs those tables in HBM for the
// boundary-strength pass, the deblocking parameter derivation and the inter stage driver; uploading them as they are costs 24 + ~60 bytes
// per 4x4 luma unit (180 MB of an 8K picture).  Here the host hands over what the parser knows per unit — 8 bytes per coding unit,
// 8 per transform unit, 32 per rectangle of equal motion — and one launch writes the tables: a wave per record, lanes over the record's
// 4x4 units, coalesced row segments.
#include "common.hpp"
#include "runtime.hpp"
#include "../../include/vvc_mi355.h"

namespace vvc355 {

// records per workgroup: 4 waves, one record each
template <typename REC, typename F>
__device__ __forceinline__ void for_units(const REC &r, int pitch, int lane, F body)
{
    const int ux = r.x0 >> 2, uy = r.y0 >> 2, uw = r.w >> 2, uh = r.h >> 2;          // 4x4 luma units; w, h are multiples of 4
    const int n = uw * uh;
    for (int i = lane; i < n; i += 64) {
        const int dy = i / uw, dx = i - dy * uw;
        body((uy + dy) * pitch + ux + dx);
    }
}

__global__ __launch_bounds__(256) void tabfill_kernel(const vvc355_tab_fill *__restrict__ fp)
{
    const vvc355_tab_fill f = load_uniform(fp);
    const int rec = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int pitch = f.unit_pitch;
    if (rec < f.n_cu) {
        const vvc355_cu_rec r = load_uniform((const vvc355_cu_rec *)f.cu + rec);
        int *cbx = (int *)f.cb_pos_x, *cby = (int *)f.cb_pos_y;
        uint8_t *cbw = (uint8_t *)f.cb_width, *cbh = (uint8_t *)f.cb_height, *msf = (uint8_t *)f.msf, *iaf = (uint8_t *)f.iaf;
        for_units(r, pitch, lane, [&](int u) {
            gst<int>(cbx + u, r.x0); gst<int>(cby + u, r.y0);
            gst<uint8_t>(cbw + u, r.w); gst<uint8_t>(cbh + u, r.h);
            gst<uint8_t>(msf + u, (uint8_t)(r.flags & 1)); gst<uint8_t>(iaf + u, (uint8_t)((r.flags >> 1) & 1));
        });
        return;
    }
    const int t = rec - f.n_cu;
    if (t < f.n_tu) {
        const vvc355_tu_rec r = load_uniform((const vvc355_tu_rec *)f.tu + t);
        const int tree = r.flags >> 7;
        int *tbx = (int *)f.tb_pos_x0[tree], *tby = (int *)f.tb_pos_y0[tree];
        uint8_t *tbw = (uint8_t *)f.tb_width[tree], *tbh = (uint8_t *)f.tb_height[tree], *pcm = (uint8_t *)f.pcmf[tree];
        const uint8_t wv = (uint8_t)(tree ? r.w >> f.hs : r.w), hv = (uint8_t)(tree ? r.h >> f.vs : r.h);      // in samples of the tree's component
        if (!tree) {
            uint8_t *cbf0 = (uint8_t *)f.tu_coded_flag[0];
            for_units(r, pitch, lane, [&](int u) {
                gst<int>(tbx + u, r.x0); gst<int>(tby + u, r.y0); gst<uint8_t>(tbw + u, wv); gst<uint8_t>(tbh + u, hv);
                gst<uint8_t>(cbf0 + u, (uint8_t)(r.flags & 1)); gst<uint8_t>(pcm + u, (uint8_t)((r.flags >> 4) & 1));
            });
        } else {
            uint8_t *cbf1 = (uint8_t *)f.tu_coded_flag[1], *cbf2 = (uint8_t *)f.tu_coded_flag[2], *jnt = (uint8_t *)f.tu_joint_cbcr;
            for_units(r, pitch, lane, [&](int u) {
                gst<int>(tbx + u, r.x0); gst<int>(tby + u, r.y0); gst<uint8_t>(tbw + u, wv); gst<uint8_t>(tbh + u, hv);
                gst<uint8_t>(cbf1 + u, (uint8_t)((r.flags >> 1) & 1)); gst<uint8_t>(cbf2 + u, (uint8_t)((r.flags >> 2) & 1));
                gst<uint8_t>(jnt + u, (uint8_t)((r.flags >> 3) & 1)); gst<uint8_t>(pcm + u, (uint8_t)((r.flags >> 4) & 1));
            });
        }
        return;
    }
    const int m = t - f.n_tu;
    if (m < f.n_mv) {
        const vvc355_mv_rec r = load_uniform((const vvc355_mv_rec *)f.mv + m);
        uint4 lo;
        uint2 hi;
        __builtin_memcpy(&lo, &r.mvf, 16);
        __builtin_memcpy(&hi, (const uint8_t *)&r.mvf + 16, 8);
        uint8_t *tab = (uint8_t *)f.mvf;
        const int mp = f.mvf_pitch;
        const int ux = r.x0 >> 2, uy = r.y0 >> 2, uw = r.w >> 2, n = uw * (r.h >> 2);
        for (int i = lane; i < n; i += 64) {
            const int dy = i / uw, dx = i - dy * uw;
            uint8_t *e = tab + (size_t)((uy + dy) * mp + ux + dx) * 24;
            gst<uint2>(e, make_uint2(lo.x, lo.y)); gst<uint2>(e + 8, make_uint2(lo.z, lo.w)); gst<uint2>(e + 16, hi);        // 24-byte entries: 8-byte aligned
        }
    }
}

} // namespace vvc355

extern "C" void vvc355_tab_fill_pass(void *stream, const vvc355_tab_fill *frame_dev, const vvc355_tab_fill *frame_host)
{
    const int n = frame_host->n_cu + frame_host->n_tu + frame_host->n_mv;
    if (n <= 0) return;
    hipLaunchKernelGGL(vvc355::tabfill_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, frame_dev);
    HIP_CHECK(hipGetLastError());
}
