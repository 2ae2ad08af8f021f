// The decoder's per-unit side tables, filled on the device from compact per-unit records.
//
// The reference's parser writes VVCFrameContext.tab (vvcdec.h:122-187) one minimum unit at a time on the host: set_cb_pos / set_cb_tab
// per coding unit (vvc_ctu.c:124-140, :1144-1160, :1230-1250), set_tb_pos / set_tb_tab per transform block (:41-75, :395-400, :511),
// ff_vvc_set_mvf and friends per prediction unit or sub-block (vvc_mvs.c).  A frame-resident backend needs those tables in HBM for the
// boundary-strength pass, the deblocking parameter derivation and the inter stage driver; uploading them as they are costs 24 + ~60 bytes
// per 4x4 luma unit (180 MB of an 8K picture).  Here the host hands over what the parser knows per unit — 8 bytes per coding unit,
// 8 per transform unit, 32 per rectangle of equal motion, grouped per CTU like the parser produces them — and one launch writes the tables.
//
// One workgroup per CTU.  Phase 1 inverts the records: each record (16 lanes per record) writes its index into an LDS map of the CTU's
// 4x4 units, one map per table family (coding unit, transform unit of each tree, motion).  Phase 2 walks the CTU's units in raster order,
// one lane per unit: the unit's record comes through the map (an 8- or 32-byte load that neighbouring lanes share) and every table gets
// its entry — rows of 32 consecutive units, so the byte tables go out as 32-byte segments, the int tables as 128-byte lines and the
// MvField table as 768-byte runs.
#include "common.hpp"
#include "runtime.hpp"
#include "../../include/vvc_mi355.h"

namespace vvc355 {

static constexpr int kMaxUnits = 32 * 32;          // 128x128 CTU in 4x4 units
static constexpr uint16_t kNoRec = 0xffff;

// records [first, last) of one kind -> map[unit within the CTU] = record index - first.  `map_tree1` != 0: the records carry a tree bit
// (flags bit 7) and those of tree 1 go to that map.  Sixteen lanes per record; widths are powers of two in every partitioning a decoder
// produces (the general case keeps the division).
template <typename REC>
__device__ __forceinline__ void map_records(uint16_t *map, uint16_t *map_tree1, uint2 *heads, const REC *recs, int first, int last, int ox, int oy, int lw)
{
    const int sub = threadIdx.x & 15;
    // the records' heads (x0 y0 | w h flags pad: the same 8 bytes for all three record kinds) come in through LDS, 1024 at a time with one
    // coalesced round trip, so that the painting passes below (sixteen records per pass) do not each wait for a global load
    for (int c0 = first; c0 < last; c0 += kMaxUnits) {
        const int nc = min(kMaxUnits, last - c0);
        __syncthreads();
        for (int i = threadIdx.x; i < nc; i += 256)
            heads[i] = gld<uint2>(recs + c0 + i);
        __syncthreads();
        for (int k = threadIdx.x >> 4; k < nc; k += 16) {
            const uint2 head = heads[k];
            const int r = c0 + k;
            const int x0 = (int16_t)(head.x & 0xffff), y0 = (int16_t)(head.x >> 16), w = head.y & 0xff, h = (head.y >> 8) & 0xff, flags = (head.y >> 16) & 0xff;
            uint16_t *m = (map_tree1 && (flags >> 7)) ? map_tree1 : map;
            const int ux = (x0 - ox) >> 2, uy = (y0 - oy) >> 2, uw = w >> 2, n = uw * (h >> 2);
            const int base = (uy << lw) + ux;
            if ((uw & (uw - 1)) == 0) {
                const int lg = __builtin_ctz(uw | 64);
                for (int i = sub; i < n; i += 16)
                    m[base + ((i >> lg) << lw) + (i & (uw - 1))] = (uint16_t)(r - first);
            } else {
                for (int i = sub; i < n; i += 16) {
                    const int dy = i / uw, dx = i - dy * uw;
                    m[base + (dy << lw) + dx] = (uint16_t)(r - first);
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void tabfill_kernel(const vvc355_tab_fill *__restrict__ fp)
{
    __shared__ uint16_t map[4][kMaxUnits];             // coding unit, transform unit tree 0, tree 1, motion
    __shared__ uint2 heads[kMaxUnits];
    const vvc355_tab_fill f = load_uniform(fp);
    const int rs = blockIdx.x, ry = rs / f.ctb_width, rx = rs - ry * f.ctb_width;
    const int lw = f.ctb_log2 - 2, side = 1 << lw, n_units = side * side;
    const int ox = rx << f.ctb_log2, oy = ry << f.ctb_log2;
    for (int i = threadIdx.x; i < 4 * kMaxUnits / 2; i += 256)
        ((uint32_t *)map)[i] = 0xffffffffu;
    __syncthreads();
    const int *fcu = (const int *)f.ctu_first_cu, *ftu = (const int *)f.ctu_first_tu, *fmv = (const int *)f.ctu_first_mv;
    const int cu0 = fcu ? gld<int>(fcu + rs) : 0, cu1 = fcu ? gld<int>(fcu + rs + 1) : 0;
    const int tu0 = ftu ? gld<int>(ftu + rs) : 0, tu1 = ftu ? gld<int>(ftu + rs + 1) : 0;
    const int mv0 = fmv ? gld<int>(fmv + rs) : 0, mv1 = fmv ? gld<int>(fmv + rs + 1) : 0;
    const vvc355_cu_rec *cus = (const vvc355_cu_rec *)f.cu;
    const vvc355_tu_rec *tus = (const vvc355_tu_rec *)f.tu;
    const vvc355_mv_rec *mvs = (const vvc355_mv_rec *)f.mv;
    map_records(map[0], (uint16_t *)nullptr, heads, cus, cu0, cu1, ox, oy, lw);
    map_records(map[1], map[2], heads, tus, tu0, tu1, ox, oy, lw);
    map_records(map[3], (uint16_t *)nullptr, heads, mvs, mv0, mv1, ox, oy, lw);          // (motion records are 32 bytes: the head is their first 8)
    __syncthreads();
    const int pw = f.width >> 2, ph = f.height >> 2;                // picture size in units
    for (int i = threadIdx.x; i < n_units; i += 256) {
        const int dy = i >> lw, dx = i & (side - 1);
        const int gx = (ox >> 2) + dx, gy = (oy >> 2) + dy;
        if (gx >= pw || gy >= ph)
            continue;
        const int u = gy * f.unit_pitch + gx;
        const uint16_t ic = map[0][i], i0 = map[1][i], i1 = map[2][i], im = map[3][i];
        if (ic != kNoRec) {
            const vvc355_cu_rec r = gld<vvc355_cu_rec>(cus + cu0 + ic);
            gst<int>((int *)f.cb_pos_x + u, r.x0); gst<int>((int *)f.cb_pos_y + u, r.y0);
            gst<uint8_t>((uint8_t *)f.cb_width + u, r.w); gst<uint8_t>((uint8_t *)f.cb_height + u, r.h);
            gst<uint8_t>((uint8_t *)f.msf + u, (uint8_t)(r.flags & 1)); gst<uint8_t>((uint8_t *)f.iaf + u, (uint8_t)((r.flags >> 1) & 1));
        }
        if (i0 != kNoRec) {
            const vvc355_tu_rec r = gld<vvc355_tu_rec>(tus + tu0 + i0);
            gst<int>((int *)f.tb_pos_x0[0] + u, r.x0); gst<int>((int *)f.tb_pos_y0[0] + u, r.y0);
            gst<uint8_t>((uint8_t *)f.tb_width[0] + u, r.w); gst<uint8_t>((uint8_t *)f.tb_height[0] + u, r.h);
            gst<uint8_t>((uint8_t *)f.tu_coded_flag[0] + u, (uint8_t)(r.flags & 1)); gst<uint8_t>((uint8_t *)f.pcmf[0] + u, (uint8_t)((r.flags >> 4) & 1));
        }
        if (i1 != kNoRec) {
            const vvc355_tu_rec r = gld<vvc355_tu_rec>(tus + tu0 + i1);
            gst<int>((int *)f.tb_pos_x0[1] + u, r.x0); gst<int>((int *)f.tb_pos_y0[1] + u, r.y0);
            gst<uint8_t>((uint8_t *)f.tb_width[1] + u, (uint8_t)(r.w >> f.hs)); gst<uint8_t>((uint8_t *)f.tb_height[1] + u, (uint8_t)(r.h >> f.vs));      // in chroma samples
            gst<uint8_t>((uint8_t *)f.tu_coded_flag[1] + u, (uint8_t)((r.flags >> 1) & 1)); gst<uint8_t>((uint8_t *)f.tu_coded_flag[2] + u, (uint8_t)((r.flags >> 2) & 1));
            gst<uint8_t>((uint8_t *)f.tu_joint_cbcr + u, (uint8_t)((r.flags >> 3) & 1)); gst<uint8_t>((uint8_t *)f.pcmf[1] + u, (uint8_t)((r.flags >> 4) & 1));
        }
        if (im != kNoRec) {
            const uint8_t *src = (const uint8_t *)(mvs + mv0 + im) + 8;
            const uint2 a = gld<uint2>(src), b = gld<uint2>(src + 8), c = gld<uint2>(src + 16);
            uint8_t *e = (uint8_t *)f.mvf + (size_t)(gy * f.mvf_pitch + gx) * 24;
            gst<uint2>(e, a); gst<uint2>(e + 8, b); gst<uint2>(e + 16, c);
        }
    }
}

} // namespace vvc355

extern "C" void vvc355_tab_fill_pass(void *stream, const vvc355_tab_fill *frame_dev, const vvc355_tab_fill *frame_host)
{
    const int n = frame_host->ctb_width * frame_host->ctb_height;
    if (n <= 0 || frame_host->n_cu + frame_host->n_tu + frame_host->n_mv <= 0) return;
    if (frame_host->ctb_log2 < 5 || frame_host->ctb_log2 > 7) {
        fprintf(stderr, "vvc_mi355: vvc355_tab_fill_pass: CTB log2 size %d outside 5..7\n", frame_host->ctb_log2);
        abort();
    }
    hipLaunchKernelGGL(vvc355::tabfill_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, frame_dev);
    HIP_CHECK(hipGetLastError());
}
