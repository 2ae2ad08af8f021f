// Inter prediction stage driver for gfx950: the sub-block job arrays of a whole picture, written on the device from the decoder's
// MvField table, reference-picture lists, prediction weight tables and a list of coding units, then the fused sub-block kernels of
// mc_fused.hip.  Reference behaviour: pred_regular_blk and its helpers, libavcodec/vvc/vvc_inter.c:129-177 (derive_weight_uni,
// derive_weight), :764-813 (derive_sb_mv, the sub-block walk), pred_regular_luma / _chroma (:549-640: filter set, uni / bi split).
#include "common.hpp"
#include "runtime.hpp"
#include "../../include/vvc_mi355.h"

namespace vvc355 {

struct MvFieldDev { int32_t mv[2][2]; int8_t ref_idx[2]; uint8_t hpel_if_idx, bcw_idx, pred_flag, ciip_flag, pad_[2]; };
static_assert(sizeof(MvFieldDev) == 24, "MvField layout (vvc_ctu.h:195-202)");

// one thread per coding unit: its sub-blocks, each in 16x16 tiles, one luma and two chroma jobs per tile
__global__ __launch_bounds__(256) void inter_build_kernel(const vvc355_inter_frame *__restrict__ fp)
{
    const vvc355_inter_frame f = load_uniform(fp);
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= f.n_pus)
        return;
    const vvc355_inter_pu pu = ((const vvc355_inter_pu *)f.pus)[u];
    const vvc355_inter_slice *sl = (const vvc355_inter_slice *)f.slices + pu.slice;
    const vvc355_ref_pic *refs = (const vvc355_ref_pic *)f.refs;
    const MvFieldDev *mvf_tab = (const MvFieldDev *)f.mvf;
    vvc355_bipred_job *jl = (vvc355_bipred_job *)f.jobs_luma, *jc = (vvc355_bipred_job *)f.jobs_chroma;
    vvc355_bipred_result *rec = (vvc355_bipred_result *)f.records;
    const int sbw = pu.cb_width / pu.num_sb_x, sbh = pu.cb_height / pu.num_sb_y;
    const int tw = min(sbw, 16), th = min(sbh, 16);
    const int bcw_w_lut[5] = { 4, 5, 3, 10, -2 };                       // vvc_inter.c:29
    uint32_t job = pu.first_job;
    for (int sby = 0; sby < pu.num_sb_y; sby++)
        for (int sbx = 0; sbx < pu.num_sb_x; sbx++) {
            const int sx = pu.x0 + sbx * sbw, sy = pu.y0 + sby * sbh;
            const MvFieldDev mv = mvf_tab[(sy >> 2) * f.mvf_stride + (sx >> 2)];              // ff_vvc_get_mvf
            const bool bi = mv.pred_flag == 3;
            for (int ty = 0; ty < sbh; ty += th)
                for (int tx = 0; tx < sbw; tx += tw, job++) {
                    for (int c = 0; c < (f.chroma_format_idc ? 3 : 1); c++) {
                        const int hs = c ? f.hs : 0, vs = c ? f.vs : 0;
                        vvc355_bipred_job j = {};
                        const int x = (sx + tx) >> hs, y = (sy + ty) >> vs;
                        j.dst = f.dst[c] + (uint64_t)y * f.dst_stride[c] + ((uint64_t)x << f.pixel_shift);
                        j.dst_stride = f.dst_stride[c];
                        for (int l = 0; l < 2; l++) {
                            if (!(mv.pred_flag & (1 << l)))
                                continue;
                            const vvc355_ref_pic rp = refs[l * 16 + mv.ref_idx[l]];
                            (l ? j.ref1 : j.ref0) = rp.plane[c];
                            (l ? j.ref1_stride : j.ref0_stride) = rp.stride[c];
                            j.mv[2 * l] = mv.mv[l][0];
                            j.mv[2 * l + 1] = mv.mv[l][1];
                        }
                        j.rec = (uint64_t)(rec + job);
                        j.x = (int16_t)x; j.y = (int16_t)y; j.w = (int16_t)(tw >> hs); j.h = (int16_t)(th >> vs);
                        j.pic_w = (int16_t)(f.width >> hs); j.pic_h = (int16_t)(f.height >> vs);
                        j.chroma = c > 0; j.hs = f.hs; j.vs = f.vs;
                        j.dmvr = bi && pu.dmvr_flag;
                        j.bdof = !c && bi && pu.bdof_flag;
                        j.hf_idx = j.vf_idx = c ? 0 : pu.hpel_if_idx;
                        j.pred_flag = mv.pred_flag;
                        // predict_inter's lmcs.filter (:888-891): luma of the coding unit goes through the forward map, CIIP units excepted
                        j.lmcs_lut = (!c && sl->lmcs_used && !pu.ciip_flag) ? f.lmcs_fwd_lut : 0;
                        if (bi) {
                            // derive_weight (:149-177)
                            const int weight_flag = sl->weighted_pred || (sl->weighted_bipred && !pu.dmvr_flag);
                            if ((weight_flag || mv.bcw_idx) && !(mv.bcw_idx && pu.ciip_flag)) {
                                j.weight_flag = 1;
                                if (mv.bcw_idx) {
                                    j.denom = 2; j.w1 = (int16_t)bcw_w_lut[mv.bcw_idx]; j.w0 = (int16_t)(8 - j.w1);
                                } else {
                                    j.denom = sl->log2_denom[c > 0];
                                    j.w0 = sl->weight[0][c][mv.ref_idx[0]]; j.w1 = sl->weight[1][c][mv.ref_idx[1]];
                                    j.o0 = sl->offset[0][c][mv.ref_idx[0]]; j.o1 = sl->offset[1][c][mv.ref_idx[1]];
                                }
                            }
                        } else if (sl->weighted_pred || sl->weighted_bipred) {
                            // derive_weight_uni (:129-146)
                            const int lx = mv.pred_flag - 1;
                            j.weight_flag = 1;
                            j.denom = sl->log2_denom[c > 0];
                            j.w0 = sl->weight[lx][c][mv.ref_idx[lx]];
                            j.o0 = sl->offset[lx][c][mv.ref_idx[lx]];
                        }
                        if (c == 0) jl[job] = j; else jc[2 * job + c - 1] = j;
                    }
                }
        }
}

// set_dmvr_info (vvc_inter.c:750-762) after the luma launch: one lane per luma job; a DMVR sub-block's 4x4 units get its MvField with
// the refined motion of its record
__global__ __launch_bounds__(256) void inter_dmvr_info_kernel(const vvc355_inter_frame *__restrict__ fp)
{
    const vvc355_inter_frame f = load_uniform(fp);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= f.n_jobs)
        return;
    const vvc355_bipred_job *j = (const vvc355_bipred_job *)f.jobs_luma + i;
    if (!j->dmvr)
        return;
    const vvc355_bipred_result *r = (const vvc355_bipred_result *)f.records + i;
    const MvFieldDev *src = (const MvFieldDev *)f.mvf;
    MvFieldDev *dst = (MvFieldDev *)f.dmvr_mvf;
    MvFieldDev m = src[(j->y >> 2) * f.mvf_stride + (j->x >> 2)];
    m.mv[0][0] = r->mv[0]; m.mv[0][1] = r->mv[1]; m.mv[1][0] = r->mv[2]; m.mv[1][1] = r->mv[3];
    for (int y = j->y; y < j->y + j->h; y += 4)
        for (int x = j->x; x < j->x + j->w; x += 4)
            dst[(y >> 2) * f.mvf_stride + (x >> 2)] = m;
}

} // namespace vvc355

extern "C" {

void vvc355_inter_frame_build(void *stream, const vvc355_inter_frame *frame_dev, const vvc355_inter_frame *frame_host)
{
    if (frame_host->n_pus <= 0) return;
    hipLaunchKernelGGL(vvc355::inter_build_kernel, dim3((frame_host->n_pus + 255) / 256), dim3(256), 0, (hipStream_t)stream, frame_dev);
    HIP_CHECK(hipGetLastError());
}

void vvc355_inter_frame_pass(void *stream, int bd, const vvc355_inter_frame *frame_dev, const vvc355_inter_frame *frame_host)
{
    if (frame_host->n_pus <= 0 || frame_host->n_jobs <= 0) return;
    vvc355_inter_frame_build(stream, frame_dev, frame_host);
    vvc355_bipred_batch(stream, bd, (const vvc355_bipred_job *)frame_host->jobs_luma, frame_host->n_jobs);
    if (frame_host->dmvr_mvf) {
        hipLaunchKernelGGL(vvc355::inter_dmvr_info_kernel, dim3((frame_host->n_jobs + 255) / 256), dim3(256), 0, (hipStream_t)stream, frame_dev);
        HIP_CHECK(hipGetLastError());
    }
    if (frame_host->chroma_format_idc)
        vvc355_bipred_chroma_batch(stream, bd, (const vvc355_bipred_job *)frame_host->jobs_chroma, 2 * frame_host->n_jobs);
}

} // extern "C"
