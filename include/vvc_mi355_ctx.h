/*
 * vvc_mi355_ctx.h — the decoder state the four context-taking slots read, as a plain-C mirror.
 *
 * intra.intra_pred, intra.intra_cclm_pred, intra.lmcs_scale_chroma (libavcodec/vvc/vvcdsp.h:98-100) take the decoder's
 * VVCLocalContext, sao.edge_restore[2] (:143) its SAOParams.  ffvvc_amd/host/dsp_ctx_shim.c reads every member through accessor
 * macros; this header gives the standalone build (the one compiled and tested here) the same struct names with exactly the
 * members those slots read.  The accessors' VVC355_IN_TREE branch maps them onto the decoder's own headers (INTEGRATION.md
 * section 2); that branch needs FFmpeg's configure output and is not compiled in this repository.
 * Field names follow the reference (vvc_ctu.h:334-460, vvc_ps.h:193-202, vvcdec.h:122-187); nothing else of those structs exists.
 */
#ifndef VVC_MI355_CTX_H
#define VVC_MI355_CTX_H

#include <stddef.h>
#include <stdint.h>

#include "vvc_mi355.h"
#include "vvc_mi355_dsp.h"

#ifdef __cplusplus
extern "C" {
#endif

#define VVC355_MAX_PARTS_IN_CTU 1024                /* (MAX_CTU_SIZE >> MIN_CU_LOG2)^2, vvc_ctu.h:38 */

typedef struct ReconstructedArea { int x, y, w, h; } ReconstructedArea;               /* vvc_ctu.h:334-339 */

typedef struct SAOParams {                          /* vvc_ctu.h:440-452, same member order */
    int      offset_abs[3][4];
    int      offset_sign[3][4];
    uint8_t  band_position[3];
    int      eo_class[3];
    int16_t  offset_val[3][5];
    uint8_t  type_idx[3];
} SAOParams;

typedef struct CodingUnit {                         /* vvc_ctu.h:226-280: the members the slots read */
    int      x0, y0, cb_width, cb_height;
    int      intra_pred_mode_y, intra_pred_mode_c;
    uint8_t  intra_luma_ref_idx, isp_split_type, mip_chroma_direct_flag;
    uint8_t  bdpcm_flag[3];
} CodingUnit;

typedef struct VVCFrameContext {                    /* vvcdec.h:122-187 + the parameter sets it points to, flattened */
    uint8_t *data[3];                               /* fc->frame->data / linesize */
    int      linesize[3];
    int      width, height;                         /* pps->width / height (luma samples) */
    int      bit_depth;
    uint8_t  hshift[3], vshift[3];                  /* sps->hshift / vshift */
    uint8_t  ctb_log2_size_y, min_cb_log2_size_y;
    int      min_cb_width;                          /* pps->min_cb_width */
    uint8_t  sps_entropy_coding_sync_enabled_flag, sps_chroma_vertical_collocated_flag;
    const uint8_t *imf, *imm, *imtf;                /* fc->tab.imf / imm / imtf: per minimum coding block */
    struct {                                        /* fc->ps.lmcs (VVCLMCS, vvc_ps.h:193-202) */
        uint8_t  min_bin_idx, max_bin_idx;
        uint16_t pivot[17], chroma_scale_coeff[16];
    } lmcs;
} VVCFrameContext;

typedef struct VVCLocalContext {                    /* vvc_ctu.h:354-436: the members the slots read */
    VVCFrameContext *fc;
    const CodingUnit *cu;
    ReconstructedArea ras[2][VVC355_MAX_PARTS_IN_CTU];
    int      num_ras[2];
    struct { int cand_up_left; } na;                /* NeighbourAvailable, set by ff_vvc_set_neighbour_available (vvc_ctu.c:2497) */
    uint8_t  ctb_left_flag, ctb_up_flag;
    int      end_of_tiles_x;
    struct { int x_vpdu, y_vpdu, chroma_scale; } lmcs;      /* per-CTU cache of lmcs_derive_chroma_scale (vvc_intra_template.c:390) */
} VVCLocalContext;

/* the flattening itself, exposed for the CPU tests: fills the job a slot call turns into (no GPU work) */
void vvc355_ctx_flatten_intra_pred(const VVCLocalContext *lc, int x0, int y0, int width, int height, int c_idx, vvc355_intra_job *job);
void vvc355_ctx_flatten_cclm(const VVCLocalContext *lc, int x0, int y0, int width, int height, vvc355_cclm_job *job);
void vvc355_ctx_flatten_lmcs_scale(const VVCLocalContext *lc, int x0_cu, int y0_cu, vvc355_lmcs_scale_job *job);
/* Reference-sample availability (the question ff_vvc_get_top_available / _left_available answer, vvc_intra.c:591-648).  The shim's own
 * derivation works from a coverage mask of the neighbouring line (dsp_ctx_shim.c); an in-tree build installs the decoder's own two
 * functions instead: vvc355_ctx_set_availability(ff_vvc_get_top_available, ff_vvc_get_left_available).  NULL = the shim's own. */
typedef int (*vvc355_avail_fn)(const VVCLocalContext *lc, int x, int y, int target_size, int c_idx);
void vvc355_ctx_set_availability(vvc355_avail_fn top, vvc355_avail_fn left);
int  vvc355_ctx_top_available(const VVCLocalContext *lc, int x, int y, int target_size, int c_idx);
int  vvc355_ctx_left_available(const VVCLocalContext *lc, int x, int y, int target_size, int c_idx);

/* installs the four context-taking slots of the table for `bit_depth` (8, 10, 12); call after ff_vvc_dsp_init_mi355 */
void ff_vvc_dsp_init_mi355_ctx(VVC355DSPContext *c, int bit_depth);

#ifdef __cplusplus
}
#endif
#endif
