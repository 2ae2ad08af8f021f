/*
 * vvc_mi355.h — C ABI of libvvc_mi355.so: hand-written HIP (gfx950 / MI355X) kernels behind the
 * VVCDSPContext function-pointer surface of the ffvvc decoder.
 *
 * Two ways in:
 *
 *  (1) Synchronous per-slot entries, `vvc355_<slot>(int bd, ...)`: HOST pointers, same argument order
 *      and meaning as the reference slot they replace (cited per entry, paths relative to the
 *      reference tree), with the bit depth the reference binds at ff_vvc_dsp_init() time
 *      (libavcodec/vvc/vvcdsp.c:228) passed first, and table indices ([luma/chroma][frac][frac], [h/v])
 *      passed as leading ints.  They stage the touched rectangle to the GPU, run the kernel and copy
 *      the result back: drop-in and bit-exact, meant for the function-pointer table
 *      (ffvvc_amd/host/dsp_init_mi355.c) and for checkasm-style parity — not for speed.
 *
 *  (2) Batched entries, `vvc355_<stage>_batch(stream, bd, jobs_dev, n_jobs)`: DEVICE-resident planes
 *      and a device array of POD job descriptors — one launch per stage for many CTUs.  This is the
 *      performance path (SURVEY §7 "batched API").
 *
 * Every entry returns void like the slot it replaces (no error channel, vvcdsp.h:48-158); a HIP
 * failure or an argument outside the slot's domain prints a message and abort()s.
 */
#ifndef VVC_MI355_H
#define VVC_MI355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ runtime helpers (runtime_api.cpp) */
/* Device count / selection and plain device memory, for C hosts that keep frames resident in HBM. */
int   vvc355_device_count(void);
void  vvc355_set_device(int ordinal);
void *vvc355_malloc(size_t bytes);
void  vvc355_free(void *dev);
void  vvc355_upload(void *dev, const void *host, size_t bytes);
void  vvc355_download(void *host, const void *dev, size_t bytes);
void *vvc355_stream_create(void);
void  vvc355_stream_destroy(void *stream);
void  vvc355_stream_sync(void *stream);          /* NULL = the default stream */
const char *vvc355_version(void);

/* ------------------------------------------------------------------ ALF (alf.hip) */

/*
 * One ALF rectangle (normally one CTB of one component).  Addresses are DEVICE addresses of the
 * rectangle's top-left sample in the destination plane and in the pre-ALF source plane.
 * ext_* = number of samples that may be READ beyond the rectangle on that side of the source
 * (>= 3 luma / 2 chroma means "plain read"; fewer means replicate the last readable sample, which is
 * what the reference's alf_prepare_buffer does at picture / slice / tile edges,
 * libavcodec/vvc/vvc_filter.c:1105-1137).
 */
typedef struct vvc355_alf_job {
    uint64_t dst;
    uint64_t src;            /* luma/chroma: source plane; CC-ALF: the co-located LUMA plane */
    uint64_t coeff;          /* luma slot mode: int16[n4x4][12]; luma fused: int16 coeff_set[][12];
                                chroma: int16[6]; CC: int16[7]; classify-only: int class_idx[n4x4] (output) */
    uint64_t clip;           /* luma slot mode: int16[n4x4][12]; luma fused: uint8 clip_idx_set[25][12];
                                chroma: int16[6]; classify-only: int transpose_idx[n4x4] (output) */
    uint64_t class_to_filt;  /* luma fused: uint8[25] */
    int32_t  dst_stride;     /* bytes */
    int32_t  src_stride;     /* bytes */
    int16_t  w, h;           /* samples; luma/chroma: multiples of 4, <= 128 */
    int16_t  vb_pos;         /* virtual boundary row relative to the rectangle (vvc_filter.c:1305-1314) */
    int8_t   ext_l, ext_r, ext_t, ext_b;
    int8_t   hs, vs;         /* CC-ALF chroma subsampling shifts */
    int8_t   pad_[4];
} vvc355_alf_job;

/* alf.filter[LUMA] over many rectangles; fused != 0 additionally runs alf.classify and
 * alf.recon_coeff_and_clip in the same kernel (what vvc_filter.c:1139-1186 chains per CTB). */
void vvc355_alf_luma_batch(void *stream, int bd, int fused, const vvc355_alf_job *jobs_dev, int n_jobs);
void vvc355_alf_chroma_batch(void *stream, int bd, const vvc355_alf_job *jobs_dev, int n_jobs);
void vvc355_alf_cc_batch(void *stream, int bd, const vvc355_alf_job *jobs_dev, int n_jobs);

/* VVCALFDSPContext.filter[LUMA] — libavcodec/vvc/vvcdsp.h:149, vvc_filter_template.c:43 */
void vvc355_alf_filter_luma(int bd, uint8_t *dst, ptrdiff_t dst_stride, const uint8_t *src, ptrdiff_t src_stride,
    int width, int height, const int16_t *filter, const int16_t *clip, int vb_pos);
/* VVCALFDSPContext.filter[CHROMA] — vvcdsp.h:149, vvc_filter_template.c:137 */
void vvc355_alf_filter_chroma(int bd, uint8_t *dst, ptrdiff_t dst_stride, const uint8_t *src, ptrdiff_t src_stride,
    int width, int height, const int16_t *filter, const int16_t *clip, int vb_pos);
/* VVCALFDSPContext.filter_cc — vvcdsp.h:151, vvc_filter_template.c:223 */
void vvc355_alf_filter_cc(int bd, uint8_t *dst, ptrdiff_t dst_stride, const uint8_t *luma, ptrdiff_t luma_stride,
    int width, int height, int hs, int vs, const int16_t *filter, int vb_pos);
/* VVCALFDSPContext.classify — vvcdsp.h:154, vvc_filter_template.c:299 (gradient_tmp is not touched) */
void vvc355_alf_classify(int bd, int *class_idx, int *transpose_idx, const uint8_t *src, ptrdiff_t src_stride,
    int width, int height, int vb_pos, int *gradient_tmp);
/* VVCALFDSPContext.recon_coeff_and_clip — vvcdsp.h:156, vvc_filter_template.c:383 */
void vvc355_alf_recon_coeff_and_clip(int bd, int16_t *coeff, int16_t *clip, const int *class_idx, const int *transpose_idx,
    int size, const int16_t *coeff_set, const uint8_t *clip_idx_set, const uint8_t *class_to_filt);

#ifdef __cplusplus
}
#endif
#endif /* VVC_MI355_H */
