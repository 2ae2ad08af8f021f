/*
 * oracle/ — CPU restatement of the VVC pixel-kernel (DSP) path of the ffvvc decoder.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call it, and only as the
 * checker / CPU baseline — never as a fallback for the HIP path.
 *
 * PARITY UNPINNED: the reference's own tests hold no golden vectors for this path (checkasm
 * compares the C template with an override at run time; the FATE framecrc lists need bitstreams
 * that are not in the container), and the reference's C path cannot be compiled here without its
 * configure-generated headers (config.h, libavutil/avconfig.h).  Every function below therefore
 * restates the reference algorithm from a reading of the cited file:line, and is cross-checked
 * only by independent properties (tests/test_oracle_*.py).
 *
 * Conventions: `bd` = bit depth (8, 10 or 12); pixels are uint8_t when bd == 8 and uint16_t
 * otherwise; pixel strides are in BYTES exactly as on the reference's function-pointer surface
 * (libavcodec/vvc/vvcdsp.h:48-158); int16 MC intermediates have an implicit row stride of 128.
 */
#ifndef ORC_COMMON_H
#define ORC_COMMON_H

#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_PB 128            /* MAX_PB_SIZE, libavcodec/vvc/vvc_ctu.h:48 */
#define ORC_INLINE static inline __attribute__((always_inline))
#define ORC_API __attribute__((visibility("default")))

ORC_INLINE int orc_min(int a, int b) { return a < b ? a : b; }
ORC_INLINE int orc_max(int a, int b) { return a > b ? a : b; }
ORC_INLINE int orc_abs(int a) { return a < 0 ? -a : a; }
ORC_INLINE int orc_clip3(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
ORC_INLINE int orc_sign(int v) { return (v > 0) - (v < 0); }
/* clamp to [0, 2^bd - 1] */
ORC_INLINE int orc_clip_px(int v, int bd) { return orc_clip3(v, 0, (1 << bd) - 1); }
/* clamp to [-2^p, 2^p - 1] (libavutil/common.h av_clip_intp2) */
ORC_INLINE int orc_clip_intp2(int v, int p) { return orc_clip3(v, -(1 << p), (1 << p) - 1); }
/* clamp to [0, 2^p - 1] */
ORC_INLINE int orc_clip_uintp2(int v, int p) { return orc_clip3(v, 0, (1 << p) - 1); }
/* floor(log2(v)), v > 0 */
ORC_INLINE int orc_log2(unsigned v) { return 31 - __builtin_clz(v | 1); }

/* pixel load/store on a byte pointer; `wide` = (bd > 8) */
ORC_INLINE int orc_ld(const uint8_t *p, ptrdiff_t i, int wide)
{
    return wide ? ((const uint16_t *)p)[i] : p[i];
}
ORC_INLINE void orc_st(uint8_t *p, ptrdiff_t i, int v, int wide)
{
    if (wide) ((uint16_t *)p)[i] = (uint16_t)v; else p[i] = (uint8_t)v;
}

/* instantiate an always-inline body for the three bit depths so the compiler folds `bd` */
#define ORC_BD_SWITCH(bd, CALL8, CALL10, CALL12) \
    do { switch (bd) { case 8: CALL8; break; case 10: CALL10; break; case 12: CALL12; break; default: abort(); } } while (0)

#endif
