// Shared device/host helpers for the MI355X (gfx950) VVC pixel-kernel backend.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>

#define VVC355_PB 128   // MAX_PB_SIZE: implicit row stride of int16 MC intermediates (libavcodec/vvc/vvc_ctu.h:48)

// DSP slots return void and cannot report failure (vvcdsp.h:48-158): a HIP error is fatal and loud.
#define HIP_CHECK(expr)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "vvc_mi355: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), \
                    __FILE__, __LINE__);                                                        \
            abort();                                                                            \
        }                                                                                       \
    } while (0)

namespace vvc355 {

template <int BD> struct Px       { using type = uint16_t; };
template <>       struct Px<8>    { using type = uint8_t;  };

__device__ __forceinline__ int clip3(int v, int lo, int hi) { return min(max(v, lo), hi); }
template <int BD> __device__ __forceinline__ int clip_px(int v) { return clip3(v, 0, (1 << BD) - 1); }
__device__ __forceinline__ int clip_intp2(int v, int p) { return clip3(v, -(1 << p), (1 << p) - 1); }
__device__ __forceinline__ int sign_of(int v) { return (v > 0) - (v < 0); }
__device__ __forceinline__ int ilog2(unsigned v) { return 31 - __clz(v | 1); }

template <int BD> __device__ __forceinline__ int ld_px(const uint8_t *p, ptrdiff_t i)
{
    return reinterpret_cast<const typename Px<BD>::type *>(p)[i];
}
template <int BD> __device__ __forceinline__ void st_px(uint8_t *p, ptrdiff_t i, int v)
{
    reinterpret_cast<typename Px<BD>::type *>(p)[i] = (typename Px<BD>::type)v;
}

// dispatch a kernel template on the runtime bit depth
#define VVC355_BD_DISPATCH(bd, CALL)                         \
    do {                                                     \
        switch (bd) {                                        \
        case 8:  { constexpr int BD = 8;  CALL; } break;     \
        case 10: { constexpr int BD = 10; CALL; } break;     \
        case 12: { constexpr int BD = 12; CALL; } break;     \
        default:                                             \
            fprintf(stderr, "vvc_mi355: unsupported bit depth %d\n", (int)(bd)); abort(); \
        }                                                    \
    } while (0)

} // namespace vvc355
