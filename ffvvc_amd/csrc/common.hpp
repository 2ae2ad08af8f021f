// Shared device/host helpers for the MI355X (gfx950) VVC pixel-kernel backend.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>

#define VVC355_PB 128   // MAX_PB_SIZE: implicit row stride of int16 MC intermediates (libavcodec/vvc/vvc_ctu.h:48)

// DSP slots return void and cannot report failure (vvcdsp.h:48-158): by default a HIP error is fatal and loud.  A host that drives the
// batched / frame entries itself can ask for errors to be recorded instead (vvc355_set_error_policy(1)): the first failure is kept
// (vvc355_last_error / _string, text written before the code is published), the entry goes on — launch and copy errors are sticky,
// later calls fail too; a thread whose stream / staging arena could not be created aborts at its first slot call — and the host
// checks after the stage or at its stream synchronisation.
namespace vvc355 { void hip_fail(const char *expr, int err, const char *what, const char *file, int line); }
#define HIP_CHECK(expr)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            ::vvc355::hip_fail(#expr, (int)e_, hipGetErrorString(e_), __FILE__, __LINE__);      \
    } while (0)

namespace vvc355 {

template <int BD> struct Px       { using type = uint16_t; };
template <>       struct Px<8>    { using type = uint8_t;  };

__device__ __forceinline__ int clip3(int v, int lo, int hi) { return min(max(v, lo), hi); }
template <int BD> __device__ __forceinline__ int clip_px(int v) { return clip3(v, 0, (1 << BD) - 1); }
__device__ __forceinline__ int clip_intp2(int v, int p) { return clip3(v, -(1 << p), (1 << p) - 1); }
__device__ __forceinline__ int sign_of(int v) { return (v > 0) - (v < 0); }
__device__ __forceinline__ int ilog2(unsigned v) { return 31 - __clz(v | 1); }

// Job descriptors carry device addresses as integers, so the compiler sees generic ("flat") pointers and would emit
// flat_load / flat_store, which arbitrate through both the LDS and the vector-memory path.  Every HBM access goes through
// these helpers instead: an explicit global address space gives global_load / global_store.
#define VVC355_GLOBAL __attribute__((address_space(1)))
template <int N> struct RawBits;
template <> struct RawBits<1> { typedef uint8_t type; };
template <> struct RawBits<2> { typedef uint16_t type; };
template <> struct RawBits<4> { typedef uint32_t type; };
template <> struct RawBits<8> { typedef uint32_t type __attribute__((ext_vector_type(2))); };
template <> struct RawBits<16> { typedef uint32_t type __attribute__((ext_vector_type(4))); };
template <typename T> __device__ __forceinline__ T gld(const void *p)
{
    typedef typename RawBits<sizeof(T)>::type raw_t;
    const raw_t r = *(const VVC355_GLOBAL raw_t *)p;
    T v;
    __builtin_memcpy(&v, &r, sizeof(T));
    return v;
}
template <typename T> __device__ __forceinline__ void gst(void *p, T v)
{
    typedef typename RawBits<sizeof(T)>::type raw_t;
    raw_t r;
    __builtin_memcpy(&r, &v, sizeof(T));
    *(VVC355_GLOBAL raw_t *)p = r;
}

// A job descriptor at a wave-uniform address, fetched with scalar loads (s_load_dwordx*): the constant address space tells the
// compiler that the scalar cache may be used.  A plain struct copy is split into per-field loads, and byte / short fields make
// those vector loads: a full vector-memory round trip before the kernel can even compute its first pixel address.
#define VVC355_CONST __attribute__((address_space(4)))
template <typename T> __device__ __forceinline__ T load_uniform(const T *p)
{
    static_assert(sizeof(T) % 4 == 0, "job descriptors are whole dwords");
    constexpr int N = sizeof(T) / 4;
    uint32_t w[N];
    const VVC355_CONST uint32_t *q = (const VVC355_CONST uint32_t *)p;
#pragma unroll
    for (int i = 0; i < N; i++) w[i] = q[i];
    T r;
    __builtin_memcpy(&r, w, sizeof(T));
    return r;
}

// Byte offset of row y of a plane: a full-rate 24-bit multiply instead of the quarter-rate 64-bit multiply-add that
// (ptrdiff_t)y * stride compiles to.  Valid for |y|, |stride| < 2^23 and planes below 2 GiB (y * stride < 2^31): an 8K 16-bit
// plane is 66 MB.
__device__ __forceinline__ ptrdiff_t row_off(int y, int stride) { return (ptrdiff_t)__mul24(y, stride); }

// pixel load / store on HBM planes (never on LDS)
template <int BD> __device__ __forceinline__ int ld_px(const uint8_t *p, ptrdiff_t i)
{
    return ((const VVC355_GLOBAL typename Px<BD>::type *)p)[i];
}
template <int BD> __device__ __forceinline__ void st_px(uint8_t *p, ptrdiff_t i, int v)
{
    ((VVC355_GLOBAL typename Px<BD>::type *)p)[i] = (typename Px<BD>::type)v;
}

// lmcs.filter on one sample (vvc_filter_template.c:25): v -> lut[v]; lut = 0 (wave-uniform) leaves the sample as it is.  The inter
// prediction kernels store luma through it when the job carries the picture's forward map (vvc_inter.c:888-891, :573-574).
template <int BD> __device__ __forceinline__ int lmcs_fwd(const uint8_t *lut, int v)
{
    return lut ? (int)gld<typename Px<BD>::type>((const typename Px<BD>::type *)lut + v) : v;
}

// Address = wave-uniform base + unsigned 32-bit per-lane byte offset: the form the hardware takes directly (SGPR pair + VGPR
// offset), so no 64-bit add per access.  Valid for planes below 4 GiB and offsets that are not negative.
template <typename T> __device__ __forceinline__ T gld_at(const uint8_t *base, uint32_t off) { return gld<T>(base + (size_t)off); }
template <typename T> __device__ __forceinline__ void gst_at(uint8_t *base, uint32_t off, T v) { gst<T>(base + (size_t)off, v); }

// Workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  xcd_chunked() renumbers workgroup b of n so that every
// XCD works through one contiguous eighth of the job list: neighbouring jobs — neighbouring blocks of the picture, whose
// reference windows overlap — then meet in the same L2 instead of being fetched from HBM once per XCD.
__device__ __forceinline__ int xcd_chunked(int b, int n)
{
    const int xcd = b & 7, idx = b >> 3, q = n >> 3, r = n & 7;
    return xcd * q + min(xcd, r) + idx;
}

// The same in groups: XCD x takes G consecutive workgroups out of every 8 * G, so the XCDs stay within 8 * G workgroups of each
// other in the picture (DRAM page locality) while G neighbours share an L2.  The tail that does not fill 8 * G keeps its number.
__device__ __forceinline__ int xcd_grouped(int b, int n, int G)
{
    const int span = 8 * G, sg = b / span, r = b - sg * span;
    if ((sg + 1) * span > n)
        return b;
    return sg * span + (r & 7) * G + (r >> 3);
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
