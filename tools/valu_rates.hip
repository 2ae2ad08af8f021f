// Issue rate of the integer VALU instructions the kernels lean on, measured on the device (gfx950): a grid that fills every SIMD
// with 8 waves runs N x 16 independent instances of one instruction per lane; cycles per wave-instruction per SIMD
// = elapsed x clock x SIMDs / (waves x N x 16).  Build: hipcc --offload-arch=gfx950 -O2 -o valu_rates tools/valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define KERNEL(name, ASM)                                                                          \
    __global__ __launch_bounds__(256) void name(uint32_t *out, int n)                               \
    {                                                                                              \
        uint32_t a[16], b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;                     \
        for (int i = 0; i < 16; i++) a[i] = threadIdx.x + i;                                        \
        for (int it = 0; it < n; it++) {                                                           \
            _Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c) : "vcc", "s2", "s3"); \
        }                                                                                          \
        uint32_t s = 0;                                                                            \
        for (int i = 0; i < 16; i++) s ^= a[i];                                                     \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                                   \
    }

KERNEL(k_add_u32,      "v_add_u32 %0, %0, %1")
KERNEL(k_mad_i32_i24,  "v_mad_i32_i24 %0, %1, %2, %0")
KERNEL(k_med3_i32,     "v_med3_i32 %0, %0, %1, %2")
KERNEL(k_pk_add_u16,   "v_pk_add_u16 %0, %0, %1")
KERNEL(k_pk_sub_i16,   "v_pk_sub_i16 %0, %0, %1")
KERNEL(k_pk_max_i16,   "v_pk_max_i16 %0, %0, %1")
KERNEL(k_pk_mad_i16,   "v_pk_mad_i16 %0, %1, %2, %0")
KERNEL(k_dot2_i32_i16, "v_dot2_i32_i16 %0, %1, %2, %0")
KERNEL(k_dot2c_i32_i16,"v_dot2c_i32_i16 %0, %1, %2")
KERNEL(k_mad_i32_i16,  "v_mad_i32_i16 %0, %1, %2, %0")
KERNEL(k_sad_u16,      "v_sad_u16 %0, %1, %2, %0")
KERNEL(k_sad_u32,      "v_sad_u32 %0, %1, %2, %0")
KERNEL(k_perm_b32,     "v_perm_b32 %0, %0, %1, %2")
KERNEL(k_alignbit,     "v_alignbit_b32 %0, %0, %1, 16")
KERNEL(k_mul_lo_u32,   "v_mul_lo_u32 %0, %0, %1")
KERNEL(k_mul_i32_i24,  "v_mul_i32_i24 %0, %0, %1")
KERNEL(k_bfe_i32,      "v_bfe_i32 %0, %0, 3, 9")
KERNEL(k_lshl_add_u32, "v_lshl_add_u32 %0, %0, 1, %1")
KERNEL(k_add3_u32,     "v_add3_u32 %0, %0, %1, %2")
KERNEL(k_cndmask,      "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL(k_and_b32,      "v_and_b32 %0, %0, %1")
KERNEL(k_xor_b32,      "v_xor_b32 %0, %0, %1")
KERNEL(k_lshlrev_b32,  "v_lshlrev_b32 %0, 3, %0")
KERNEL(k_sub_u32,      "v_sub_u32 %0, %0, %1")
KERNEL(k_max_i32,      "v_max_i32 %0, %0, %1")
KERNEL(k_min_u32,      "v_min_u32 %0, %0, %1")
KERNEL(k_add_u32_e64,  "v_add_u32_e64 %0, %0, %1")
KERNEL(k_bfi_b32,      "v_bfi_b32 %0, %0, %1, %2")
KERNEL(k_and_or_b32,   "v_and_or_b32 %0, %0, %1, %2")
KERNEL(k_cndmask_e64,  "v_cndmask_b32_e64 %0, %0, %1, s[2:3]")
KERNEL(k_mov_b32,      "v_mov_b32 %0, %1")
KERNEL(k_max3_i32,     "v_max3_i32 %0, %0, %1, %2")
KERNEL(k_cmp_cnd_vcc,  "v_cmp_lt_i32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc")
KERNEL(k_cmp_cnd_sgpr, "v_cmp_lt_i32_e64 s[2:3], %0, %1\n\tv_cndmask_b32_e64 %0, %0, %2, s[2:3]")
KERNEL(k_cmp_vcc,      "v_cmp_lt_i32 vcc, %0, %1")
KERNEL(k_cmp_sgpr,     "v_cmp_lt_i32_e64 s[2:3], %0, %1")
KERNEL(k_sub_sdwa,     "v_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1")

int main()
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount, simds = cus * 4;
    const double ghz = p.clockRate * 1e-6;
    const int wgs = cus * 8, n = 4096;                       // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    uint32_t *out;
    CHECK(hipMalloc(&out, (size_t)wgs * 256 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_ghz\": %.3f, \"cycles_per_wave_instruction_per_simd\": {", p.gcnArchName, cus, ghz);
    struct { const char *name; void (*k)(uint32_t *, int); } ks[] = {
        { "v_add_u32", k_add_u32 }, { "v_mad_i32_i24", k_mad_i32_i24 }, { "v_med3_i32", k_med3_i32 }, { "v_pk_add_u16", k_pk_add_u16 },
        { "v_pk_sub_i16", k_pk_sub_i16 }, { "v_pk_max_i16", k_pk_max_i16 }, { "v_pk_mad_i16", k_pk_mad_i16 },
        { "v_dot2_i32_i16", k_dot2_i32_i16 }, { "v_dot2c_i32_i16", k_dot2c_i32_i16 }, { "v_mad_i32_i16", k_mad_i32_i16 },
        { "v_sad_u16", k_sad_u16 }, { "v_sad_u32", k_sad_u32 }, { "v_perm_b32", k_perm_b32 }, { "v_alignbit_b32", k_alignbit },
        { "v_mul_lo_u32", k_mul_lo_u32 }, { "v_mul_i32_i24", k_mul_i32_i24 }, { "v_bfe_i32", k_bfe_i32 }, { "v_lshl_add_u32", k_lshl_add_u32 },
        { "v_add3_u32", k_add3_u32 }, { "v_cndmask_b32", k_cndmask }, { "v_sub_u32_sdwa", k_sub_sdwa },
        { "v_and_b32", k_and_b32 }, { "v_xor_b32", k_xor_b32 }, { "v_lshlrev_b32", k_lshlrev_b32 }, { "v_sub_u32", k_sub_u32 },
        { "v_max_i32", k_max_i32 }, { "v_min_u32", k_min_u32 }, { "v_add_u32_e64", k_add_u32_e64 }, { "v_bfi_b32", k_bfi_b32 },
        { "v_cmp_lt_i32 vcc + v_cndmask vcc (pair)", k_cmp_cnd_vcc }, { "v_cmp_lt_i32_e64 sgpr + v_cndmask_e64 sgpr (pair)", k_cmp_cnd_sgpr },
        { "v_cmp_lt_i32 vcc", k_cmp_vcc }, { "v_cmp_lt_i32_e64 sgpr", k_cmp_sgpr }, { "v_and_or_b32", k_and_or_b32 }, { "v_cndmask_b32_e64(sgpr mask)", k_cndmask_e64 }, { "v_mov_b32", k_mov_b32 }, { "v_max3_i32", k_max3_i32 },
    };
    bool first = true;
    for (auto &k : ks) {
        hipLaunchKernelGGL(k.k, dim3(wgs), dim3(256), 0, 0, out, 64);              // warm-up
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k.k, dim3(wgs), dim3(256), 0, 0, out, n);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double insts_per_simd = (double)wgs * 4 * n * 16 / simds;
        printf("%s\"%s\": %.2f", first ? "" : ", ", k.name, ms * 1e-3 * ghz * 1e9 / insts_per_simd);
        first = false;
    }
    printf("}}\n");
    return 0;
}
