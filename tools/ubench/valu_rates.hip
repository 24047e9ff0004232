// Micro-benchmark: issue rate of the VALU instructions the render kernel is made of (gfx950).
// Reports cycles per wave64 instruction per SIMD at full occupancy (8 independent chains per wave).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP 64
#define ITERS 4096

#define KERNEL(NAME, DECL, BODY)                                                   \
    __global__ __launch_bounds__(256) void NAME(unsigned *out, unsigned seed) {  \
        DECL;                                                                      \
        for (int it = 0; it < ITERS; it++) {                                       \
            _Pragma("unroll") for (int r = 0; r < REP / 8; r++) { BODY }          \
        }                                                                          \
        unsigned acc = 0;                                                          \
        _Pragma("unroll") for (int k = 0; k < 8; k++) acc ^= __float_as_uint(*(float *)&x[k]); \
        if (acc == 0x12345678u) out[threadIdx.x] = acc;                           \
    }

#define ASM8F(INS) _Pragma("unroll") for (int k = 0; k < 8; k++) asm volatile(INS : "+v"(x[k]) : "v"(y), "v"(z));
#define ASM8U(INS) _Pragma("unroll") for (int k = 0; k < 8; k++) asm volatile(INS : "+v"(x[k]) : "v"(y), "v"(z));

#define FDECL float x[8]; float y = __uint_as_float(seed | 0x3f800000u) , z = 0.5f; for (int k = 0; k < 8; k++) x[k] = threadIdx.x * 0.001f + k
#define UDECL unsigned x[8]; unsigned y = seed | 1u, z = seed * 77u + 3u; for (int k = 0; k < 8; k++) x[k] = threadIdx.x * 2654435761u + k

KERNEL(k_fma, FDECL, ASM8F("v_fma_f32 %0, %0, %1, %2"))
KERNEL(k_add, FDECL, ASM8F("v_add_f32 %0, %0, %1"))
KERNEL(k_mul, FDECL, ASM8F("v_mul_f32 %0, %0, %1"))
KERNEL(k_sqrt, FDECL, ASM8F("v_sqrt_f32 %0, %0"))
KERNEL(k_rcp, FDECL, ASM8F("v_rcp_f32 %0, %0"))
KERNEL(k_rsq, FDECL, ASM8F("v_rsq_f32 %0, %0"))
KERNEL(k_sin, FDECL, ASM8F("v_sin_f32 %0, %0"))
KERNEL(k_cndmask, FDECL, ASM8F("v_cndmask_b32 %0, %0, %1, vcc"))
KERNEL(k_max, FDECL, ASM8F("v_max_f32 %0, %0, %1"))
KERNEL(k_mul_lo, UDECL, ASM8U("v_mul_lo_u32 %0, %0, %1"))
KERNEL(k_mul_hi, UDECL, ASM8U("v_mul_hi_u32 %0, %0, %1"))
KERNEL(k_mul_u24, UDECL, ASM8U("v_mul_u32_u24 %0, %0, %1"))
KERNEL(k_mad_u24, UDECL, ASM8U("v_mad_u32_u24 %0, %0, %1, %2"))
KERNEL(k_xor, UDECL, ASM8U("v_xor_b32 %0, %0, %1"))
KERNEL(k_lshr, UDECL, ASM8U("v_lshrrev_b32 %0, 3, %0"))
KERNEL(k_add_u32, UDECL, ASM8U("v_add_u32 %0, %0, %1"))
KERNEL(k_add3, UDECL, ASM8U("v_add3_u32 %0, %0, %1, %2"))
KERNEL(k_xad, UDECL, ASM8U("v_xad_u32 %0, %0, %1, %2"))
KERNEL(k_lshl_add, UDECL, ASM8U("v_lshl_add_u32 %0, %0, 3, %1"))
KERNEL(k_alignbit, UDECL, ASM8U("v_alignbit_b32 %0, %0, %0, 13"))
KERNEL(k_bfe, UDECL, ASM8U("v_bfe_u32 %0, %0, 8, 24"))
KERNEL(k_cvt_f32_u32, UDECL, ASM8U("v_cvt_f32_u32 %0, %0"))
KERNEL(k_perm, UDECL, ASM8U("v_perm_b32 %0, %0, %1, %2"))
KERNEL(k_and_or, UDECL, ASM8U("v_and_or_b32 %0, %0, %1, %2"))

// compare + select pairs (2 instructions per "REP" unit: reported cycles are per PAIR)
KERNEL(k_cmp_cnd_vcc, FDECL, _Pragma("unroll") for (int k = 0; k < 8; k++) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(x[k]) : "v"(y), "v"(z) : "vcc");)
KERNEL(k_cmp_cnd_sgpr, FDECL, _Pragma("unroll") for (int k = 0; k < 8; k++) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1\n\tv_cndmask_b32 %0, %0, %2, s[20:21]" : "+v"(x[k]) : "v"(y), "v"(z) : "s20", "s21");)
KERNEL(k_cmp_only, FDECL, _Pragma("unroll") for (int k = 0; k < 8; k++) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : "+v"(x[k]) : "v"(y), "v"(z) : "vcc");)
KERNEL(k_cmp_only_sgpr, FDECL, _Pragma("unroll") for (int k = 0; k < 8; k++) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1" : "+v"(x[k]) : "v"(y), "v"(z) : "s20", "s21");)
KERNEL(k_cnd_after_cmp, FDECL, asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(x[0]), "v"(y) : "vcc"); _Pragma("unroll") for (int k = 0; k < 8; k++) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[k]) : "v"(y), "v"(z));)
// one compare feeding several selects (cycles reported per GROUP of 1 cmp + n selects)
KERNEL(k_cmp_2cnd, FDECL, _Pragma("unroll") for (int k = 0; k < 8; k += 2) asm volatile("v_cmp_lt_f32 vcc, %0, %2\n\tv_cndmask_b32 %0, %0, %3, vcc\n\tv_cndmask_b32 %1, %1, %3, vcc" : "+v"(x[k]), "+v"(x[k + 1]) : "v"(y), "v"(z) : "vcc");)
KERNEL(k_cmp_4cnd, FDECL, _Pragma("unroll") for (int k = 0; k < 8; k += 4) asm volatile("v_cmp_lt_f32 vcc, %0, %4\n\tv_cndmask_b32 %0, %0, %5, vcc\n\tv_cndmask_b32 %1, %1, %5, vcc\n\tv_cndmask_b32 %2, %2, %5, vcc\n\tv_cndmask_b32 %3, %3, %5, vcc" : "+v"(x[k]), "+v"(x[k + 1]), "+v"(x[k + 2]), "+v"(x[k + 3]) : "v"(y), "v"(z) : "vcc");)
KERNEL(k_cmp_4cnd_sgpr, FDECL, _Pragma("unroll") for (int k = 0; k < 8; k += 4) asm volatile("v_cmp_lt_f32 s[20:21], %0, %4\n\tv_cndmask_b32 %0, %0, %5, s[20:21]\n\tv_cndmask_b32 %1, %1, %5, s[20:21]\n\tv_cndmask_b32 %2, %2, %5, s[20:21]\n\tv_cndmask_b32 %3, %3, %5, s[20:21]" : "+v"(x[k]), "+v"(x[k + 1]), "+v"(x[k + 2]), "+v"(x[k + 3]) : "v"(y), "v"(z) : "s20", "s21");)
KERNEL(k_bfi, FDECL, ASM8F("v_bfi_b32 %0, %1, %0, %2"))
KERNEL(k_cmp_4cnd_nop, FDECL, _Pragma("unroll") for (int k = 0; k < 8; k += 4) asm volatile("v_cmp_lt_f32 vcc, %0, %4\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %5, vcc\n\tv_cndmask_b32 %1, %1, %5, vcc\n\tv_cndmask_b32 %2, %2, %5, vcc\n\tv_cndmask_b32 %3, %3, %5, vcc" : "+v"(x[k]), "+v"(x[k + 1]), "+v"(x[k + 2]), "+v"(x[k + 3]) : "v"(y), "v"(z) : "vcc");)
KERNEL(k_min, FDECL, ASM8F("v_min_f32 %0, %0, %1"))
KERNEL(k_min3, FDECL, ASM8F("v_min3_f32 %0, %0, %1, %2"))
KERNEL(k_med3, FDECL, ASM8F("v_med3_f32 %0, %0, %1, %2"))
KERNEL(k_fmac, FDECL, ASM8F("v_fmac_f32 %0, %1, %2"))
KERNEL(k_fma_sgpr, FDECL, ASM8F("v_fma_f32 %0, %0, s8, %2"))
KERNEL(k_sub, FDECL, ASM8F("v_sub_f32 %0, %0, %1"))
KERNEL(k_sub_sgpr, FDECL, ASM8F("v_subrev_f32 %0, s8, %0"))
KERNEL(k_cos, FDECL, ASM8F("v_cos_f32 %0, %0"))
KERNEL(k_cvt_i32, FDECL, ASM8F("v_cvt_i32_f32 %0, %0"))

// ---- where does a select get slow? (round 2) ----
// raw select throughput with the mask in an SGPR pair / in VCC through the VOP3 encoding
KERNEL(k_cnd_sgpr_only, FDECL, ASM8F("v_cndmask_b32_e64 %0, %0, %1, s[20:21]"))
KERNEL(k_cnd_vcc_e64, FDECL, ASM8F("v_cndmask_b32_e64 %0, %0, %1, vcc"))
// a compare, nine independent FMAs, then the select that uses it (mask in VCC / in an SGPR pair): reported per group of 11
KERNEL(k_cmp_far_vcc, FDECL, _Pragma("unroll") for (int k = 0; k < 8; k++) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(x[k]) : "v"(y), "v"(z) : "vcc");)
KERNEL(k_cmp_far_sgpr, FDECL, _Pragma("unroll") for (int k = 0; k < 8; k++) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_cndmask_b32 %0, %0, %2, s[20:21]" : "+v"(x[k]) : "v"(y), "v"(z) : "s20", "s21");)
KERNEL(k_fma_x10, FDECL, _Pragma("unroll") for (int k = 0; k < 8; k++) asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(y), "v"(z));)
// two compares combined on the scalar unit, then two selects (the tail of a sphere test): mask in VCC / kept in an SGPR pair: per group of 4 VALU
KERNEL(k_and_2cnd_vcc, FDECL, _Pragma("unroll") for (int k = 0; k < 8; k += 2) asm volatile("v_cmp_lt_f32 s[22:23], %0, %2\n\tv_cmp_nlt_f32 vcc, %1, %3\n\ts_and_b64 vcc, s[22:23], vcc\n\tv_cndmask_b32 %0, %0, %3, vcc\n\tv_cndmask_b32 %1, %1, %2, vcc" : "+v"(x[k]), "+v"(x[k + 1]) : "v"(y), "v"(z) : "vcc", "s22", "s23");)
KERNEL(k_and_2cnd_sgpr, FDECL, _Pragma("unroll") for (int k = 0; k < 8; k += 2) asm volatile("v_cmp_lt_f32 s[22:23], %0, %2\n\tv_cmp_nlt_f32 s[20:21], %1, %3\n\ts_and_b64 s[20:21], s[22:23], s[20:21]\n\tv_cndmask_b32 %0, %0, %3, s[20:21]\n\tv_cndmask_b32 %1, %1, %2, s[20:21]" : "+v"(x[k]), "+v"(x[k + 1]) : "v"(y), "v"(z) : "s20", "s21", "s22", "s23");)
KERNEL(k_mov, FDECL, ASM8F("v_mov_b32 %0, %1"))
KERNEL(k_fma_neg, FDECL, ASM8F("v_fma_f32 %0, %0, %1, -%2"))
KERNEL(k_cmp_e64_neg, FDECL, _Pragma("unroll") for (int k = 0; k < 8; k++) asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, -%1" : "+v"(x[k]) : "v"(y), "v"(z) : "s20", "s21");)

// packed fp32: x is a float2 pair per chain
__global__ __launch_bounds__(256) void k_pk_fma(unsigned *out, unsigned seed) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 x[8];
    f2 y = {__uint_as_float(seed | 0x3f800000u), 1.0001f}, z = {0.5f, 0.25f};
    for (int k = 0; k < 8; k++) x[k] = f2{threadIdx.x * 0.001f + k, 1.0f};
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int r = 0; r < REP / 8; r++) {
#pragma unroll
            for (int k = 0; k < 8; k++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(y), "v"(z));
        }
    }
    float acc = 0;
    for (int k = 0; k < 8; k++) acc += x[k].x + x[k].y;
    if (acc == 1.2345f) out[threadIdx.x] = 1;
}
__global__ __launch_bounds__(256) void k_pk_mul(unsigned *out, unsigned seed) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 x[8];
    f2 y = {__uint_as_float(seed | 0x3f800000u), 1.0001f};
    for (int k = 0; k < 8; k++) x[k] = f2{threadIdx.x * 0.001f + k, 1.0f};
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int r = 0; r < REP / 8; r++) {
#pragma unroll
            for (int k = 0; k < 8; k++) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x[k]) : "v"(y));
        }
    }
    float acc = 0;
    for (int k = 0; k < 8; k++) acc += x[k].x + x[k].y;
    if (acc == 1.2345f) out[threadIdx.x] = 1;
}
__global__ __launch_bounds__(256) void k_mad_u64_u32(unsigned *out, unsigned seed) {
    unsigned long long x[8];
    unsigned y = seed | 1u;
    for (int k = 0; k < 8; k++) x[k] = threadIdx.x * 2654435761ull + k;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int r = 0; r < REP / 8; r++) {
#pragma unroll
            for (int k = 0; k < 8; k++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(x[k]) : "v"(y) : "vcc");
        }
    }
    unsigned long long acc = 0;
    for (int k = 0; k < 8; k++) acc ^= x[k];
    if (acc == 0x12345678ull) out[threadIdx.x] = 1;
}

typedef void (*kern_t)(unsigned *, unsigned);
int main() {
    unsigned *d;
    hipMalloc(&d, 4096);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double clk = prop.clockRate * 1e3;  // Hz
    printf("device %s CUs %d clock %.0f MHz\n", prop.name, cus, clk / 1e6);
    struct E { const char *name; kern_t k; };
    std::vector<E> ks = {{"v_fma_f32", k_fma}, {"v_add_f32", k_add}, {"v_mul_f32", k_mul}, {"v_pk_fma_f32", k_pk_fma}, {"v_pk_mul_f32", k_pk_mul},
        {"v_sqrt_f32", k_sqrt}, {"v_rcp_f32", k_rcp}, {"v_rsq_f32", k_rsq}, {"v_sin_f32", k_sin}, {"v_cndmask_b32(no cmp)", k_cndmask}, {"v_max_f32", k_max},
        {"cmp+cndmask vcc (pair)", k_cmp_cnd_vcc}, {"cmp+cndmask sgpr (pair)", k_cmp_cnd_sgpr}, {"v_cmp vcc", k_cmp_only}, {"v_cmp sgpr", k_cmp_only_sgpr},
        {"v_cndmask after 1 cmp", k_cnd_after_cmp}, {"cmp + 2 cndmask (x4)", k_cmp_2cnd}, {"cmp + 4 cndmask (x2)", k_cmp_4cnd},
        {"cmp + 4 cndmask sgpr (x2)", k_cmp_4cnd_sgpr}, {"cmp + nop + 4 cndmask (x2)", k_cmp_4cnd_nop}, {"v_bfi_b32", k_bfi}, {"v_min_f32", k_min}, {"v_min3_f32", k_min3}, {"v_med3_f32", k_med3}, {"v_fmac_f32", k_fmac},
        {"v_fma_f32 sgpr src", k_fma_sgpr}, {"v_sub_f32", k_sub}, {"v_subrev_f32 sgpr", k_sub_sgpr}, {"v_cos_f32", k_cos}, {"v_cvt_i32_f32", k_cvt_i32},
        {"v_mul_lo_u32", k_mul_lo}, {"v_mul_hi_u32", k_mul_hi}, {"v_mul_u32_u24", k_mul_u24}, {"v_mad_u32_u24", k_mad_u24}, {"v_mad_u64_u32", k_mad_u64_u32},
        {"v_xor_b32", k_xor}, {"v_lshrrev_b32", k_lshr}, {"v_add_u32", k_add_u32}, {"v_add3_u32", k_add3}, {"v_xad_u32", k_xad}, {"v_lshl_add_u32", k_lshl_add},
        {"v_alignbit_b32", k_alignbit}, {"v_bfe_u32", k_bfe}, {"v_cvt_f32_u32", k_cvt_f32_u32}, {"v_perm_b32", k_perm}, {"v_and_or_b32", k_and_or},
        {"v_cndmask e64 sgpr mask only", k_cnd_sgpr_only}, {"v_cndmask e64 vcc", k_cnd_vcc_e64}, {"cmp, 9 fma, cndmask vcc (11)", k_cmp_far_vcc},
        {"cmp, 9 fma, cndmask sgpr (11)", k_cmp_far_sgpr}, {"11 fma", k_fma_x10}, {"2 cmp, s_and, 2 cndmask vcc (x4)", k_and_2cnd_vcc},
        {"2 cmp, s_and, 2 cndmask sgpr (x4)", k_and_2cnd_sgpr}, {"v_mov_b32", k_mov}, {"v_fma_f32 neg src", k_fma_neg}, {"v_cmp e64 neg", k_cmp_e64_neg}};
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    const int blocks = cus * 8;  // 8 blocks x 4 waves = 32 waves/CU = 8 waves/SIMD
    for (auto &e : ks) {
        hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, d, 12345u);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, d, 12345u);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        // per SIMD: waves = 8, each issues ITERS*REP instructions
        double insts_per_simd = 8.0 * ITERS * REP;
        double cycles = ms * 1e-3 * clk;
        printf("%-26s %8.3f ms  %6.2f cycles per wave-instruction per SIMD (at %.0f MHz nominal)\n", e.name, ms, cycles / insts_per_simd, clk / 1e6);
    }
    return 0;
}
