// Micro-benchmark: does a wave64 VALU instruction cost less when only part of the wave's lanes are switched on (gfx950)?
// The same loop of independent v_fmac_f32 / v_fma_f32 (VOP3) / v_cndmask chains as valu_rates.hip, executed under a lane mask.
// Reports cycles per wave instruction per SIMD at 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP 64
#define ITERS 4096
template <int KIND>
__global__ __launch_bounds__(256) void k(unsigned *out, unsigned seed, unsigned long long mask, unsigned long long mask2) {
    float x[8];
    float y = __uint_as_float(seed | 0x3f800000u), z = 0.5f;
    for (int i = 0; i < 8; i++) x[i] = threadIdx.x * 0.001f + i;
    // (mask2 != 0: every second wave of the workgroup runs under mask2 instead — do sparse waves cost the same next to full ones?)
    if ((((threadIdx.x >> 6) & 1u) && mask2 ? mask2 >> (threadIdx.x & 63u) : mask >> (threadIdx.x & 63u)) & 1ull) {
        for (int it = 0; it < ITERS; it++) {
#pragma unroll
            for (int r = 0; r < REP / 8; r++) {
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    if (KIND == 0) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[i]) : "v"(y), "v"(z));
                    if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y), "v"(z));
                    if (KIND == 2) asm volatile("v_sqrt_f32 %0, %0" : "+v"(x[i]));
                }
            }
        }
    }
    unsigned acc = 0;
    for (int i = 0; i < 8; i++) acc ^= __float_as_uint(x[i]);
    if (acc == 0x12345678u) out[threadIdx.x] = acc;
}
typedef void (*kern_t)(unsigned *, unsigned, unsigned long long, unsigned long long);
int main() {
    unsigned *out;
    hipMalloc(&out, 4096);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const double ghz = prop.clockRate * 1e-6;
    const int blocks = prop.multiProcessorCount * 8;  // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    struct M { const char *name; unsigned long long m, m2; } masks[] = {
        {"half the waves full, half lanes 0-4", ~0ull, 0x1full}, {"half lanes 0-15, half lanes 0-4", 0xffffull, 0x1full},
        {"all 64", ~0ull}, {"lanes 0-31", 0xffffffffull}, {"lanes 32-63", 0xffffffff00000000ull}, {"lanes 0-15", 0xffffull},
        {"lanes 0-4", 0x1full}, {"lanes 0 and 32", 0x100000001ull}, {"lanes 0,16,32,48", 0x0001000100010001ull}, {"every 2nd", 0x5555555555555555ull},
        {"1 lane", 0x1ull}, {"lanes 0-5", 0x3full}, {"lanes 0-6", 0x7full}, {"lanes 0-7", 0xffull}, {"lanes 0-9", 0x3ffull}, {"lanes 0-11", 0xfffull},
        {"lanes 20-24", 0x1f00000ull}, {"every 8th", 0x0101010101010101ull}, {"every 4th", 0x1111111111111111ull}};
    kern_t ks[3] = {k<0>, k<1>, k<2>};
    const char *kn[3] = {"v_fmac_f32 (VOP2)", "v_fma_f32 (VOP3)", "v_sqrt_f32"};
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int kk = 0; kk < 3; kk++)
        for (auto &m : masks) {
            hipLaunchKernelGGL(ks[kk], dim3(blocks), dim3(256), 0, 0, out, 1u, m.m, m.m2);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(ks[kk], dim3(blocks), dim3(256), 0, 0, out, 1u, m.m, m.m2);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double insts_per_simd = double(ITERS) * REP * 8.0;  // 8 waves per SIMD, each ITERS * REP instructions
            printf("%-20s %-38s %.2f cycles per wave instruction per SIMD (at the nominal %.2f GHz)\n", kn[kk], m.name, ms * 1e-3 * ghz * 1e9 / insts_per_simd, ghz);
        }
    return 0;
}
