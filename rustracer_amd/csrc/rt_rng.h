// rt_rng.h — the counter-based RNG of the device path (replaces rand 0.7 SmallRng + gen_range,
// SURVEY §8a row a14; the reference streams are entropy-seeded, lambertian_diffuse.rs:22 etc.).
// Specified in DESIGN.md §RNG; the CPU oracle (oracle/rt_oracle.hpp) implements the same integer
// function independently.
//
//   pixel_key = mix32(mix32(seed ^ 0x52545243) + pixel_index)
//   draw pair of (sample s, slot j):  h = mix32(pixel_key + (s*256 + j) * 0x9E3779B9)
//        u_a = (h >> 8) * 2^-24
//        g = h * 0x2C1B3C6D;  g ^= g >> 15;   u_b = (g >> 8) * 2^-24
//   slot 0 = pixel jitter (u_a = x offset drawn first, u_b = y offset; lib.rs:283,285)
//   slot 1 + bounce = material draws of segment `bounce`: Lambertian/Metal u_a -> a in [0,TAU),
//        u_b -> z in [-1,1) (a drawn first, lambertian_diffuse.rs:40-41); Dielectric u_a = the
//        reflect/refract coin (dielectric.rs:94).
// One mix32 per pair: the counter walks in steps of the golden ratio, no per-path state.
#pragma once
#include <stdint.h>

#ifndef RT_HD
#ifdef __HIPCC__
#define RT_HD __host__ __device__ __forceinline__
#else
#define RT_HD inline
#endif
#endif

#define RT_RNG_GOLDEN 0x9E3779B9u
#define RT_RNG_SLOTS 256u
#define RT_RNG_MUL2 0x2C1B3C6Du

RT_HD uint32_t rt_mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
RT_HD uint32_t rt_seed_key(uint32_t seed) { return rt_mix32(seed ^ 0x52545243u); }
RT_HD uint32_t rt_pixel_key(uint32_t seed_key, uint32_t pixel_index) { return rt_mix32(seed_key + pixel_index); }
// counter of (sample, slot): pixel_key + (sample*256 + slot)*GOLDEN; advance by GOLDEN per slot, 256*GOLDEN per sample
RT_HD uint32_t rt_counter(uint32_t pixel_key, uint32_t sample, uint32_t slot) {
    return pixel_key + (sample * RT_RNG_SLOTS + slot) * RT_RNG_GOLDEN;
}
RT_HD uint32_t rt_second(uint32_t h) {
    uint32_t g = h * RT_RNG_MUL2;
    return g ^ (g >> 15);
}
RT_HD float rt_unit(uint32_t bits) { return float(bits >> 8) * (1.0f / 16777216.0f); }  // U[0,1) on a 24-bit grid (exact in binary32)
