// rt_rng.h — the counter-based RNG of the device path (replaces rand 0.7 SmallRng + gen_range,
// SURVEY §8a row a14; the reference streams are entropy-seeded, lambertian_diffuse.rs:22 etc.).
// value = f(seed, pixel_index, sample, counter); counter 0,1 = pixel jitter x,y (lib.rs:283,285),
// 2 + 2*bounce + dim = material draws of segment `bounce`. Specified in DESIGN.md §RNG.
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
RT_HD uint32_t rt_path_key(uint32_t pixel_key, uint32_t sample) { return rt_mix32(pixel_key + sample * RT_RNG_GOLDEN); }
RT_HD uint32_t rt_rng_u32(uint32_t key, uint32_t counter) { return rt_mix32(key + counter * RT_RNG_GOLDEN); }
// U[0,1) on a 24-bit grid (exact in binary32)
RT_HD float rt_rng_unit(uint32_t key, uint32_t counter) { return float(rt_rng_u32(key, counter) >> 8) * (1.0f / 16777216.0f); }
