// rt_prepass.h — primary-visibility lists for MODE_BVH (extension; the scan to equal is raytracer_core/src/shapes/ray.rs:47-67).
//
// OPT-IN (RTHIP_F_TILE_LISTS): bit-identical to the plain traversal and measured SLOWER (DESIGN.md section 5, round 3,
// profiles/r03_bvh_pmc_summary.md); the default kernels carry none of it (render_kernel<.., LISTS = false>).
//
// Every primary ray of a pixel — whatever its sample's jitter (lib.rs:282-285) — starts at the camera and runs inside the thin
// pyramid through the pixel's footprint, and the 64 pixels of a wave's 8x8 tile share one slightly wider pyramid for the whole
// launch. So the tree is walked ONCE per tile with that pyramid: `prepass_kernel` walks it breadth-first (one node per lane),
// collects up to RT_PRE_COLLECT leaves whose (quantised) boxes the tile's pyramid touches, keeps the RT_PRE_LEAVES nearest
// in order of (near, reference) — near = a lower bound of the distance from the camera to the leaf's box — and gives every
// pixel a 64-bit mask of the kept leaves ITS pyramid touches plus a cut distance: the smallest near of a leaf that touches
// the pixel's pyramid and was not kept. A primary ray tests the leaves of its pixel's mask in list order through the leaf code
// of the traversal (leaf_test, rt_kernels.hip), stops at the first leaf whose near lies beyond its best hit, and is done if that
// hit is nearer than the cut; otherwise (and on tiles marked RT_PRE_OVERFLOW) it traverses as before. Bounce rays always traverse.
// Why the hit is the same: a traversal tests a leaf's primitives only after the ray has passed the slab test of that leaf's
// box; a ray inside the pyramid that passes that test makes the pyramid touch the box, so the candidates contain every
// primitive the traversal would test that could be nearer than the cut, and the closest hit under "smaller t, then larger
// index" is the same (bit for bit: same test code).
//
// The geometry below is shared by the device kernel and its host mirror (rt_prepass_host.cpp: the checker of the lists,
// tests/test_prepass.py), hence plain C++ with RT_HD.
#pragma once
#include <math.h>
#include <stdint.h>

#include "rt_device.h"

#ifdef __HIP__ /* the compiler is in HIP mode (hipcc treats .cpp files so too) */
#define RT_PRE_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define RT_PRE_HD inline
#endif

#define RT_PRE_LEAVES 64u          /* leaves per tile list: one mask bit each */
#define RT_PRE_COLLECT 256u        /* leaves the tile walk collects before it keeps the RT_PRE_LEAVES nearest; more -> overflow */
#define RT_PRE_FRONT 192u          /* inner nodes of one breadth-first level of the tile walk; more -> overflow */
#define RT_PRE_MARGIN 0.25f        /* pixels by which a footprint is widened on every side (FP32 noise is ~1e-3 of that) */
#define RT_PRE_OVERFLOW 0xffffffffu
#ifndef RT_PRE_MIN_SPP
#define RT_PRE_MIN_SPP 4u           /* a pass that starts a frame makes lists from this many samples per pixel on */
#endif

// What prepass_kernel leaves per wave tile (1280 B).
//   ent[i]  = {leaf reference (RT_BVH_LEAF_BIT | first << 2 | count - 1), near}: the tile's leaves in order of `near`, a lower
//             bound of the distance from the camera to the leaf's box (ties: smaller reference first); unused entries
//             {RT_BVH_DONE, +inf}; ent[0].ref == RT_PRE_OVERFLOW: no list for this tile, its primary rays traverse
//   mask[p] = the entries pixel p's pyramid touches (p = lane of the render kernel: x & 7 | (row & 7) << 3)
//   cut[p]  = the smallest `near` of a leaf that touches pixel p's pyramid but did not fit the list (+inf if none): a primary
//             ray whose closest candidate hit is nearer than cut[p] is done, any other goes on with a normal traversal
struct RtTileEnt {
    uint32_t ref;
    float near;
};
struct RtTilePre {
    RtTileEnt ent[RT_PRE_LEAVES];
    unsigned long long mask[64];
    float cut[64];
};

// the rays from `o` through a rectangle of the image plane: four side planes through o (normals pointing inward) and the
// central direction
struct RtPyramid {
    float o[3];
    float n[5][3];  // [4] = the central direction (a box wholly behind the apex is outside)
};

RT_PRE_HD float rt_pre_bits_f(uint32_t u) {
    return __builtin_bit_cast(float, u);
}

// Pyramid of the pixel rectangle [x0, x1] x [y0, y1] (pixel units; a pixel x covers [x, x + 1): the jitter is in [0, 1)).
// Directions as the kernel builds them: d = d00 + (H/(W-1)) * x + (V/(H-1)) * y (camera.rs:43-50, lib.rs:282-285).
RT_PRE_HD void rt_pre_pyramid(const KernelParams &P, float x0, float x1, float y0, float y1, RtPyramid &py) {
    py.o[0] = P.ox, py.o[1] = P.oy, py.o[2] = P.oz;
    const float d00[3] = {P.d00x, P.d00y, P.d00z}, hw[3] = {P.hwx, P.hwy, P.hwz}, vh[3] = {P.vhx, P.vhy, P.vhz};
    float c[4][3];  // corners 00, 10, 01, 11
    for (int k = 0; k < 3; k++) {
        c[0][k] = fmaf(vh[k], y0, fmaf(hw[k], x0, d00[k]));
        c[1][k] = fmaf(vh[k], y0, fmaf(hw[k], x1, d00[k]));
        c[2][k] = fmaf(vh[k], y1, fmaf(hw[k], x0, d00[k]));
        c[3][k] = fmaf(vh[k], y1, fmaf(hw[k], x1, d00[k]));
    }
    for (int k = 0; k < 3; k++) py.n[4][k] = (c[0][k] + c[1][k]) + (c[2][k] + c[3][k]);
    const int side[4][2] = {{0, 2}, {1, 3}, {0, 1}, {2, 3}};  // left, right, bottom, top
    // (every multiply-add below is an explicit fmaf and no other expression can be contracted, so that the host mirror and
    // the device kernel compute the same bits whatever their compilers' contraction defaults are)
    for (int s = 0; s < 4; s++) {
        const float *a = c[side[s][0]], *b = c[side[s][1]];
        const float n[3] = {fmaf(a[1], b[2], -(a[2] * b[1])), fmaf(a[2], b[0], -(a[0] * b[2])), fmaf(a[0], b[1], -(a[1] * b[0]))};
        const float towards = fmaf(n[2], py.n[4][2], fmaf(n[1], py.n[4][1], n[0] * py.n[4][0]));
        const float sgn = towards < 0.0f ? -1.0f : 1.0f;
        for (int k = 0; k < 3; k++) py.n[s][k] = n[k] * sgn;
    }
}

// true only if the box [lo, hi] lies wholly on the outer side of one plane, by more than the rounding of this very test
// (16x the worst-case error of the three products and sums). NaN anywhere -> false (conservative).
RT_PRE_HD bool rt_pre_box_outside(const RtPyramid &py, const float lo[3], const float hi[3]) {
    for (int s = 0; s < 5; s++) {
        float far = 0.0f, err = 0.0f;
        for (int k = 0; k < 3; k++) {
            const float n = py.n[s][k];
            const float fa = n * (lo[k] - py.o[k]), fb = n * (hi[k] - py.o[k]);
            far = far + fmaxf(fa, fb);
            err = fmaf(fabsf(n), fmaxf(fabsf(lo[k]), fabsf(hi[k])) + fabsf(py.o[k]), err);
        }
        if (far < -4e-6f * err) return true;
    }
    return false;
}

// the two child boxes of a quantised node as floats (DevBvhNodeQ: bound = origin + q * 2^e; the low byte of the origin's
// pattern IS e and is part of the origin — the builder accounts for it). empty[c]: the child has no box (lo byte > hi byte).
RT_PRE_HD void rt_pre_decode(const DevBvhNodeQ &nd, float lo[2][3], float hi[2][3], bool empty[2]) {
    empty[0] = empty[1] = false;
    for (int k = 0; k < 3; k++) {
        const float origin = rt_pre_bits_f(nd.frame[k]);
        const float step = ldexpf(1.0f, int(int8_t(nd.frame[k] & 0xffu)));
        const uint32_t q = nd.q[k];
        lo[0][k] = fmaf(float(q & 0xffu), step, origin), hi[0][k] = fmaf(float((q >> 8) & 0xffu), step, origin);
        lo[1][k] = fmaf(float((q >> 16) & 0xffu), step, origin), hi[1][k] = fmaf(float(q >> 24), step, origin);
        empty[0] = empty[0] || (q & 0xffu) > ((q >> 8) & 0xffu);
        empty[1] = empty[1] || ((q >> 16) & 0xffu) > (q >> 24);
    }
}

// A lower bound of the distance from the apex `o` to any point of the box (so of the t, along a unit direction, of any hit
// inside it), a hair below the float result to cover its own rounding; 0 when the apex is inside, or on NaN.
RT_PRE_HD float rt_pre_box_near(const float o[3], const float lo[3], const float hi[3]) {
    float d2 = 0.0f, mag = 0.0f;
    for (int k = 0; k < 3; k++) {
        const float g = fmaxf(fmaxf(lo[k] - o[k], o[k] - hi[k]), 0.0f);
        d2 = fmaf(g, g, d2);
        mag = mag + (fmaxf(fabsf(lo[k]), fabsf(hi[k])) + fabsf(o[k]));
    }
    const float slack = 4e-6f * mag;
    return fmaxf(fmaf(sqrtf(d2), 0.99999f, -slack), 0.0f);
}
