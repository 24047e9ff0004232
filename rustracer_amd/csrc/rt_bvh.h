// rt_bvh.h — host-built linear BVH over spheres + triangles (extension: the reference scans
// linearly, raytracer_core/src/shapes/ray.rs:47-67). Declared here, built in rt_bvh.cpp.
#pragma once
#include <stdint.h>

#include <vector>

#include "../../include/rthip.h"
#include "rt_device.h"

// A scene gets a BVH when the linear scan would cost more than this many sphere tests per ray; a triangle costs the scan about
// four sphere tests (its record comes from global memory through scalar loads, spheres sit in LDS). Measured on the MI355X
// (tools/bvh_crossover.py, 1080p, 32 spp, depth 8): 82 spheres scan 4.9 ms / BVH 5.4 ms, 101 spheres 5.9 / 5.7, 170 spheres
// 10.3 / 7.2; 17 spheres + 20 triangles 4.5 / 4.7, + 80 triangles 17.9 / 5.6.
#ifndef RT_BVH_THRESHOLD
#define RT_BVH_THRESHOLD 96ull
#endif
inline bool rt_wants_bvh(uint64_t n_spheres, uint64_t n_triangles) { return n_spheres + 4ull * n_triangles > RT_BVH_THRESHOLD; }

// 64-byte float nodes -> the 32-byte quantised nodes the kernels read (conservative: see DevBvhNodeQ). false on failure.
bool rt_bvh_quantize(const std::vector<DevBvhNode> &nodes, std::vector<DevBvhNodeQ> &out);
// a bound of a quantised node as the kernels see it: origin + q * step, in double
double rt_bvh_dequantize(uint32_t frame, uint32_t q8);
// returns the number of traversal stack levels the tree needs (<= RT_BVH_MAX_DEPTH), 0 for an empty scene
uint32_t rt_build_bvh(const rthip_scene_desc *scene, std::vector<DevBvhNode> &nodes, std::vector<uint32_t> &prims);
// top of the tree first, breadth-first (see rt_bvh.cpp); returns how many nodes are in that order
#define RT_BVH_TOP_MAX 255u
uint32_t rt_bvh_top_first(std::vector<DevBvhNode> &nodes, uint32_t max_top);
