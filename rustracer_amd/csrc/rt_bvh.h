// rt_bvh.h — host-built linear BVH over spheres + triangles (extension: the reference scans
// linearly, raytracer_core/src/shapes/ray.rs:47-67). Declared here, built in rt_bvh.cpp.
#pragma once
#include <stdint.h>

#include <vector>

#include "../../include/rthip.h"
#include "rt_device.h"

// scenes with more bounded primitives than this get a BVH (below it the scalar-load linear scan wins)
#define RT_BVH_THRESHOLD 64ull

// returns the number of traversal stack levels the tree needs (<= RT_BVH_MAX_DEPTH), 0 for an empty scene
// 64-byte float nodes -> the 32-byte quantised nodes the kernels read (conservative: see DevBvhNodeQ). false on failure.
bool rt_bvh_quantize(const std::vector<DevBvhNode> &nodes, std::vector<DevBvhNodeQ> &out);
// a bound of a quantised node as the kernels see it: origin + q * step, in double
double rt_bvh_dequantize(uint32_t frame, uint32_t q8);
uint32_t rt_build_bvh(const rthip_scene_desc *scene, std::vector<DevBvhNode> &nodes, std::vector<uint32_t> &prims);
