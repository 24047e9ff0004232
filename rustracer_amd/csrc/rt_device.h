// rt_device.h — device-side data layout of the flattened scene and the kernel parameter block.
// Shared by rthip_api.cpp (host flattening/upload) and rt_kernels.hip (gfx950 kernels).
//
// HBM layout (all arrays 16-byte aligned, read-only during a render):
//   sph_geom  [n_spheres] float4 {cx, cy, cz, r*r}          -> the closest-hit scan reads ONLY this
//   sph_shade [n_spheres] 4 x float4 {c, 1/r} {albedo/pi*3, param} {kind, r1 | texture index, 1/ior, material} {albedo/pi, shininess}: the merged
//             sphere+material shading record, read once per hit by the hitting lane
//   (scenes of <= RT_LDS_SPHERES spheres: both arrays are staged in LDS once per workgroup)
//   pln_geom  [n_planes]  2 x float4 {nx,ny,nz, n.p0} {mat(bits), 0,0,0} + tangent frame 2 x float4
//   tri_geom  [n_tris]    3 x float4 {v0, e1, e2} w = mat(bits)/0/0
//   materials [n_mat]     3 x float4, see DevMaterial
//   texels    RGBA8, one u32 per texel, 8x8-texel tiles (64 consecutive u32 = 256 B per tile)
//   tex_info  [n_tex]     {width, height, tiles_per_row, first_texel}
//   bvh_nodes [n_nodes]   32 B each (both child boxes, 8-bit quantised), bvh_recs [n_prims] 3 x float4 in leaf order, see DevBvhNodeQ
#pragma once
#include <stdint.h>

#define RT_LDS_SPHERES 256u
#define RT_SHADE_RECS 4u  // float4 records per sphere in sph_shade (64 B: the record address is a shift)
#define RT_RING_ROWS 4u  // MODE_BVH: sample rows in a wave's radiance ring
// samples per slice: 64, doubled until at most 16 slices cover spp; a function of spp alone, always a power of two
#ifndef RT_SLICE_SPP
#define RT_SLICE_SPP 64u
#endif
#define RT_MAX_SLICES 16u
#ifndef RT_ORDER_MIN_SPP
#define RT_ORDER_MIN_SPP 16u /* scan scenes: passes of at least this many samples are launched in order of estimated cost */
#endif
// scenes with a tree: sample groups of (at least) RT_TREE_SLICE_SPP samples, at most RT_TREE_SLICES of them, always traced by different waves
#ifndef RT_TREE_SLICE_SPP
#define RT_TREE_SLICE_SPP 4u
#endif
#ifndef RT_TREE_SLICES
#define RT_TREE_SLICES 4u
#endif
#ifndef RT_TREE_SLICE_BELOW_WAVES
#define RT_TREE_SLICE_BELOW_WAVES 300000u /* ... launches of fewer wave tiles than this are sliced */
#endif
#ifndef RT_SLICE_BELOW_BLOCKS
#define RT_SLICE_BELOW_BLOCKS 20000u
#endif /* launches of fewer workgroups than this are sliced */
// pixels of a 256-thread workgroup: four 8x8 wave tiles; RT_TILE_W * RT_TILE_H = 256, both multiples of 8
#ifndef RT_TILE_W
#define RT_TILE_W 32u
#endif
#define RT_TILE_H (256u / RT_TILE_W)
#define RT_T_MIN 0.001f      /* raytracer_core/src/shapes/ray.rs:5 */
#define RT_T_MAX 100000.0f   /* raytracer_core/src/shapes/ray.rs:6 */

struct rt_f4 { float x, y, z, w; };

// material record: 3 x 16 B
struct DevMaterial {
    float tr, tg, tb;      // throughput multiplier albedo/pi*3  (lambertian_diffuse.rs:60-65, metal.rs:51-55, dielectric.rs:61-65)
    float param;           // METAL fuzziness | DIELECTRIC ior | TEXTURE scale
    uint32_t kind;         // RTHIP_MAT_*
    uint32_t texture;      // texture index
    float shininess;       // ext: Phong exponent
    float r1;              // DIELECTRIC: ((1-ior)/(1+ior))^2   dielectric.rs:48-49 | TEXTURE: the texture index, as bits
    float ar, ag, ab;      // albedo/pi (direct-light term of the point-light extension)
    float inv_ior;         // DIELECTRIC: 1/ior  dielectric.rs:81
};

struct DevTexInfo {
    uint32_t width, height;
    uint32_t tiles_per_row;  // ceil(width/8)
    uint32_t first_texel;    // offset into texels (in u32)
    uint32_t magic_w, magic_h;  // floor(2^32 / width), floor(2^32 / height) (0xffffffff for 1): Texture::wrap's `%` without a division
    float width_f, height_f;    // float(width), float(height): the factor of Texture::wrap's val * scale * bound
};

struct DevLight { float px, py, pz, _p0, ir, ig, ib, _p1; };

// BVH node, 64 B = one cache line = 4 x float4: the boxes of BOTH children and a reference to each,
// so that one fetch decides both descents (half the dependent loads of a box-per-node layout).
// A reference is an inner node index (< RT_BVH_DONE; the root is node 0) or, with RT_BVH_LEAF_BIT
// set, a leaf: (first << 2) | (count - 1), primitives [first, first + count) of the BVH-ordered
// record array bvh_recs (count <= RT_BVH_LEAF, first < 2^29).
// bvh_recs: 3 x float4 per primitive, in leaf order (no index indirection):
//   triangle {v0, global shape index} {e1, kind = 1} {e2, 0};  sphere {c, global shape index} {r*r, 0, 0, kind = 0} {0}
#ifndef RT_BVH_LEAF
#define RT_BVH_LEAF 2u
#endif
#ifndef RT_BVH_MAX_DEPTH
#define RT_BVH_MAX_DEPTH 32
#endif
#define RT_BVH_LEAF_BIT 0x80000000u
#define RT_BVH_DONE 0x7fffffffu
#define RT_BVH_MAX_PRIMS (1ull << 29)
// The kernels read a 32-byte QUANTISED form of the node (DevBvhNodeQ): the traversal is bound by the number of vector-memory
// instructions per node visit (a divergent 16-byte load occupies the CU's address unit for ~16-27 cycles whatever it hits),
// and 32 bytes are two loads instead of four. Per axis k: frame[k] is the bit pattern of the float origin o_k, whose low 8
// bits double as e_k, the (signed) exponent of the power-of-two step s_k = 2^e_k; q[k] = lo0 | hi0 << 8 | lo1 << 16 |
// hi1 << 24, the bounds of the two children on that axis in steps from the origin: bound = o_k + q * s_k. The builder puts
// the origin below the boxes and rounds lower bounds down and upper bounds up, so a quantised box always CONTAINS the float
// box (the traversal visits a superset of nodes and returns the same hits). An empty child has lo = 255 > hi = 0.
struct DevBvhNodeQ {
    uint32_t frame[3];
    uint32_t ref0;
    uint32_t q[3];
    uint32_t ref1;
};

// host-side form produced by the builder (float boxes), quantised by rt_bvh_quantize
struct DevBvhNode {
    float lo0[3];
    uint32_t ref0;
    float hi0[3];
    uint32_t ref1;
    float lo1[3];
    uint32_t _pad0;
    float hi1[3];
    uint32_t _pad1;
};

struct KernelParams {
    // camera (camera.rs:43-50)
    float ox, oy, oz;
    float d00x, d00y, d00z;
    float hx, hy, hz;
    float vx, vy, vz;
    float inv_wm1, inv_hm1;  // 1/(width-1), 1/(height-1)   lib.rs:283,285
    float hwx, hwy, hwz;     // horizontal / (width-1): the direction change per unit of x jitter
    float vhx, vhy, vhz;     // vertical / (height-1)
    float inv_nb;            // 1/nb_samples                lib.rs:294
    uint32_t width, height;
    uint32_t spp, max_depth, seed, sample0, flags;
    uint32_t band_rows, band_first, band_stride, owned_rows;
    uint32_t accumulate;  // add the prior running sum (lib.rs:289-291)
    uint32_t count_executed;  // RTHIP_F_COUNT_EXECUTED: the instrumented kernel takes the production kernel's loops (tests culled by them are not counted)
    // scene
    uint32_t n_spheres, n_planes, n_tris, n_lights, n_bvh_nodes;
    uint32_t tree_scene;  // the scene has a tree, used or not (RTHIP_F_NO_BVH clears n_bvh_nodes): its linear scan is MODE_GLOBAL, which tests spheres
                          // with the expressions of the tree's leaves (sphere_roots), so that tree == scan holds bit for bit
    uint32_t bvh_levels;  // traversal stack levels = depth of the tree
    uint32_t bvh_top_nodes;  // nodes [0, bvh_top_nodes) are the top of the tree in breadth-first order; the render kernel stages
                             // min(bvh_top_nodes, its LDS budget) of them in LDS (rt_launch_render)
    uint32_t ring_rows;   // MODE_BVH: sample rows in a wave's radiance ring, a power of two <= 16 (set by rt_launch_render)
    const rt_f4 *sph_geom;
    const rt_f4 *sph_shade;
    const rt_f4 *pln_geom;
    const rt_f4 *tri_geom;
    const DevMaterial *materials;
    const uint32_t *texels;
    const DevTexInfo *tex_info;
    const DevLight *lights;
    const DevBvhNodeQ *bvh_nodes;
    const rt_f4 *bvh_recs;
    const uint32_t *tile_order;        // scan modes: launch order of the pixel groups (expensive first), null = row-major
    // Mixed launches (tile_order != null and mixed != 0): of the ordered pixel groups, the first head_groups are traced slice by slice
    // (their waves are long against the launch), those up to whole_end in one piece, the rest slice by slice again (the end of the launch
    // is made of short pieces); n_slices / partial serve the sliced ones. mixed == 0: every group the same way (n_slices).
    uint32_t mixed, head_groups, whole_end, n_groups;
    uint32_t *tile_cost_out;           // scan modes: per pixel group, the run time of its longest wave (100 MHz ticks; atomic max); null = not measured
    const struct RtTilePre *tile_pre;  // MODE_BVH: per wave tile of this launch, the primary-visibility list (rt_prepass.h); null = none
    // outputs
    float *accum;        // context-owned running sum, owned_rows*width*3
    float *accum_out;    // optional caller copy (may be null)
    uint32_t *rgba8;     // optional, owned_rows*width (R | G<<8 | B<<16 | 255<<24)
    unsigned long long *stats;  // RT_STAT_* counters, RT_STAT_SHARDS shards of RT_STAT_N
    // sample slicing: blockIdx.y = slice of slice_spp samples; partial sums [n_slices][owned_rows*width][3] when n_slices > 1
    uint32_t slice_spp, n_slices;
    float *partial;
    // RTHIP_F_ADAPTIVE (lib.rs:276-313)
    uint32_t *pix_state;          // per owned pixel: RT_PIX_* word (null: not adaptive)
    const uint32_t *worklist;     // passes after the first: owned-pixel indices (lrow*width + x) still to trace (null: every pixel)
    const uint32_t *worklist_count;
};

// per-pixel convergence state, the PixelCache fields of lib.rs:67-73 that the rule of lib.rs:300-311 needs
#define RT_PIX_RGB_MASK 0x00ffffffu  /* last_color */
#define RT_PIX_COUNT_SHIFT 24        /* same_color_count (<= 4) */
#define RT_PIX_COUNT_MASK 0xfu
#define RT_PIX_FINAL 0x40000000u     /* status == Final */
#define RT_PIX_HAS_LAST 0x80000000u  /* last_color.is_some() */

enum {
    RT_STAT_PATHS = 0, RT_STAT_RAYS = 1, RT_STAT_SPHERE = 2, RT_STAT_PLANE = 3, RT_STAT_TRI = 4, RT_STAT_NODE = 5,
    RT_STAT_HIT0 = 6 /* +kind, 4 slots */, RT_STAT_SKY = 10, RT_STAT_SHADOW = 11, RT_STAT_PIX_TRACED = 12, RT_STAT_PIX_FINAL = 13, RT_STAT_N = 14
};
#define RT_STAT_SHARDS 256 /* 129 600 waves at 4K bump each counter once: few shards serialise in L2 */
// rthip_render_device_noalloc: layout of the static scratch array (RTHIP_NOALLOC_SCRATCH_BYTES, include/rthip.h):
// [0, RT_NOALLOC_STATS_BYTES) the counter shards, then the partial sums of a sliced launch
#define RT_NOALLOC_STATS_BYTES 32768u /* >= RT_STAT_SHARDS * RT_STAT_N * 8 = 28 672, a multiple of 256 */
