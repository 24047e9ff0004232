//! #[repr(C)] mirrors of include/rthip.h and the extern "C" entry points. Keep in sync with the header
//! (RTHIP_ABI_VERSION = 4); tests/test_abi.py pins the struct sizes the same way.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

pub const RTHIP_MAT_LAMBERTIAN: u32 = 0;
pub const RTHIP_MAT_METAL: u32 = 1;
pub const RTHIP_MAT_DIELECTRIC: u32 = 2;
pub const RTHIP_MAT_TEXTURE: u32 = 3;

pub const RTHIP_F_ACCUMULATE: u32 = 1 << 0;
pub const RTHIP_F_POINT_LIGHTS: u32 = 1 << 1;
pub const RTHIP_F_COUNT_STATS: u32 = 1 << 2;
pub const RTHIP_F_NO_BVH: u32 = 1 << 3;
pub const RTHIP_F_ADAPTIVE: u32 = 1 << 4;
pub const RTHIP_F_TILE_LISTS: u32 = 1 << 5;
pub const RTHIP_F_COUNT_EXECUTED: u32 = 1 << 6;
pub const RTHIP_STATUS_NOT_STARTED: u8 = 0;
pub const RTHIP_STATUS_UNSTABLE: u8 = 2;
pub const RTHIP_STATUS_FINAL: u8 = 3;
pub const RTHIP_GROUP_F_PEER_COPY: u32 = 1 << 0;
pub const RTHIP_GROUP_F_FORCE_RCCL: u32 = 1 << 1;
pub const RTHIP_NOALLOC_SCRATCH_BYTES: usize = 64 << 20;

#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct rthip_material { pub kind: u32, pub texture: u32, pub albedo: [f64; 3], pub param: f64, pub shininess: f64 }
#[repr(C)] #[derive(Clone, Copy)]
pub struct rthip_texture { pub width: u32, pub height: u32, pub rgb8: *const u8 }
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct rthip_sphere { pub center: [f64; 3], pub radius: f64, pub material: u32, pub _pad: u32 }
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct rthip_plane { pub point: [f64; 3], pub normal: [f64; 3], pub material: u32, pub _pad: u32 }
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct rthip_triangle { pub v0: [f64; 3], pub v1: [f64; 3], pub v2: [f64; 3], pub material: u32, pub _pad: u32 }
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct rthip_light { pub position: [f64; 3], pub intensity: [f64; 3] }
#[repr(C)]
pub struct rthip_scene_desc {
    pub spheres: *const rthip_sphere, pub n_spheres: u64,
    pub planes: *const rthip_plane, pub n_planes: u64,
    pub triangles: *const rthip_triangle, pub n_triangles: u64,
    pub materials: *const rthip_material, pub n_materials: u64,
    pub textures: *const rthip_texture, pub n_textures: u64,
    pub lights: *const rthip_light, pub n_lights: u64,
}
#[repr(C)] #[derive(Clone, Copy, Default, PartialEq)]
pub struct rthip_camera { pub origin: [f64; 3], pub d00: [f64; 3], pub horizontal: [f64; 3], pub vertical: [f64; 3] }
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct rthip_render_params {
    pub width: u32, pub height: u32, pub spp: u32, pub max_depth: u32, pub seed: u32, pub flags: u32,
    pub band_rows: u32, pub band_first: u32, pub band_stride: u32, pub reserved: [u32; 7],
}
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct rthip_stats {
    pub paths: u64, pub rays: u64, pub sphere_tests: u64, pub plane_tests: u64, pub tri_tests: u64, pub node_tests: u64,
    pub hits: [u64; 4], pub sky_evals: u64, pub shadow_rays: u64, pub rows: u64, pub nb_samples: u64,
    pub kernel_ms: f64, pub d2h_ms: f64,
    pub pixels_traced: u64, pub pixels_final: u64,
    pub gather_ms: f64,
}
pub enum rthip_ctx {}
pub enum rthip_group {}

extern "C" {
    pub fn rthip_abi_version() -> c_int;
    pub fn rthip_device_count(out_count: *mut c_int) -> c_int;
    pub fn rthip_create(device: c_int, out_ctx: *mut *mut rthip_ctx) -> c_int;
    pub fn rthip_upload_scene(ctx: *mut rthip_ctx, scene: *const rthip_scene_desc) -> c_int;
    pub fn rthip_set_camera(ctx: *mut rthip_ctx, camera: *const rthip_camera) -> c_int;
    pub fn rthip_camera_lookat(x: f64, y: f64, z: f64, lookat: *const f64, out: *mut rthip_camera) -> c_int;
    pub fn rthip_owned_rows(params: *const rthip_render_params, out_rows: *mut u32) -> c_int;
    pub fn rthip_render(ctx: *mut rthip_ctx, params: *const rthip_render_params, rgba8_out: *mut u8, accum_out: *mut f32,
                        stats: *mut rthip_stats) -> c_int;
    pub fn rthip_render_device(ctx: *mut rthip_ctx, params: *const rthip_render_params, d_rgba8: *mut c_void, d_accum: *mut c_void,
                               hip_stream: *mut c_void, stats: *mut rthip_stats) -> c_int;
    pub fn rthip_reset_accum(ctx: *mut rthip_ctx) -> c_int;
    pub fn rthip_render_device_noalloc(ctx: *mut rthip_ctx, params: *const rthip_render_params, d_rgba8: *mut c_void, d_accum: *mut c_void,
                                       hip_stream: *mut c_void, stats: *mut rthip_stats) -> c_int;
    pub fn rthip_pixel_status(ctx: *mut rthip_ctx, status_out: *mut u8, same_color_count_out: *mut u8, capacity_pixels: u64) -> c_int;
    pub fn rthip_accum_layout(ctx: *mut rthip_ctx, out_rows: *mut u32, out_width: *mut u32, out_nb_samples: *mut u64) -> c_int;
    pub fn rthip_group_create(devices: *const c_int, ndev: c_int, flags: u32, out_group: *mut *mut rthip_group) -> c_int;
    pub fn rthip_group_upload_scene(group: *mut rthip_group, scene: *const rthip_scene_desc) -> c_int;
    pub fn rthip_group_set_camera(group: *mut rthip_group, camera: *const rthip_camera) -> c_int;
    pub fn rthip_group_render(group: *mut rthip_group, params: *const rthip_render_params, rgba8_out: *mut u8, accum_out: *mut f32,
                              stats: *mut rthip_stats) -> c_int;
    pub fn rthip_group_reset_accum(group: *mut rthip_group) -> c_int;
    pub fn rthip_group_pixel_status(group: *mut rthip_group, status_out: *mut u8, same_color_count_out: *mut u8, capacity_pixels: u64) -> c_int;
    pub fn rthip_group_size(group: *const rthip_group) -> c_int;
    pub fn rthip_group_context(group: *mut rthip_group, index: c_int) -> *mut rthip_ctx;
    pub fn rthip_group_last_error(group: *const rthip_group) -> *const c_char;
    pub fn rthip_group_destroy(group: *mut rthip_group);
    pub fn rthip_band_row_owner(band_rows: u32, ndev: u32, y: u32, out_rank: *mut u32, out_local_row: *mut u32) -> c_int;
    pub fn rthip_render_async(ctx: *mut rthip_ctx, params: *const rthip_render_params, rgba8_out: *mut u8, slot: u32) -> c_int;
    pub fn rthip_wait(ctx: *mut rthip_ctx, slot: u32) -> c_int;
    pub fn rthip_host_alloc(bytes: usize, out_ptr: *mut *mut c_void) -> c_int;
    pub fn rthip_host_free(ptr: *mut c_void) -> c_int;
    pub fn rthip_pick(ctx: *mut rthip_ctx, width: u32, height: u32, x: f64, y: f64, out_shape: *mut i64) -> c_int;
    pub fn rthip_algorithmic_flops(stats: *const rthip_stats) -> f64;
    pub fn rthip_last_error(ctx: *const rthip_ctx) -> *const c_char;
    pub fn rthip_status_string(status: c_int) -> *const c_char;
    pub fn rthip_destroy(ctx: *mut rthip_ctx);
}
