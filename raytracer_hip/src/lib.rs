//! raytracer_hip — GPU (MI355X / gfx950) drop-in for raytracer_core's per-pixel render loop.
//!
//! `raytracer_core` keeps every geometry/material field private and `Scene = Vec<Box<dyn Shape>>` is opaque
//!, so the scene is described ONCE as a POD `SceneDesc`; `to_core_scene()` rebuilds the reference's
//! own `Scene` from it through the public constructors only, and `HipRaytracer` renders the same description on
//! the GPU. `HipRaytracer` mirrors `Raytracer` (raytracer_core/src/lib.rs:236-316): `new`, `pub camera`,
//! `render` (= `generate_pixel` for every pixel, the loop of wasm/src/lib.rs:112-123), `get_shape`.
pub mod ffi;

use ffi::*;
use raytracer_core::camera::Camera;
use raytracer_core::materials::{dielectric::Dielectric, lambertian_diffuse::Lambertian, metal::Metal, texture::Texture};
use raytracer_core::shapes::sphere::Sphere;
use raytracer_core::{GenerationStatus, PixelColor, PixelPosition, PixelRenderer, Scene, Vector3};
use std::ffi::CStr;

#[derive(Debug)]
pub struct HipError { pub status: i32, pub message: String }
pub type Result<T> = std::result::Result<T, HipError>;

#[derive(Clone)]
pub enum MaterialDesc {
    Lambertian { albedo: [f64; 3] },
    Metal { albedo: [f64; 3], fuzziness: f64 },
    Dielectric { albedo: [f64; 3], refraction_idx: f64 },
    /// `path`: the image file, what `Texture::load_from_file(path, scale)` takes (texture.rs:21) — the CPU path loads it from
    /// there; `rgb8`: its decoded RGB8 texels, row 0 = top (what `image::open(path)` + `get_pixel().to_rgb()` yield,
    /// texture.rs:22,50) — the GPU path uploads these. `MaterialDesc::texture_from_file` fills both from one file.
    Texture { path: std::path::PathBuf, width: u32, height: u32, rgb8: Vec<u8>, scale: f64 },
}

impl MaterialDesc {
    /// One file feeds both paths: decoded here with the same `image` crate the reference uses (texture.rs:22), so the
    /// device sees exactly the texels `Texture::scatter` reads.
    pub fn texture_from_file(path: &std::path::Path, scale: f64) -> image::ImageResult<MaterialDesc> {
        let img = image::open(path)?.to_rgb();
        let (width, height) = img.dimensions();
        Ok(MaterialDesc::Texture { path: path.to_path_buf(), width, height, rgb8: img.into_raw(), scale })
    }
}

#[derive(Clone, Default)]
pub struct SceneDesc {
    pub materials: Vec<MaterialDesc>,
    pub spheres: Vec<rthip_sphere>,
    pub planes: Vec<rthip_plane>,       // extension
    pub triangles: Vec<rthip_triangle>, // extension
    pub lights: Vec<rthip_light>,       // extension
}

impl SceneDesc {
    pub fn add_material(&mut self, m: MaterialDesc) -> u32 { self.materials.push(m); (self.materials.len() - 1) as u32 }
    pub fn add_sphere(&mut self, center: [f64; 3], radius: f64, material: u32) {
        self.spheres.push(rthip_sphere { center, radius, material, _pad: 0 });
    }
    /// The POD material / texture records of include/rthip.h (the texel pointers borrow from `self`).
    fn pod_materials(&self) -> (Vec<rthip_material>, Vec<rthip_texture>) {
        let mut textures = Vec::new();
        let materials = self.materials.iter().map(|m| match m {
            MaterialDesc::Lambertian { albedo } => rthip_material { kind: RTHIP_MAT_LAMBERTIAN, albedo: *albedo, ..Default::default() },
            MaterialDesc::Metal { albedo, fuzziness } => rthip_material { kind: RTHIP_MAT_METAL, albedo: *albedo, param: *fuzziness, ..Default::default() },
            MaterialDesc::Dielectric { albedo, refraction_idx } => rthip_material { kind: RTHIP_MAT_DIELECTRIC, albedo: *albedo, param: *refraction_idx, ..Default::default() },
            MaterialDesc::Texture { width, height, rgb8, scale, .. } => {
                textures.push(rthip_texture { width: *width, height: *height, rgb8: rgb8.as_ptr() });
                rthip_material { kind: RTHIP_MAT_TEXTURE, texture: (textures.len() - 1) as u32, param: *scale, ..Default::default() }
            }
        }).collect();
        (materials, textures)
    }
    fn pod(&self, materials: &[rthip_material], textures: &[rthip_texture]) -> rthip_scene_desc {
        rthip_scene_desc {
            spheres: self.spheres.as_ptr(), n_spheres: self.spheres.len() as u64,
            planes: self.planes.as_ptr(), n_planes: self.planes.len() as u64,
            triangles: self.triangles.as_ptr(), n_triangles: self.triangles.len() as u64,
            materials: materials.as_ptr(), n_materials: materials.len() as u64,
            textures: textures.as_ptr(), n_textures: textures.len() as u64,
            lights: self.lights.as_ptr(), n_lights: self.lights.len() as u64,
        }
    }

    /// The reference scene built through raytracer_core's public constructors (spheres only: the reference has no
    /// other shape). A texture material is loaded with `Texture::load_from_file(path, scale)` (texture.rs:21) like
    /// standalone/src/main.rs:85 does, and fails the same way (`unwrap` on the ImageResult) if the file is unreadable.
    pub fn to_core_scene(&self) -> Scene {
        let v = |a: [f64; 3]| Vector3::new(a[0], a[1], a[2]);
        self.spheres.iter().map(|s| {
            let m: Box<dyn raytracer_core::materials::material::Material> = match &self.materials[s.material as usize] {
                MaterialDesc::Lambertian { albedo } => Box::new(Lambertian::new(v(*albedo))),
                MaterialDesc::Metal { albedo, fuzziness } => Box::new(Metal::new(v(*albedo), *fuzziness)),
                MaterialDesc::Dielectric { albedo, refraction_idx } => Box::new(Dielectric::new(v(*albedo), *refraction_idx)),
                MaterialDesc::Texture { path, scale, .. } => Box::new(Texture::load_from_file(path, *scale).unwrap()),
            };
            Box::new(Sphere::new(v(s.center), s.radius, m)) as Box<dyn raytracer_core::Shape>
        }).collect()
    }
}

pub struct HipRaytracer {
    pub camera: Camera,
    ctx: *mut rthip_ctx,
    width: f64,
    height: f64,
    seed: u32,
    uploaded_camera: Option<rthip_camera>,
    status: Vec<u8>, // RTHIP_STATUS_* per pixel after the last adaptive pass
}
unsafe impl Send for HipRaytracer {} // one caller thread at a time, like the reference's single render thread

impl HipRaytracer {
    pub fn new(width: f64, height: f64, seed: u32, device: i32) -> Result<Self> {
        let mut ctx = std::ptr::null_mut();
        let st = unsafe { rthip_create(device, &mut ctx) };
        if st != 0 { return Err(err(std::ptr::null(), st)); }
        Ok(HipRaytracer { camera: Camera::new(-1.8, 1.0, 2.0), ctx, width, height, seed, uploaded_camera: None, status: Vec::new() })
    }

    pub fn upload(&mut self, scene: &SceneDesc) -> Result<()> {
        let (materials, textures) = scene.pod_materials();
        let desc = scene.pod(&materials, &textures);
        self.check(unsafe { rthip_upload_scene(self.ctx, &desc) })
    }

    fn sync_camera(&mut self) -> Result<()> {
        let cam = camera_pod(&self.camera);
        if self.uploaded_camera != Some(cam) {
            self.check(unsafe { rthip_set_camera(self.ctx, &cam) })?;
            self.uploaded_camera = Some(cam);
        }
        Ok(())
    }

    /// `generate_pixel` for every pixel with `samples` (batch semantics); `set_pixel` is called for each pixel,
    /// y-up like the reference, with status Final.
    pub fn render<S: PixelRenderer>(&mut self, samples: u64, max_depth: u32, accumulate: bool, renderer: &mut S) -> Result<rthip_stats> {
        self.sync_camera()?;
        let (w, h) = (self.width as u32, self.height as u32);
        let p = rthip_render_params { width: w, height: h, spp: samples as u32, max_depth, seed: self.seed,
                                      flags: if accumulate { RTHIP_F_ACCUMULATE } else { 0 }, ..Default::default() };
        let mut rgba = vec![0u8; (w as usize) * (h as usize) * 4];
        let mut stats = rthip_stats::default();
        self.check(unsafe { rthip_render(self.ctx, &p, rgba.as_mut_ptr(), std::ptr::null_mut(), &mut stats) })?;
        for y in 0..h as usize { for x in 0..w as usize {
            let i = (y * w as usize + x) * 4;
            renderer.set_pixel(PixelPosition { x, y }, PixelColor { r: rgba[i], g: rgba[i + 1], b: rgba[i + 2], status: GenerationStatus::Final });
        } }
        Ok(stats)
    }

    /// One progressive pass with the convergence rule of `generate_pixel` (lib.rs:276-313): pixels that are already
    /// `Final` are skipped on the device and get no `set_pixel`; the others are reported with their new status.
    pub fn render_adaptive_pass<S: PixelRenderer>(&mut self, samples: u64, max_depth: u32, first: bool, renderer: &mut S) -> Result<rthip_stats> {
        self.sync_camera()?;
        let (w, h) = (self.width as u32, self.height as u32);
        let npx = (w as usize) * (h as usize);
        if first || self.status.len() != npx { self.status = vec![RTHIP_STATUS_NOT_STARTED; npx]; }
        let before = self.status.clone();
        let p = rthip_render_params { width: w, height: h, spp: samples as u32, max_depth, seed: self.seed,
                                      flags: RTHIP_F_ADAPTIVE | if first { 0 } else { RTHIP_F_ACCUMULATE }, ..Default::default() };
        let mut rgba = vec![0u8; npx * 4];
        let mut stats = rthip_stats::default();
        self.check(unsafe { rthip_render(self.ctx, &p, rgba.as_mut_ptr(), std::ptr::null_mut(), &mut stats) })?;
        self.check(unsafe { rthip_pixel_status(self.ctx, self.status.as_mut_ptr(), std::ptr::null_mut(), self.status.len() as u64) })?;
        for i in 0..npx {
            if before[i] == RTHIP_STATUS_FINAL { continue; } // lib.rs:277-279
            let status = if self.status[i] == RTHIP_STATUS_FINAL { GenerationStatus::Final } else { GenerationStatus::Unstable };
            renderer.set_pixel(PixelPosition { x: i % w as usize, y: i / w as usize },
                               PixelColor { r: rgba[i * 4], g: rgba[i * 4 + 1], b: rgba[i * 4 + 2], status });
        }
        Ok(stats)
    }

    pub fn invalidate_pixels(&mut self) -> Result<()> { self.status.clear(); self.check(unsafe { rthip_reset_accum(self.ctx) }) }

    /// Raytracer::get_shape (lib.rs:262-267)
    pub fn get_shape(&mut self, x: f64, y: f64) -> Result<Option<usize>> {
        self.sync_camera()?;
        let mut s: i64 = -1;
        self.check(unsafe { rthip_pick(self.ctx, self.width as u32, self.height as u32, x, y, &mut s) })?;
        Ok(if s < 0 { None } else { Some(s as usize) })
    }

    fn check(&self, st: i32) -> Result<()> { if st == 0 { Ok(()) } else { Err(err(self.ctx, st)) } }
}

impl Drop for HipRaytracer { fn drop(&mut self) { unsafe { rthip_destroy(self.ctx) } } }

/// One render thread, several GPUs (rthip_group_*): `render` returns the whole frame, rendered as interleaved row bands on
/// every listed device and gathered to the first over RCCL. Same batch semantics as `HipRaytracer::render`.
pub struct HipGroupRaytracer {
    pub camera: Camera,
    grp: *mut rthip_group,
    width: f64,
    height: f64,
    seed: u32,
    uploaded_camera: Option<rthip_camera>,
}
unsafe impl Send for HipGroupRaytracer {}

impl HipGroupRaytracer {
    pub fn new(width: f64, height: f64, seed: u32, devices: &[i32]) -> Result<Self> {
        let mut grp = std::ptr::null_mut();
        let st = unsafe { rthip_group_create(devices.as_ptr(), devices.len() as i32, 0, &mut grp) };
        if st != 0 { return Err(group_err(std::ptr::null(), st)); }
        Ok(HipGroupRaytracer { camera: Camera::new(-1.8, 1.0, 2.0), grp, width, height, seed, uploaded_camera: None })
    }

    pub fn upload(&mut self, scene: &SceneDesc) -> Result<()> {
        let (materials, textures) = scene.pod_materials();
        let desc = scene.pod(&materials, &textures);
        self.check(unsafe { rthip_group_upload_scene(self.grp, &desc) })
    }

    pub fn render<S: PixelRenderer>(&mut self, samples: u64, max_depth: u32, accumulate: bool, renderer: &mut S) -> Result<rthip_stats> {
        let cam = camera_pod(&self.camera);
        if self.uploaded_camera != Some(cam) {
            self.check(unsafe { rthip_group_set_camera(self.grp, &cam) })?;
            self.uploaded_camera = Some(cam);
        }
        let (w, h) = (self.width as u32, self.height as u32);
        let p = rthip_render_params { width: w, height: h, spp: samples as u32, max_depth, seed: self.seed,
                                      flags: if accumulate { RTHIP_F_ACCUMULATE } else { 0 }, band_rows: 8, ..Default::default() };
        let mut rgba = vec![0u8; (w as usize) * (h as usize) * 4];
        let mut stats = rthip_stats::default();
        self.check(unsafe { rthip_group_render(self.grp, &p, rgba.as_mut_ptr(), std::ptr::null_mut(), &mut stats) })?;
        for y in 0..h as usize { for x in 0..w as usize {
            let i = (y * w as usize + x) * 4;
            renderer.set_pixel(PixelPosition { x, y }, PixelColor { r: rgba[i], g: rgba[i + 1], b: rgba[i + 2], status: GenerationStatus::Final });
        } }
        Ok(stats)
    }

    pub fn invalidate_pixels(&mut self) -> Result<()> { self.check(unsafe { rthip_group_reset_accum(self.grp) }) }
    fn check(&self, st: i32) -> Result<()> { if st == 0 { Ok(()) } else { Err(group_err(self.grp, st)) } }
}

impl Drop for HipGroupRaytracer { fn drop(&mut self) { unsafe { rthip_group_destroy(self.grp) } } }

fn group_err(grp: *const rthip_group, status: i32) -> HipError {
    let message = unsafe { CStr::from_ptr(rthip_group_last_error(grp)) }.to_string_lossy().into_owned();
    HipError { status, message }
}

/// Camera constants through the PUBLIC API only: origin, emit_ray_at(0,0), (1,0), (0,1).
fn camera_pod(camera: &Camera) -> rthip_camera {
    let d = |u, v| { let r = camera.emit_ray_at(u, v); let d = r.direction(); [d.x, d.y, d.z] };
    let (d00, d10, d01) = (d(0.0, 0.0), d(1.0, 0.0), d(0.0, 1.0));
    let o = camera.origin;
    rthip_camera {
        origin: [o.x, o.y, o.z], d00,
        horizontal: [d10[0] - d00[0], d10[1] - d00[1], d10[2] - d00[2]],
        vertical: [d01[0] - d00[0], d01[1] - d00[1], d01[2] - d00[2]],
    }
}

fn err(ctx: *const rthip_ctx, status: i32) -> HipError {
    let message = unsafe { CStr::from_ptr(rthip_last_error(ctx)) }.to_string_lossy().into_owned();
    HipError { status, message }
}
