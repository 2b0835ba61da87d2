//! src/integrator/hip_wavefront.rs -- the type that stands next to HwssNaive & co (integrator/mod.rs:8-15).
//!
//! `Integrator::radiance` is called once per camera sample (tile.rs:159-162); a per-ray call cannot feed a GPU, so this
//! type replaces the loop around it: `render_film` does what TileData::render -> get_pixel_color -> radiance do
//! (tile.rs:75-166) for a whole film and a range of sample indices.  The enum surface of the reference is kept: `Bsdf`,
//! `Spectrum` and the integrator choice travel as tags in the flattened scene.
//!
//! Not compiled in the build image (no Rust toolchain there); the same ABI is exercised by iris_amd/host/iris_main.cpp
//! and iris_amd/ffi.py.  Needs the accessor patch in ffi/iris_patch/accessors.patch.
use iris_hip_sys as sys;

use crate::{
    bsdf::Bsdf,
    color::Xyz,
    scene::Scene,
    shape::Geometry,
    spectrum::Spectrum,
    Render,
};

/// `type CurrentIntegrator` of main.rs:31 as a run-time tag.
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
#[repr(u32)]
pub enum DeviceIntegrator {
    HwssNaive = sys::IRIS_INT_HWSS_NAIVE,
    HwssSlow = sys::IRIS_INT_HWSS_SLOW,
    SwssNaive = sys::IRIS_INT_SWSS_NAIVE,
    SwssSlow = sys::IRIS_INT_SWSS_SLOW,
}

pub struct HipWavefront {
    ctx: *mut sys::iris_ctx,
    pub integrator: DeviceIntegrator,
    pub max_depth: u32, // hwss_naive.rs:16
    pub min_depth: u32, // hwss_naive.rs:17
}
// one context per GPU; calls on it are serialised by the owner
unsafe impl Send for HipWavefront {}

fn flat_spectrum(s: &Spectrum) -> sys::iris_spectrum {
    match s {
        // spectrum/mod.rs:25-28
        Spectrum::UpsampledSpectrum(u) => {
            let c = u.coefficients();
            sys::iris_spectrum { kind: sys::IRIS_SPEC_UPSAMPLED, c0: c[0], c1: c[1], c2: c[2], hdr: 1.0 }
        }
        Spectrum::ConstantSpectrum(c) => sys::iris_spectrum { kind: sys::IRIS_SPEC_CONSTANT, c0: c.value(), c1: 0.0, c2: 0.0, hdr: 1.0 },
    }
}

fn flat_bsdf(b: &Bsdf, push: &mut dyn FnMut(&Spectrum) -> u32) -> sys::iris_bsdf {
    match b {
        // bsdf/mod.rs:49-55
        Bsdf::LambertianBsdf(l) => sys::iris_bsdf { kind: sys::IRIS_BSDF_LAMBERTIAN, spec_a: push(l.albedo()), spec_b: 0, p0: 0.0, p1: 0.0 },
        Bsdf::MicrofacetBsdf(m) => {
            let (ax, ay) = m.alpha(); // already mapped by ggx::roughness_to_alpha in MicrofacetBsdf::new (microfacet.rs:21-29)
            sys::iris_bsdf { kind: sys::IRIS_BSDF_MICROFACET, spec_a: push(m.reflectance()), spec_b: 0, p0: ax, p1: ay }
        }
        Bsdf::SpecularBsdf(s) => sys::iris_bsdf { kind: sys::IRIS_BSDF_SPECULAR, spec_a: push(s.reflected_color()), spec_b: 0, p0: 0.0, p1: 0.0 },
        Bsdf::FresnelBsdf(f) => sys::iris_bsdf {
            kind: sys::IRIS_BSDF_FRESNEL,
            spec_a: push(f.reflected_color()),
            spec_b: push(f.transmitted_color()),
            p0: f.base_ior(),
            p1: f.dispersion(),
        },
        Bsdf::NullBsdf(_) => sys::iris_bsdf { kind: sys::IRIS_BSDF_NULL, spec_a: 0, spec_b: 0, p0: 0.0, p1: 0.0 },
    }
}

impl HipWavefront {
    pub fn new(device: i32, scene: &Scene, integrator: DeviceIntegrator) -> Result<Self, String> {
        let mut ctx = std::ptr::null_mut();
        if unsafe { sys::iris_ctx_create(device, &mut ctx) } != sys::IRIS_OK {
            return Err(format!("iris_ctx_create({}) failed: no HIP device?", device));
        }
        let me = HipWavefront { ctx, integrator, max_depth: 15, min_depth: 2 };
        // flatten scene.rs:22-28
        let mut spectra: Vec<sys::iris_spectrum> = Vec::new();
        let mut push = |s: &Spectrum| -> u32 {
            spectra.push(flat_spectrum(s));
            spectra.len() as u32 - 1
        };
        let bsdfs: Vec<sys::iris_bsdf> = scene.materials.iter().map(|m| flat_bsdf(&m.data, &mut push)).collect();
        let lights: Vec<sys::iris_light> =
            scene.lights.iter().map(|l| sys::iris_light { spectrum: push(&l.data), prim: l.prim_index as u32 }).collect();
        let spheres: Vec<sys::iris_sphere> = scene
            .primitives
            .iter()
            .map(|p| match &p.geometry {
                Geometry::Sphere(s) => {
                    let (c, r) = s.center_radius(); // sphere.rs:7-11
                    sys::iris_sphere { cx: c.x, cy: c.y, cz: c.z, r }
                }
            })
            .collect();
        let mat: Vec<i32> = scene.primitives.iter().map(|p| p.material_index.map_or(-1, |i| i as i32)).collect();
        let lig: Vec<i32> = scene.primitives.iter().map(|p| p.light_index.map_or(-1, |i| i as i32)).collect();
        let desc = sys::iris_scene_desc {
            n_prims: spheres.len() as u32,
            spheres: spheres.as_ptr(),
            prim_material: mat.as_ptr(),
            prim_light: lig.as_ptr(),
            n_bsdfs: bsdfs.len() as u32,
            bsdfs: bsdfs.as_ptr(),
            n_spectra: spectra.len() as u32,
            spectra: spectra.as_ptr(),
            n_lights: lights.len() as u32,
            lights: lights.as_ptr(),
            env_map: std::ptr::null(), // or &iris_envmap_desc built from Scene._env_map once something fills it (scene.rs:27)
        };
        me.check(unsafe { sys::iris_scene_upload(me.ctx, &desc) })?;
        Ok(me)
    }

    /// Replaces TileData::render + get_pixel_color + Integrator::radiance (tile.rs:75-166) for sample indices
    /// [spp_begin, spp_begin + spp_count) of every pixel; returns xyz_sum * (1 / render.spp) per pixel (tile.rs:165),
    /// indexed x + y * width like Render.buffer (main.rs:120-122).
    pub fn render_film(&self, render: &Render, spp_begin: usize, spp_count: usize) -> Result<Vec<Xyz>, String> {
        let cam = sys::iris_camera {
            position: [render.camera.position.x, render.camera.position.y, render.camera.position.z],
            clip_to_world: render.camera.clip_to_world.to_row_major(),
        };
        let p = sys::iris_render_params {
            width: render.width as u32,
            height: render.height as u32,
            spp_total: render.spp as u32,
            spp_begin: spp_begin as u32,
            spp_count: spp_count as u32,
            seed: crate::tile::SEED,
            max_depth: self.max_depth,
            min_depth: self.min_depth,
            integrator: self.integrator as u32,
        };
        let mut out = vec![0f32; render.width * render.height * 3];
        self.check(unsafe { sys::iris_render(self.ctx, &cam, &p, out.as_mut_ptr()) })?;
        Ok(out.chunks_exact(3).map(|c| Xyz::new(c[0], c[1], c[2])).collect())
    }

    /// Work counters and kernel time of the last render (rays, sphere tests, bounces: the inputs of the roofline model).
    pub fn stats(&self) -> Result<sys::iris_stats, String> {
        let mut st = sys::iris_stats::default();
        self.check(unsafe { sys::iris_get_stats(self.ctx, &mut st) })?;
        Ok(st)
    }

    fn check(&self, rc: i32) -> Result<(), String> {
        if rc == sys::IRIS_OK {
            return Ok(());
        }
        let msg = unsafe { std::ffi::CStr::from_ptr(sys::iris_last_error(self.ctx)) };
        Err(format!("libiris_hip error {}: {}", rc, msg.to_string_lossy()))
    }
}

impl Drop for HipWavefront {
    fn drop(&mut self) {
        unsafe { sys::iris_ctx_destroy(self.ctx) }
    }
}

/// What do_render's thread pool (main.rs:87-107) becomes: one call, then to_rgb_hdr + clamp per pixel (tile.rs:108-109).
pub fn render_on_gpu(render: &Render) -> Result<(), String> {
    let gpu = HipWavefront::new(0, &render.scene, DeviceIntegrator::HwssNaive)?;
    let film = gpu.render_film(render, 0, render.spp)?;
    let mut buffer = render.buffer.write().unwrap();
    for (px, xyz) in buffer.iter_mut().zip(film) {
        let rgb = xyz.to_rgb_hdr();
        *px = (rgb.0.max(0.0), rgb.1.max(0.0), rgb.2.max(0.0));
    }
    Ok(())
}
