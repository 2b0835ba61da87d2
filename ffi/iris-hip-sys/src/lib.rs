//! Declarations only: one `pub fn` per symbol of include/iris_hip.h, same order of arguments.
//! (tests/test_boundary.py checks names and arity against the header; no Rust toolchain exists in the build image,
//! so this crate is shipped as source and compiled by the maintainer who wires it into iris.)
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

pub const IRIS_OK: c_int = 0;
pub const IRIS_E_ARG: c_int = -1;
pub const IRIS_E_HIP: c_int = -2;
pub const IRIS_E_STATE: c_int = -3;
pub const IRIS_E_LIMIT: c_int = -4;
pub const IRIS_E_RCCL: c_int = -5;

/// enum Bsdf, bsdf/mod.rs:49-55
pub const IRIS_BSDF_LAMBERTIAN: u32 = 0;
pub const IRIS_BSDF_MICROFACET: u32 = 1;
pub const IRIS_BSDF_SPECULAR: u32 = 2;
pub const IRIS_BSDF_FRESNEL: u32 = 3;
pub const IRIS_BSDF_NULL: u32 = 4;
/// enum Spectrum, spectrum/mod.rs:25-28 (+ UpsampledHdrSpectrum, upsample.rs:27-42)
pub const IRIS_SPEC_UPSAMPLED: u32 = 0;
pub const IRIS_SPEC_CONSTANT: u32 = 1;
pub const IRIS_SPEC_UPSAMPLED_HDR: u32 = 2;
/// integrator/mod.rs:8-11
pub const IRIS_INT_HWSS_NAIVE: u32 = 0;
pub const IRIS_INT_HWSS_SLOW: u32 = 1;
pub const IRIS_INT_SWSS_NAIVE: u32 = 2;
pub const IRIS_INT_SWSS_SLOW: u32 = 3;

pub const IRIS_TRACE_MAX_DEPTH: usize = 64;

#[repr(C)] pub struct iris_ctx { _private: [u8; 0] }
#[repr(C)] pub struct iris_upsample_table { _private: [u8; 0] }
#[repr(C)] #[derive(Clone, Copy, Debug, Default)] pub struct iris_sphere   { pub cx: f32, pub cy: f32, pub cz: f32, pub r: f32 }
#[repr(C)] #[derive(Clone, Copy, Debug, Default)] pub struct iris_spectrum { pub kind: u32, pub c0: f32, pub c1: f32, pub c2: f32, pub hdr: f32 }
#[repr(C)] #[derive(Clone, Copy, Debug, Default)] pub struct iris_bsdf     { pub kind: u32, pub spec_a: u32, pub spec_b: u32, pub p0: f32, pub p1: f32 }
#[repr(C)] #[derive(Clone, Copy, Debug, Default)] pub struct iris_light    { pub spectrum: u32, pub prim: u32 }
#[repr(C)] pub struct iris_envmap_desc { pub width: u32, pub height: u32, pub texels: *const f32, pub importance: *const f32 }
#[repr(C)] pub struct iris_scene_desc {
    pub n_prims: u32, pub spheres: *const iris_sphere, pub prim_material: *const i32, pub prim_light: *const i32,
    pub n_bsdfs: u32, pub bsdfs: *const iris_bsdf, pub n_spectra: u32, pub spectra: *const iris_spectrum,
    pub n_lights: u32, pub lights: *const iris_light,
    /// null: none (the reference never fills Scene._env_map, scene.rs:27)
    pub env_map: *const iris_envmap_desc,
}
#[repr(C)] #[derive(Clone, Copy, Debug)] pub struct iris_camera { pub position: [f32; 3], pub clip_to_world: [f32; 16] }
#[repr(C)] #[derive(Clone, Copy, Debug, Default)] pub struct iris_render_params {
    pub width: u32, pub height: u32, pub spp_total: u32, pub spp_begin: u32, pub spp_count: u32,
    pub seed: u32, pub max_depth: u32, pub min_depth: u32, pub integrator: u32,
}
#[repr(C)] #[derive(Clone, Copy, Debug, Default)] pub struct iris_stats {
    pub samples: u64, pub bounces: u64, pub rays: u64, pub sphere_tests: u64,
    pub kernel_ms: f64, pub resolve_ms: f64, pub launches: u32, pub chunks: u32, pub kernel: u32, pub bvh_depth: u32,
    pub sky_samples: u64,
}
#[repr(C)] #[derive(Clone, Copy)] pub struct iris_trace {
    pub hero: [f32; 4], pub ray_o: [f32; 3], pub ray_d: [f32; 3], pub radiance: [f32; 4], pub xyz: [f32; 3],
    pub n_hits: u32, pub hit_prim: [i32; IRIS_TRACE_MAX_DEPTH], pub hit_t: [f32; IRIS_TRACE_MAX_DEPTH], pub draws: u32,
}

extern "C" {
    // ---- context
    pub fn iris_ctx_create(device: c_int, out: *mut *mut iris_ctx) -> c_int;
    pub fn iris_ctx_destroy(ctx: *mut iris_ctx);
    pub fn iris_last_error(ctx: *const iris_ctx) -> *const c_char;
    pub fn iris_ctx_set_option(ctx: *mut iris_ctx, key: *const c_char, value: i64) -> c_int;
    // ---- scene / camera
    pub fn iris_scene_upload(ctx: *mut iris_ctx, scene: *const iris_scene_desc) -> c_int;
    pub fn iris_camera_new(position: *const f32, aspect_ratio: f32, out: *mut iris_camera) -> c_int;
    pub fn iris_roughness_to_alpha(roughness: f32) -> f32;
    pub fn iris_upsample_table_load(path: *const c_char, out: *mut *mut iris_upsample_table) -> c_int;
    pub fn iris_upsample_table_free(table: *mut iris_upsample_table);
    pub fn iris_upsample_table_resolution(table: *const iris_upsample_table) -> u32;
    pub fn iris_upsample_get_spectrum(table: *const iris_upsample_table, rgb: *const f32, out: *mut iris_spectrum) -> c_int;
    pub fn iris_upsample_get_spectrum_hdr(table: *const iris_upsample_table, rgb: *const f32, out: *mut iris_spectrum) -> c_int;
    pub fn iris_upsample_table_upload(ctx: *mut iris_ctx, table: *const iris_upsample_table) -> c_int;
    pub fn iris_upsample_get_spectrum_hdr_batch(ctx: *mut iris_ctx, rgb: *const f32, n: usize, max_hdr: f32, texels: *mut f32, importance: *mut f32, on_device: c_int, stream: *mut c_void) -> c_int;
    pub fn iris_envmap_tables(map: *const iris_envmap_desc, func: *mut f32, cond_cdf: *mut f32, marg_cdf: *mut f32, integral: *mut f32) -> c_int;
    pub fn iris_oct_to_dir(u: f32, v: f32, dir: *mut f32);
    pub fn iris_dir_to_oct(dir: *const f32, uv: *mut f32);
    pub fn iris_envmap_from_latlong(rgb: *const f32, width: u32, height: u32, out_width: u32, out_height: u32, out_rgb: *mut f32) -> c_int;
    pub fn iris_envmap_texels_from_rgb(table: *const iris_upsample_table, rgb: *const f32, n_texels: usize, max_hdr: f32, texels: *mut f32, importance: *mut f32) -> c_int;
    pub fn iris_hdr_read(path: *const c_char, width: *mut u32, height: *mut u32, rgb: *mut f32, rgb_floats: usize) -> c_int;
    // ---- the hot path: replaces TileData::render -> get_pixel_color -> Integrator::radiance (tile.rs:75-166)
    pub fn iris_render(ctx: *mut iris_ctx, camera: *const iris_camera, params: *const iris_render_params, out_xyz: *mut f32) -> c_int;
    pub fn iris_render_device(ctx: *mut iris_ctx, camera: *const iris_camera, params: *const iris_render_params, d_out_xyz: *mut f32, stream: *mut c_void) -> c_int;
    pub fn iris_render_device_noalloc(ctx: *mut iris_ctx, camera: *const iris_camera, params: *const iris_render_params, d_out_xyz: *mut f32, stream: *mut c_void) -> c_int;
    pub fn iris_noalloc_workspace_bytes() -> usize;
    // ---- several GPUs in one process: thread per GPU + spp split + RCCL all-reduce of the device films
    pub fn iris_allreduce_films(ctxs: *const *mut iris_ctx, n: c_int, d_films: *const *mut f32, n_floats: usize) -> c_int;
    pub fn iris_allreduce_shutdown();
    pub fn iris_render_multi(ctxs: *const *mut iris_ctx, n: c_int, camera: *const iris_camera, params: *const iris_render_params, out_xyz: *mut f32) -> c_int;
    pub fn iris_sample_range(spp_count: u32, part: u32, n_parts: u32, begin: *mut u32, count: *mut u32) -> c_int;
    pub fn iris_film_to_rgb(xyz: *const f32, n_pixels: usize, rgb: *mut f32) -> c_int;
    pub fn iris_get_stats(ctx: *const iris_ctx, out: *mut iris_stats) -> c_int;
    pub fn iris_get_invariant_counts(ctx: *const iris_ctx, out: *mut u64) -> c_int;
    // ---- stage-level entry points for parity tests
    pub fn iris_dbg_sampler(ctx: *mut iris_ctx, n: u32, xys: *const u32, seed: u32, ndraws: u32, out: *mut f32) -> c_int;
    pub fn iris_dbg_raygen(ctx: *mut iris_ctx, camera: *const iris_camera, params: *const iris_render_params, n: u32, xys: *const u32, hero: *mut f32, ray_o: *mut f32, ray_d: *mut f32) -> c_int;
    pub fn iris_dbg_intersect(ctx: *mut iris_ctx, n: u32, ray_o: *const f32, ray_d: *const f32, prim: *mut i32, t: *mut f32) -> c_int;
    pub fn iris_dbg_trace(ctx: *mut iris_ctx, camera: *const iris_camera, params: *const iris_render_params, n: u32, xys: *const u32, out: *mut iris_trace) -> c_int;
}
