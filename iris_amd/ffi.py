"""ctypes binding of libiris_hip.so (the C ABI declared in include/iris_hip.h).

There is no CPU fallback: if the HIP library is missing this module raises at
load time, and every entry point raises IrisError when the library reports a
failure.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# IRIS_HIP_LIB: alternative build of the same library (tuning experiments)
LIB_PATH = os.environ.get("IRIS_HIP_LIB") or os.path.join(HERE, "lib", "libiris_hip.so")
# The opt-in relaxed-arithmetic build of the same sources (iris_device.h, IRIS_RELAXED): never loaded unless asked for.
LIB_RELAXED_PATH = os.path.join(HERE, "lib", "libiris_hip_relaxed.so")
# The debug-invariants build (-DIRIS_DEBUG_INVARIANTS=1): counts where the reference's debug build would have panicked.
LIB_DEBUG_PATH = os.path.join(HERE, "lib", "libiris_hip_debug.so")

IRIS_OK = 0
TRACE_MAX_DEPTH = 64

# every symbol include/iris_hip.h declares
EXPORTS = (
    "iris_ctx_create", "iris_ctx_destroy", "iris_last_error", "iris_ctx_set_option", "iris_scene_upload",
    "iris_camera_new", "iris_roughness_to_alpha", "iris_render", "iris_render_device", "iris_render_device_noalloc",
    "iris_noalloc_workspace_bytes", "iris_film_to_rgb",
    "iris_get_stats", "iris_get_invariant_counts", "iris_allreduce_films", "iris_allreduce_shutdown", "iris_render_multi", "iris_sample_range", "iris_upsample_table_load", "iris_upsample_table_free", "iris_upsample_table_resolution",
    "iris_upsample_get_spectrum", "iris_upsample_get_spectrum_hdr", "iris_upsample_table_upload", "iris_upsample_get_spectrum_hdr_batch", "iris_envmap_tables", "iris_oct_to_dir", "iris_dir_to_oct",
    "iris_envmap_from_latlong", "iris_envmap_texels_from_rgb", "iris_hdr_read", "iris_dbg_sampler", "iris_dbg_raygen", "iris_dbg_intersect", "iris_dbg_trace",
)


class IrisError(RuntimeError):
    pass


class Sphere(C.Structure):
    _fields_ = [("cx", C.c_float), ("cy", C.c_float), ("cz", C.c_float), ("r", C.c_float)]


class Spectrum(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("c0", C.c_float), ("c1", C.c_float), ("c2", C.c_float), ("hdr", C.c_float)]


class Bsdf(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("spec_a", C.c_uint32), ("spec_b", C.c_uint32), ("p0", C.c_float), ("p1", C.c_float)]


class Light(C.Structure):
    _fields_ = [("spectrum", C.c_uint32), ("prim", C.c_uint32)]


class EnvmapDesc(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("texels", C.POINTER(C.c_float)), ("importance", C.POINTER(C.c_float))]


class SceneDesc(C.Structure):
    _fields_ = [("n_prims", C.c_uint32), ("spheres", C.POINTER(Sphere)), ("prim_material", C.POINTER(C.c_int32)),
                ("prim_light", C.POINTER(C.c_int32)), ("n_bsdfs", C.c_uint32), ("bsdfs", C.POINTER(Bsdf)),
                ("n_spectra", C.c_uint32), ("spectra", C.POINTER(Spectrum)), ("n_lights", C.c_uint32),
                ("lights", C.POINTER(Light)), ("env_map", C.POINTER(EnvmapDesc))]


class Camera(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("clip_to_world", C.c_float * 16)]


class RenderParams(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("width", "height", "spp_total", "spp_begin", "spp_count", "seed",
                                          "max_depth", "min_depth", "integrator")]


class Stats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("bounces", C.c_uint64), ("rays", C.c_uint64), ("sphere_tests", C.c_uint64),
                ("kernel_ms", C.c_double), ("resolve_ms", C.c_double), ("launches", C.c_uint32), ("chunks", C.c_uint32),
                ("kernel", C.c_uint32), ("bvh_depth", C.c_uint32), ("sky_samples", C.c_uint64)]


KERNEL_NAMES = ("k_wavefront", "k_stage_gen_extend+k_stage_shade", "k_regwave", "k_wfq_shade+k_wfq_trace")


class Trace(C.Structure):
    _fields_ = [("hero", C.c_float * 4), ("ray_o", C.c_float * 3), ("ray_d", C.c_float * 3),
                ("radiance", C.c_float * 4), ("xyz", C.c_float * 3), ("n_hits", C.c_uint32),
                ("hit_prim", C.c_int32 * TRACE_MAX_DEPTH), ("hit_t", C.c_float * TRACE_MAX_DEPTH),
                ("draws", C.c_uint32)]


_libs = {}


def load(relaxed=False, debug=False):
    """Load libiris_hip.so (relaxed: libiris_hip_relaxed.so, the opt-in build whose film is not the reference's bit for bit);
    raises IrisError if it has not been built (python -m iris_amd.build)."""
    path = LIB_DEBUG_PATH if debug else (LIB_RELAXED_PATH if relaxed else LIB_PATH)
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise IrisError("%s is missing at %s: build it with `python -m iris_amd.build`; "
                        "there is no CPU fallback" % (os.path.basename(path), path))
    L = C.CDLL(path)
    vp, fp, u32 = C.c_void_p, C.POINTER(C.c_float), C.c_uint32
    u32p, i32p = C.POINTER(C.c_uint32), C.POINTER(C.c_int32)
    sig = {
        "iris_ctx_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
        "iris_ctx_destroy": (None, [vp]),
        "iris_last_error": (C.c_char_p, [vp]),
        "iris_ctx_set_option": (C.c_int, [vp, C.c_char_p, C.c_int64]),
        "iris_scene_upload": (C.c_int, [vp, C.POINTER(SceneDesc)]),
        "iris_camera_new": (C.c_int, [fp, C.c_float, C.POINTER(Camera)]),
        "iris_roughness_to_alpha": (C.c_float, [C.c_float]),
        "iris_render": (C.c_int, [vp, C.POINTER(Camera), C.POINTER(RenderParams), fp]),
        "iris_render_device": (C.c_int, [vp, C.POINTER(Camera), C.POINTER(RenderParams), vp, vp]),
        "iris_render_device_noalloc": (C.c_int, [vp, C.POINTER(Camera), C.POINTER(RenderParams), vp, vp]),
        "iris_noalloc_workspace_bytes": (C.c_size_t, []),
        "iris_film_to_rgb": (C.c_int, [fp, C.c_size_t, fp]),
        "iris_get_stats": (C.c_int, [vp, C.POINTER(Stats)]),
        "iris_get_invariant_counts": (C.c_int, [vp, C.POINTER(C.c_uint64)]),
        "iris_allreduce_films": (C.c_int, [C.POINTER(vp), C.c_int, C.POINTER(vp), C.c_size_t]),
        "iris_allreduce_shutdown": (None, []),
        "iris_render_multi": (C.c_int, [C.POINTER(vp), C.c_int, C.POINTER(Camera), C.POINTER(RenderParams), fp]),
        "iris_sample_range": (C.c_int, [u32, u32, u32, u32p, u32p]),
        "iris_upsample_table_load": (C.c_int, [C.c_char_p, C.POINTER(vp)]),
        "iris_upsample_table_free": (None, [vp]),
        "iris_upsample_table_resolution": (C.c_uint32, [vp]),
        "iris_upsample_get_spectrum": (C.c_int, [vp, fp, C.POINTER(Spectrum)]),
        "iris_upsample_get_spectrum_hdr": (C.c_int, [vp, fp, C.POINTER(Spectrum)]),
        "iris_upsample_table_upload": (C.c_int, [vp, vp]),
        "iris_upsample_get_spectrum_hdr_batch": (C.c_int, [vp, vp, C.c_size_t, C.c_float, vp, vp, C.c_int, vp]),
        "iris_envmap_tables": (C.c_int, [C.POINTER(EnvmapDesc), fp, fp, fp, fp]),
        "iris_oct_to_dir": (None, [C.c_float, C.c_float, fp]),
        "iris_dir_to_oct": (None, [fp, fp]),
        "iris_envmap_from_latlong": (C.c_int, [fp, u32, u32, u32, u32, fp]),
        "iris_envmap_texels_from_rgb": (C.c_int, [vp, fp, C.c_size_t, C.c_float, fp, fp]),
        "iris_hdr_read": (C.c_int, [C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), fp, C.c_size_t]),
        "iris_dbg_sampler": (C.c_int, [vp, u32, u32p, u32, u32, fp]),
        "iris_dbg_raygen": (C.c_int, [vp, C.POINTER(Camera), C.POINTER(RenderParams), u32, u32p, fp, fp, fp]),
        "iris_dbg_intersect": (C.c_int, [vp, u32, fp, fp, i32p, fp]),
        "iris_dbg_trace": (C.c_int, [vp, C.POINTER(Camera), C.POINTER(RenderParams), u32, u32p, C.POINTER(Trace)]),
    }
    assert set(sig) == set(EXPORTS)
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _libs[path] = L
    return L
