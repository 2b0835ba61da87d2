"""Host-side driver above the C ABI: the part of main.rs / tile.rs that stays on the host.

`Camera` mirrors camera.rs, `Render` mirrors the `Render` struct of main.rs:33-41 and `HipWavefront`
stands where `CurrentIntegrator` does (main.rs:31,40): `render_film` replaces the whole
TileData::render -> get_pixel_color -> Integrator::radiance stack (tile.rs:75-166) for one film.
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import ffi
from .scene import HWSS_NAIVE, MAX_DEPTH, MIN_DEPTH, SEED, FlatScene, Scene


def _check(L, ctx, rc, what):
    if rc != ffi.IRIS_OK:
        msg = L.iris_last_error(ctx).decode() if ctx else ""
        raise ffi.IrisError("%s failed (%d): %s" % (what, rc, msg))


class Camera:
    """camera.rs:3-21."""

    def __init__(self, position, aspect_ratio):
        L = ffi.load()
        self.c = ffi.Camera()
        pos = (C.c_float * 3)(*[float(v) for v in position])
        rc = L.iris_camera_new(pos, C.c_float(aspect_ratio), C.byref(self.c))
        if rc != ffi.IRIS_OK:
            raise ffi.IrisError("iris_camera_new failed (%d)" % rc)

    @property
    def position(self):
        return np.array(self.c.position, np.float32)

    @property
    def clip_to_world(self):
        return np.array(self.c.clip_to_world, np.float32).reshape(4, 4)


@dataclass
class Render:
    """main.rs:33-41 (the film buffer is returned by render_film instead of being held here)."""
    width: int
    height: int
    spp: int
    scene: Scene
    camera: Camera = None
    integrator: int = HWSS_NAIVE
    seed: int = SEED
    max_depth: int = MAX_DEPTH
    min_depth: int = MIN_DEPTH

    def __post_init__(self):
        if self.camera is None:  # main.rs:51-54
            self.camera = Camera((0.0, 0.0, 0.0), np.float32(self.width) / np.float32(self.height))

    def params(self, spp_begin=0, spp_count=None):
        return ffi.RenderParams(self.width, self.height, self.spp, spp_begin, self.spp if spp_count is None else spp_count,
                                self.seed, self.max_depth, self.min_depth, self.integrator)


class HipWavefront:
    """One GPU's wavefront path tracer (an iris_ctx)."""

    def __init__(self, device=0, relaxed=False, debug=False, **options):
        """relaxed: the opt-in build whose radiometric arithmetic uses 1-ulp reciprocals/roots and contraction (faster; the film
        then meets the statistical image gate, not the reference's bits).  Default: the exact build."""
        self.L = ffi.load(relaxed=relaxed, debug=debug)  # debug: libiris_hip_debug.so, which counts the reference's debug invariants
        self.ctx = C.c_void_p()
        rc = self.L.iris_ctx_create(int(device), C.byref(self.ctx))
        if rc != ffi.IRIS_OK:
            self.ctx = None
            raise ffi.IrisError("iris_ctx_create(device=%d) failed (%d): is a HIP device present?" % (device, rc))
        self._scene_keep = None
        for k, v in options.items():
            self.set_option(k, v)

    def close(self):
        if getattr(self, "ctx", None):
            self.L.iris_ctx_destroy(self.ctx)
            self.ctx = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_option(self, key, value):
        _check(self.L, self.ctx, self.L.iris_ctx_set_option(self.ctx, key.encode(), int(value)), "iris_ctx_set_option")

    def upload(self, scene):
        flat = scene.flatten() if isinstance(scene, Scene) else scene
        assert isinstance(flat, FlatScene)
        desc, keep = flat.to_desc()
        _check(self.L, self.ctx, self.L.iris_scene_upload(self.ctx, C.byref(desc)), "iris_scene_upload")
        self._scene_keep = keep
        return flat

    # -- the hot path ------------------------------------------------------
    def render_film(self, render: Render, spp_begin=0, spp_count=None):
        """-> (H, W, 3) f32 numpy array of xyz_sum/spp (tile.rs:165) for the given sample range."""
        p = render.params(spp_begin, spp_count)
        out = np.zeros((render.height, render.width, 3), np.float32)
        rc = self.L.iris_render(self.ctx, C.byref(render.camera.c), C.byref(p), out.ctypes.data_as(C.POINTER(C.c_float)))
        _check(self.L, self.ctx, rc, "iris_render")
        return out

    def render_film_device(self, render: Render, d_out_ptr, spp_begin=0, spp_count=None, stream=None):
        """Same, into device memory (e.g. a torch CUDA tensor's data_ptr()) of H*W*3 f32."""
        p = render.params(spp_begin, spp_count)
        rc = self.L.iris_render_device(self.ctx, C.byref(render.camera.c), C.byref(p), C.c_void_p(d_out_ptr),
                                       C.c_void_p(stream or 0))
        _check(self.L, self.ctx, rc, "iris_render_device")

    def render_film_device_noalloc(self, render: Render, d_out_ptr, spp_begin=0, spp_count=None, stream=None):
        """iris_render_device_noalloc: the same film, with the workspace taken from the library's static device array
        (no allocation inside the call; IrisError with the sizes when the render does not fit it)."""
        p = render.params(spp_begin, spp_count)
        rc = self.L.iris_render_device_noalloc(self.ctx, C.byref(render.camera.c), C.byref(p), C.c_void_p(d_out_ptr),
                                               C.c_void_p(stream or 0))
        _check(self.L, self.ctx, rc, "iris_render_device_noalloc")

    # -- UpsampleTable on the device --------------------------------------
    def upload_upsample_table(self, table):
        """The RGB -> spectrum table (spectrum/upsample.rs:5-9) resident in this GPU's HBM, for get_spectrum_hdr_batch."""
        _check(self.L, self.ctx, self.L.iris_upsample_table_upload(self.ctx, table.handle), "iris_upsample_table_upload")

    def get_spectrum_hdr_batch(self, rgb, max_hdr=50.0):
        """UpsampleTable::get_spectrum_hdr (upsample.rs:46-56) for an (n, 3) array of RGB values on the GPU: returns
        ((n, 4) texels (c0, c1, c2, hdr_coefficient), (n,) luminance).  Host arrays in and out."""
        a = np.ascontiguousarray(np.asarray(rgb, np.float32).reshape(-1, 3))
        n = len(a)
        texels, imp = np.zeros((n, 4), np.float32), np.zeros(n, np.float32)
        rc = self.L.iris_upsample_get_spectrum_hdr_batch(self.ctx, a.ctypes.data, n, C.c_float(max_hdr), texels.ctypes.data, imp.ctypes.data, 0, None)
        _check(self.L, self.ctx, rc, "iris_upsample_get_spectrum_hdr_batch")
        return texels, imp

    def get_spectrum_hdr_batch_device(self, d_rgb_ptr, n, d_texels_ptr, d_importance_ptr=None, max_hdr=50.0, stream=None):
        """The same on DEVICE buffers of this GPU (raw pointers); enqueued on `stream`, returns without waiting."""
        rc = self.L.iris_upsample_get_spectrum_hdr_batch(self.ctx, d_rgb_ptr, n, C.c_float(max_hdr), d_texels_ptr, d_importance_ptr, 1, stream)
        _check(self.L, self.ctx, rc, "iris_upsample_get_spectrum_hdr_batch")

    def stats(self):
        st = ffi.Stats()
        _check(self.L, self.ctx, self.L.iris_get_stats(self.ctx, C.byref(st)), "iris_get_stats")
        return st

    def invariant_counts(self):
        """[Vec3 not finite, SpectralSample negative/NaN, PdfSet negative/NaN, light point off its sphere] of the last render:
        where the reference's debug build would have panicked (debug=True contexts only; IrisError otherwise)."""
        out = (C.c_uint64 * 4)()
        _check(self.L, self.ctx, self.L.iris_get_invariant_counts(self.ctx, out), "iris_get_invariant_counts")
        return [int(v) for v in out]

    # -- stage-level parity entry points ----------------------------------
    @staticmethod
    def _xys(xys):
        a = np.ascontiguousarray(np.asarray(xys, dtype=np.uint32).reshape(-1, 3))
        return a, a.ctypes.data_as(C.POINTER(C.c_uint32)), len(a)

    def dbg_sampler(self, xys, seed, ndraws):
        a, ptr, n = self._xys(xys)
        out = np.zeros((n, ndraws), np.float32)
        rc = self.L.iris_dbg_sampler(self.ctx, n, ptr, seed, ndraws, out.ctypes.data_as(C.POINTER(C.c_float)))
        _check(self.L, self.ctx, rc, "iris_dbg_sampler")
        return out

    def dbg_raygen(self, render: Render, xys):
        a, ptr, n = self._xys(xys)
        p = render.params()
        hero, o, d = np.zeros((n, 4), np.float32), np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
        fp = C.POINTER(C.c_float)
        rc = self.L.iris_dbg_raygen(self.ctx, C.byref(render.camera.c), C.byref(p), n, ptr, hero.ctypes.data_as(fp),
                                    o.ctypes.data_as(fp), d.ctypes.data_as(fp))
        _check(self.L, self.ctx, rc, "iris_dbg_raygen")
        return hero, o, d

    def dbg_intersect(self, ray_o, ray_d):
        o = np.ascontiguousarray(np.asarray(ray_o, np.float32).reshape(-1, 3))
        d = np.ascontiguousarray(np.asarray(ray_d, np.float32).reshape(-1, 3))
        n = len(o)
        prim, t = np.zeros(n, np.int32), np.zeros(n, np.float32)
        fp = C.POINTER(C.c_float)
        rc = self.L.iris_dbg_intersect(self.ctx, n, o.ctypes.data_as(fp), d.ctypes.data_as(fp),
                                       prim.ctypes.data_as(C.POINTER(C.c_int32)), t.ctypes.data_as(fp))
        _check(self.L, self.ctx, rc, "iris_dbg_intersect")
        return prim, t

    def dbg_trace(self, render: Render, xys):
        a, ptr, n = self._xys(xys)
        p = render.params()
        out = (ffi.Trace * n)()
        rc = self.L.iris_dbg_trace(self.ctx, C.byref(render.camera.c), C.byref(p), n, ptr, out)
        _check(self.L, self.ctx, rc, "iris_dbg_trace")
        return out


def render_film_multi(gpus, render: Render, spp_begin=0, spp_count=None):
    """iris_render_multi: the film rendered by several HipWavefront contexts (one per GPU, each holding the
    scene), sample indices split across them, partial films summed by one RCCL all-reduce over xGMI (on the host
    when contexts share a GPU).  -> (H, W, 3) f32."""
    L = ffi.load()
    p = render.params(spp_begin, spp_count)
    arr = (C.c_void_p * len(gpus))(*[g.ctx for g in gpus])
    out = np.zeros((render.height, render.width, 3), np.float32)
    rc = L.iris_render_multi(arr, len(gpus), C.byref(render.camera.c), C.byref(p), out.ctypes.data_as(C.POINTER(C.c_float)))
    _check(L, gpus[0].ctx, rc, "iris_render_multi")
    return out


def allreduce_films(gpus, device_ptrs, n_floats):
    """iris_allreduce_films: in-place sum of one device film per context (all on different GPUs) with RCCL."""
    L = ffi.load()
    ctxs = (C.c_void_p * len(gpus))(*[g.ctx for g in gpus])
    ptrs = (C.c_void_p * len(gpus))(*[C.c_void_p(p) for p in device_ptrs])
    _check(L, gpus[0].ctx, L.iris_allreduce_films(ctxs, len(gpus), ptrs, n_floats), "iris_allreduce_films")


def film_to_rgb(xyz):
    """Xyz::to_rgb_hdr + clamp (color.rs:36-41, tile.rs:108-109) on a (H, W, 3) film."""
    L = ffi.load()
    xyz = np.ascontiguousarray(xyz, np.float32)
    rgb = np.zeros_like(xyz)
    fp = C.POINTER(C.c_float)
    rc = L.iris_film_to_rgb(xyz.ctypes.data_as(fp), xyz.size // 3, rgb.ctypes.data_as(fp))
    if rc != ffi.IRIS_OK:
        raise ffi.IrisError("iris_film_to_rgb failed (%d)" % rc)
    return rgb
