"""Host-side mirror of the reference's scene-building interface (src/scene.rs, src/shape, src/bsdf,
src/spectrum), producing the flat `iris_scene_desc` the C ABI takes.

Same names and argument meaning as the reference so that tests read like it:
    scene.add_emissive_material(Sphere((0, 2.3, 3), 1.0), LambertianBsdf(ConstantSpectrum(0.5)), ConstantSpectrum(3.0))
is scene.rs:43-47 verbatim.
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np

from . import ffi

# enum orders of include/iris_hip.h (= bsdf/mod.rs:49-55, spectrum/mod.rs:25-28, integrator/mod.rs:8-11)
BSDF_LAMBERTIAN, BSDF_MICROFACET, BSDF_SPECULAR, BSDF_FRESNEL, BSDF_NULL = range(5)
SPEC_UPSAMPLED, SPEC_CONSTANT, SPEC_UPSAMPLED_HDR = range(3)
HWSS_NAIVE, HWSS_SLOW, SWSS_NAIVE, SWSS_SLOW = range(4)
INTEGRATORS = {"hwss_naive": HWSS_NAIVE, "hwss_slow": HWSS_SLOW, "swss_naive": SWSS_NAIVE, "swss_slow": SWSS_SLOW}

SEED = 123_456_789  # tile.rs:15
MAX_DEPTH = 15      # hwss_naive.rs:16
MIN_DEPTH = 2       # hwss_naive.rs:17


def _f32(x):
    return float(np.float32(x))


# ---- spectrum/ -------------------------------------------------------------
@dataclass
class ConstantSpectrum:  # spectrum/constant.rs
    value: float

    def flat(self):
        return (SPEC_CONSTANT, _f32(self.value), 0.0, 0.0, 0.0)


@dataclass
class UpsampledSpectrum:  # spectrum/upsample.rs:11-24
    coefficients: Tuple[float, float, float]

    def flat(self):
        c = self.coefficients
        return (SPEC_UPSAMPLED, _f32(c[0]), _f32(c[1]), _f32(c[2]), 0.0)


@dataclass
class UpsampledHdrSpectrum:  # spectrum/upsample.rs:27-42
    coefficients: Tuple[float, float, float]
    hdr_coefficient: float

    def flat(self):
        c = self.coefficients
        return (SPEC_UPSAMPLED_HDR, _f32(c[0]), _f32(c[1]), _f32(c[2]), _f32(self.hdr_coefficient))


class UpsampleTable:
    """spectrum/upsample.rs:5-9, :44-165: the RGB -> sigmoid-polynomial table of data/srgb.coeff.  The file is
    not in the reference tree; tools/rgb2spec_opt.c regenerates one in the same format."""

    def __init__(self, path):  # UpsampleTable::load, upsample.rs:108-144
        self.L = ffi.load()
        self.handle = C.c_void_p()
        rc = self.L.iris_upsample_table_load(str(path).encode(), C.byref(self.handle))
        if rc != ffi.IRIS_OK:
            self.handle = None
            raise ffi.IrisError("iris_upsample_table_load(%s) failed (%d): missing file, bad header or short file" % (path, rc))

    load = classmethod(lambda cls, path: cls(path))

    @property
    def resolution(self):
        return int(self.L.iris_upsample_table_resolution(self.handle))

    def _call(self, fn, rgb):
        arr = (C.c_float * 3)(*[float(np.float32(v)) for v in rgb])
        out = ffi.Spectrum()
        rc = fn(self.handle, arr, C.byref(out))
        if rc != ffi.IRIS_OK:
            raise ffi.IrisError("upsample look-up failed (%d)" % rc)
        return out

    def get_spectrum(self, rgb):  # upsample.rs:59-106
        o = self._call(self.L.iris_upsample_get_spectrum, rgb)
        return UpsampledSpectrum((o.c0, o.c1, o.c2))

    def get_spectrum_hdr(self, rgb):  # upsample.rs:46-56
        o = self._call(self.L.iris_upsample_get_spectrum_hdr, rgb)
        return UpsampledHdrSpectrum((o.c0, o.c1, o.c2), o.hdr)

    def close(self):
        if getattr(self, "handle", None):
            self.L.iris_upsample_table_free(self.handle)
            self.handle = None

    __del__ = close


class EnvMap:
    """Scene._env_map: Vec<UpsampledHdrSpectrum> (scene.rs:27).  The reference declares it and never fills it;
    layout and sampling are this build's (docs/SPEC.md section 6): (height, width, 4) f32 texels
    (c0, c1, c2, hdr_coefficient) in the octahedral layout, plus per-texel importance for the sampling CDFs."""

    def __init__(self, texels, importance=None):
        self.texels = np.ascontiguousarray(texels, np.float32)
        assert self.texels.ndim == 3 and self.texels.shape[2] == 4
        self.height, self.width = self.texels.shape[:2]
        self.importance = None if importance is None else np.ascontiguousarray(importance, np.float32).reshape(self.height, self.width)

    @classmethod
    def constant(cls, value, width=8, height=8):
        """Uniform white environment of spectral radiance `value`: zero sigmoid coefficients evaluate to 0.5
        (upsample.rs:35-41), so hdr_coefficient = 2 * value."""
        t = np.zeros((height, width, 4), np.float32)
        t[..., 3] = 2.0 * value
        return cls(t)

    @classmethod
    def from_rgb(cls, rgb_oct, table, max_hdr=float("inf"), gpu=None):
        """Octahedral (H, W, 3) linear-sRGB image -> texels through UpsampleTable::get_spectrum_hdr
        (upsample.rs:46-56; its clamp at 50 is the reference's stop-gap for maps that are NOT importance
        sampled, so the default here is no clamp); importance = luminance.
        gpu: a HipWavefront -- the conversion then runs as one batch on that GPU from the table in its HBM
        (iris_upsample_get_spectrum_hdr_batch; the same texels bit for bit)."""
        rgb = np.ascontiguousarray(rgb_oct, np.float32)
        h, w = rgb.shape[:2]
        if gpu is not None:
            gpu.upload_upsample_table(table)
            texels, imp = gpu.get_spectrum_hdr_batch(rgb.reshape(-1, 3), max_hdr=max_hdr)
            return cls(texels.reshape(h, w, 4), imp.reshape(h, w))
        texels = np.zeros((h, w, 4), np.float32)
        imp = np.zeros((h, w), np.float32)
        fp = C.POINTER(C.c_float)
        rc = ffi.load().iris_envmap_texels_from_rgb(table.handle, rgb.ctypes.data_as(fp), h * w, C.c_float(max_hdr),
                                                    texels.ctypes.data_as(fp), imp.ctypes.data_as(fp))
        if rc != ffi.IRIS_OK:
            raise ffi.IrisError("iris_envmap_texels_from_rgb failed (%d)" % rc)
        return cls(texels, imp)

    @staticmethod
    def latlong_to_oct(rgb_latlong, out_width, out_height):
        src = np.ascontiguousarray(rgb_latlong, np.float32)
        out = np.zeros((out_height, out_width, 3), np.float32)
        fp = C.POINTER(C.c_float)
        rc = ffi.load().iris_envmap_from_latlong(src.ctypes.data_as(fp), src.shape[1], src.shape[0], out_width, out_height, out.ctypes.data_as(fp))
        if rc != ffi.IRIS_OK:
            raise ffi.IrisError("iris_envmap_from_latlong failed (%d)" % rc)
        return out

    @staticmethod
    def read_hdr(path):
        """Radiance RGBE picture (.hdr) -> (H, W, 3) linear RGB f32 (iris_hdr_read)."""
        L = ffi.load()
        w, h = C.c_uint32(), C.c_uint32()
        rc = L.iris_hdr_read(str(path).encode(), C.byref(w), C.byref(h), None, 0)
        if rc != ffi.IRIS_OK:
            raise ffi.IrisError("iris_hdr_read: cannot read %s as an RGBE picture (%d)" % (path, rc))
        out = np.zeros((h.value, w.value, 3), np.float32)
        rc = L.iris_hdr_read(str(path).encode(), C.byref(w), C.byref(h), out.ctypes.data_as(C.POINTER(C.c_float)), out.size)
        if rc != ffi.IRIS_OK:
            raise ffi.IrisError("iris_hdr_read: %s is damaged (%d)" % (path, rc))
        return out

    @classmethod
    def from_hdr(cls, path, table, width=1024, height=1024, max_hdr=float("inf")):
        """An equirectangular .hdr picture (row 0 = +y pole) as an environment map: read, resample to the octahedral
        layout, upsample per texel."""
        return cls.from_rgb(cls.latlong_to_oct(cls.read_hdr(path), width, height), table, max_hdr)

    @staticmethod
    def texel_directions(width, height):
        """(H, W, 3) unit directions of the texel centres (iris_oct_to_dir)."""
        L = ffi.load()
        out = np.zeros((height, width, 3), np.float32)
        d = (C.c_float * 3)()
        for r in range(height):
            for c in range(width):
                L.iris_oct_to_dir(C.c_float(np.float32((c + 0.5) / width * 2 - 1)), C.c_float(np.float32((r + 0.5) / height * 2 - 1)), d)
                out[r, c] = d[:]
        return out

    @classmethod
    def sky(cls, table, width=256, height=256, sun_scale=400.0):
        """Procedural HDR sky (no .hdr file ships with the reference, .gitignore:7): blue-to-white gradient over a
        dark ground, and a small sun four hundred times brighter than the sky -- the case CDF sampling is for."""
        u = (np.arange(width, dtype=np.float64) + 0.5) / width * 2 - 1
        v = (np.arange(height, dtype=np.float64) + 0.5) / height * 2 - 1
        uu, vv = np.meshgrid(u, v)
        y = 1 - np.abs(uu) - np.abs(vv)
        x = np.where(y >= 0, uu, (1 - np.abs(vv)) * np.where(uu >= 0, 1.0, -1.0))
        z = np.where(y >= 0, vv, (1 - np.abs(uu)) * np.where(vv >= 0, 1.0, -1.0))
        n = np.sqrt(x * x + y * y + z * z)
        x, y, z = x / n, y / n, z / n
        up = np.clip(y, 0.0, 1.0)[..., None]
        zenith, horizon, ground = np.array([0.25, 0.45, 1.0]), np.array([0.9, 0.9, 0.85]), np.array([0.08, 0.07, 0.06])
        rgb = np.where(y[..., None] >= 0, horizon + (zenith - horizon) * np.sqrt(up), ground)
        sun = np.array([0.45, 0.65, -0.6])
        sun = sun / np.linalg.norm(sun)
        cos_s = x * sun[0] + y * sun[1] + z * sun[2]
        disc = np.clip((cos_s - np.cos(np.radians(4.0))) / (np.cos(np.radians(2.0)) - np.cos(np.radians(4.0))), 0.0, 1.0)
        rgb = rgb + disc[..., None] * sun_scale * np.array([1.0, 0.9, 0.7])
        return cls.from_rgb(rgb.astype(np.float32), table)

    def desc(self):
        """-> (iris_envmap_desc, keepalive)"""
        fp = C.POINTER(C.c_float)
        d = ffi.EnvmapDesc(self.width, self.height, self.texels.ctypes.data_as(fp),
                           self.importance.ctypes.data_as(fp) if self.importance is not None else None)
        return d, (self.texels, self.importance)

    def tables(self):
        """The sampling tables iris_scene_upload builds: func (H, W), cond_cdf (H, W+1), marg_cdf (H+1,), integral."""
        d, keep = self.desc()
        func = np.zeros((self.height, self.width), np.float32)
        cond = np.zeros((self.height, self.width + 1), np.float32)
        marg = np.zeros(self.height + 1, np.float32)
        integral = C.c_float()
        fp = C.POINTER(C.c_float)
        rc = ffi.load().iris_envmap_tables(C.byref(d), func.ctypes.data_as(fp), cond.ctypes.data_as(fp), marg.ctypes.data_as(fp), C.byref(integral))
        if rc != ffi.IRIS_OK:
            raise ffi.IrisError("iris_envmap_tables failed (%d)" % rc)
        return func, cond, marg, np.float32(integral.value)


# ---- shape/ ----------------------------------------------------------------
@dataclass
class Sphere:  # shape/sphere.rs:7-16
    position: Tuple[float, float, float]
    radius: float


# ---- bsdf/ -----------------------------------------------------------------
@dataclass
class LambertianBsdf:  # bsdf/lambertian.rs
    albedo: object

    def flat(self, add_spectrum):
        return (BSDF_LAMBERTIAN, add_spectrum(self.albedo), 0, 0.0, 0.0)


@dataclass
class MicrofacetBsdf:  # bsdf/microfacet.rs:20-30: roughness is mapped to alpha at construction
    reflectance: object
    roughness_x: float
    roughness_y: float

    def flat(self, add_spectrum):
        assert self.roughness_x != 0.0 and self.roughness_y != 0.0
        L = ffi.load()
        ax = L.iris_roughness_to_alpha(C.c_float(self.roughness_x))
        ay = L.iris_roughness_to_alpha(C.c_float(self.roughness_y))
        return (BSDF_MICROFACET, add_spectrum(self.reflectance), 0, _f32(ax), _f32(ay))


@dataclass
class SpecularBsdf:  # bsdf/specular.rs
    reflected_color: object

    def flat(self, add_spectrum):
        return (BSDF_SPECULAR, add_spectrum(self.reflected_color), 0, 0.0, 0.0)


@dataclass
class FresnelBsdf:  # bsdf/fresnel.rs:20-28
    reflected_color: object
    transmitted_color: object
    base_ior: float
    dispersion: float

    def flat(self, add_spectrum):
        return (BSDF_FRESNEL, add_spectrum(self.reflected_color), add_spectrum(self.transmitted_color),
                _f32(self.base_ior), _f32(self.dispersion))


@dataclass
class NullBsdf:  # bsdf/null.rs
    def flat(self, add_spectrum):
        return (BSDF_NULL, 0, 0, 0.0, 0.0)


# ---- scene.rs --------------------------------------------------------------
@dataclass
class FlatScene:
    spheres: np.ndarray        # (n, 4) f32: cx, cy, cz, r
    prim_material: np.ndarray  # (n,) i32, -1 = None
    prim_light: np.ndarray     # (n,) i32, -1 = None
    bsdfs: List[tuple]         # (kind, spec_a, spec_b, p0, p1)
    spectra: List[tuple]       # (kind, c0, c1, c2, hdr)
    lights: List[tuple]        # (spectrum, prim)
    env_map: Optional[EnvMap] = None

    def to_desc(self):
        """Returns (iris_scene_desc, keepalive) for the C ABI."""
        n = len(self.spheres)
        spheres = (ffi.Sphere * n)(*[ffi.Sphere(*map(float, s)) for s in self.spheres])
        pm = (C.c_int32 * n)(*[int(v) for v in self.prim_material])
        pl = (C.c_int32 * n)(*[int(v) for v in self.prim_light])
        bs = (ffi.Bsdf * max(1, len(self.bsdfs)))(*[ffi.Bsdf(*b) for b in self.bsdfs])
        sp = (ffi.Spectrum * max(1, len(self.spectra)))(*[ffi.Spectrum(*s) for s in self.spectra])
        li = (ffi.Light * max(1, len(self.lights)))(*[ffi.Light(*l) for l in self.lights])
        env_ptr, env_keep = None, None
        if self.env_map is not None:
            ed, ek = self.env_map.desc()
            env_ptr, env_keep = C.pointer(ed), (ed, ek)
        desc = ffi.SceneDesc(n, spheres, pm, pl, len(self.bsdfs), bs, len(self.spectra), sp, len(self.lights), li, env_ptr)
        return desc, (spheres, pm, pl, bs, sp, li, env_keep)


@dataclass
class Scene:
    """scene.rs:22-28: lights, materials, primitives (spheres only, shape/mod.rs:49-53)."""
    lights: List[tuple] = field(default_factory=list)     # PrimIndex<Spectrum>: (spectrum, prim_index)
    materials: List[object] = field(default_factory=list)  # PrimIndex<Bsdf>.data
    primitives: List[tuple] = field(default_factory=list)  # (Sphere, material_index|None, light_index|None)
    env_map: Optional[EnvMap] = None                        # scene.rs:27 (`_env_map`)

    def add_light(self, geom: Sphere, light):  # scene.rs:71-78
        self.lights.append((light, len(self.primitives)))
        self.primitives.append((geom, None, len(self.lights) - 1))

    def add_material(self, geom: Sphere, material):  # scene.rs:80-89
        self.materials.append(material)
        self.primitives.append((geom, len(self.materials) - 1, None))

    def add_emissive_material(self, geom: Sphere, material, light):  # scene.rs:91-110
        self.materials.append(material)
        self.lights.append((light, len(self.primitives)))
        self.primitives.append((geom, len(self.materials) - 1, len(self.lights) - 1))

    @classmethod
    def dummy(cls):
        """Scene::dummy, scene.rs:31-69: the reference's default scene (S1)."""
        scene = cls()
        scene.add_emissive_material(Sphere((0.0, 2.3, 3.0), 1.0), LambertianBsdf(ConstantSpectrum(0.5)), ConstantSpectrum(3.0))
        scene.add_material(Sphere((0.0, -0.2, 3.0), 1.0), LambertianBsdf(ConstantSpectrum(0.5)))
        scene.add_material(Sphere((0.0, -101.5, 2.0), 100.0), LambertianBsdf(ConstantSpectrum(0.8)))
        return scene

    @classmethod
    def furnace(cls, albedo=0.5, emission=0.25):
        """The "light oven" of scene.rs:36-41 / README.md:23: camera inside an emissive unit sphere;
        every pixel converges to emission / (1 - albedo)."""
        scene = cls()
        scene.add_emissive_material(Sphere((0.0, 0.0, 0.0), 1.0), LambertianBsdf(ConstantSpectrum(albedo)), ConstantSpectrum(emission))
        return scene

    @classmethod
    def many_spheres(cls, n_total=10_000, seed=0x1515):
        """S2 (BASELINE config D): S1's light and ground plus n_total-2 small Lambertian spheres from a
        fixed PCG32 stream, so that every implementation gets byte-identical arrays."""
        rng = _Pcg32(seed)
        scene = cls()
        scene.add_emissive_material(Sphere((0.0, 2.3, 3.0), 1.0), LambertianBsdf(ConstantSpectrum(0.5)), ConstantSpectrum(3.0))
        scene.add_material(Sphere((0.0, -101.5, 2.0), 100.0), LambertianBsdf(ConstantSpectrum(0.8)))
        while len(scene.primitives) < n_total:
            x = -20.0 + 40.0 * rng.uniform()
            y = -1.3 + 7.3 * rng.uniform()
            z = 2.0 + 58.0 * rng.uniform()
            r = 0.05 + 0.30 * rng.uniform()
            albedo = 0.2 + 0.7 * rng.uniform()
            # keep the light and the camera free
            if (x * x + (y - 2.3) ** 2 + (z - 3.0) ** 2) ** 0.5 < 1.0 + r + 0.05:
                continue
            if (x * x + y * y + z * z) ** 0.5 < r + 0.5:
                continue
            scene.add_material(Sphere((_f32(x), _f32(y), _f32(z)), _f32(r)), LambertianBsdf(ConstantSpectrum(_f32(albedo))))
        return scene

    @classmethod
    def glossy_dielectric(cls):
        """S3 (BASELINE config E): S1 with the middle sphere made a dispersive dielectric (the values in
        the comment at scene.rs:55-60), plus a GGX and a mirror sphere."""
        scene = cls()
        scene.add_emissive_material(Sphere((0.0, 2.3, 3.0), 1.0), LambertianBsdf(ConstantSpectrum(0.5)), ConstantSpectrum(3.0))
        scene.add_material(Sphere((0.0, -0.2, 3.0), 1.0), FresnelBsdf(ConstantSpectrum(1.0), ConstantSpectrum(1.0), 1.5220, 0.00459))
        scene.add_material(Sphere((0.0, -101.5, 2.0), 100.0), LambertianBsdf(ConstantSpectrum(0.8)))
        scene.add_material(Sphere((-2.2, -0.2, 3.5), 1.0), MicrofacetBsdf(ConstantSpectrum(0.9), 0.2, 0.2))
        scene.add_material(Sphere((2.2, -0.2, 3.5), 1.0), SpecularBsdf(ConstantSpectrum(0.9)))
        return scene

    @classmethod
    def enclosed(cls):
        """A test scene for the light code the designed scenes do not reach: the default scene plus a GGX sphere, all INSIDE a second
        light -- a sphere of radius 40 around the camera without a material (`add_light`, scene.rs:71-78: paths end on it).  Every
        shading point lies inside that light, so its `sample` and `pdf` take the uniform-sphere branch (sphere.rs:83-95, :141-153),
        and the light pick chooses between two lights (scene.rs:168-172)."""
        scene = cls()
        scene.add_emissive_material(Sphere((0.0, 2.3, 3.0), 1.0), LambertianBsdf(ConstantSpectrum(0.5)), ConstantSpectrum(3.0))
        scene.add_material(Sphere((0.0, -0.2, 3.0), 1.0), LambertianBsdf(ConstantSpectrum(0.5)))
        scene.add_material(Sphere((0.0, -101.5, 2.0), 100.0), LambertianBsdf(ConstantSpectrum(0.8)))
        scene.add_material(Sphere((-2.2, -0.2, 3.5), 1.0), MicrofacetBsdf(ConstantSpectrum(0.9), 0.2, 0.2))
        scene.add_light(Sphere((0.0, 0.0, 0.0), 40.0), ConstantSpectrum(0.25))
        return scene

    @classmethod
    def glossy_dielectric_env(cls, table, env_size=256):
        """BASELINE config E as named: the glossy/dielectric scene lit by an HDR environment map (procedural sky
        with a sun) in addition to the sphere light."""
        scene = cls.glossy_dielectric()
        scene.env_map = EnvMap.sky(table, env_size, env_size)
        return scene

    @classmethod
    def env_furnace(cls, albedo=0.5, emission=1.0):
        """White-furnace check of the environment map: one Lambertian sphere of albedo rho in a uniform
        environment of radiance L.  A convex body sees only the environment, so it reflects exactly rho*L in
        every direction: pixels on the sphere converge to rho*L, the others to L (rho = 1: the sphere vanishes)."""
        scene = cls()
        scene.add_material(Sphere((0.0, 0.0, 3.0), 1.0), LambertianBsdf(ConstantSpectrum(albedo)))
        scene.env_map = EnvMap.constant(emission)
        return scene

    @classmethod
    def coloured(cls, table):
        """S1's geometry with RGB materials and an HDR RGB light, all through the upsampling table
        (UpsampleTable::get_spectrum / get_spectrum_hdr, upsample.rs:46-106): exercises the sigmoid spectra."""
        scene = cls()
        scene.add_emissive_material(Sphere((0.0, 2.3, 3.0), 1.0), LambertianBsdf(table.get_spectrum((0.5, 0.5, 0.5))),
                                    table.get_spectrum_hdr((4.0, 3.5, 3.0)))
        scene.add_material(Sphere((0.0, -0.2, 3.0), 1.0), LambertianBsdf(table.get_spectrum((0.8, 0.15, 0.1))))
        scene.add_material(Sphere((0.0, -101.5, 2.0), 100.0), LambertianBsdf(table.get_spectrum((0.2, 0.6, 0.3))))
        scene.add_material(Sphere((-2.2, -0.2, 3.5), 1.0), MicrofacetBsdf(table.get_spectrum((0.9, 0.8, 0.3)), 0.3, 0.3))
        scene.add_material(Sphere((2.2, -0.2, 3.5), 1.0), SpecularBsdf(table.get_spectrum((0.3, 0.5, 0.9))))
        return scene

    def flatten(self) -> FlatScene:
        spectra: List[tuple] = []

        def add_spectrum(s):
            spectra.append(s.flat())
            return len(spectra) - 1

        bsdfs = [m.flat(add_spectrum) for m in self.materials]
        lights = [(add_spectrum(s), prim) for s, prim in self.lights]
        n = len(self.primitives)
        spheres = np.zeros((n, 4), np.float32)
        pm = np.full(n, -1, np.int32)
        pl = np.full(n, -1, np.int32)
        for i, (g, m, l) in enumerate(self.primitives):
            spheres[i] = (g.position[0], g.position[1], g.position[2], g.radius)
            pm[i] = -1 if m is None else m
            pl[i] = -1 if l is None else l
        return FlatScene(spheres, pm, pl, bsdfs, spectra, lights, self.env_map)


class _Pcg32:
    """PCG-XSH-RR 64/32 (O'Neill), default stream."""

    def __init__(self, seed):
        self.state = 0
        self.inc = (54 << 1) | 1
        self._next()
        self.state = (self.state + seed) & 0xFFFFFFFFFFFFFFFF
        self._next()

    def _next(self):
        old = self.state
        self.state = (old * 6364136223846793005 + self.inc) & 0xFFFFFFFFFFFFFFFF
        xorshifted = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
        rot = old >> 59
        return ((xorshifted >> rot) | (xorshifted << ((-rot) & 31))) & 0xFFFFFFFF

    def uniform(self):
        return (self._next() >> 8) * (1.0 / 16777216.0)
