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

    def to_desc(self):
        """Returns (iris_scene_desc, keepalive) for the C ABI."""
        n = len(self.spheres)
        spheres = (ffi.Sphere * n)(*[ffi.Sphere(*map(float, s)) for s in self.spheres])
        pm = (C.c_int32 * n)(*[int(v) for v in self.prim_material])
        pl = (C.c_int32 * n)(*[int(v) for v in self.prim_light])
        bs = (ffi.Bsdf * max(1, len(self.bsdfs)))(*[ffi.Bsdf(*b) for b in self.bsdfs])
        sp = (ffi.Spectrum * max(1, len(self.spectra)))(*[ffi.Spectrum(*s) for s in self.spectra])
        li = (ffi.Light * max(1, len(self.lights)))(*[ffi.Light(*l) for l in self.lights])
        desc = ffi.SceneDesc(n, spheres, pm, pl, len(self.bsdfs), bs, len(self.spectra), sp, len(self.lights), li)
        return desc, (spheres, pm, pl, bs, sp, li)


@dataclass
class Scene:
    """scene.rs:22-28: lights, materials, primitives (spheres only, shape/mod.rs:49-53)."""
    lights: List[tuple] = field(default_factory=list)     # PrimIndex<Spectrum>: (spectrum, prim_index)
    materials: List[object] = field(default_factory=list)  # PrimIndex<Bsdf>.data
    primitives: List[tuple] = field(default_factory=list)  # (Sphere, material_index|None, light_index|None)

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
        return FlatScene(spheres, pm, pl, bsdfs, spectra, lights)


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
