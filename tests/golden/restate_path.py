#!/usr/bin/env python3
"""A SECOND, independent restatement of one camera sample through HwssNaive on the reference's default scene, in numpy
float32 scalars -- written from the Rust text (64/iris @ 2025-01-03, src/), not from oracle/iris_oracle.c and not from the
kernels: tile.rs:127-166, integrator/hwss_naive.rs:28-157, scene.rs:43-66,116-172, shape/sphere.rs:25-162,
shape/mod.rs:24-37, bsdf/lambertian.rs:23-51, math/ray.rs:14-85, math/vec3.rs:45-96, math/matrix.rs:168-196,
sampling/mod.rs:22-59, sampling/sampler.rs:13-88, sampling/mis.rs:3-10, spectrum/wavelength.rs:14-64,
spectrum/sample.rs:45-52, spectrum/constant.rs:15-23, color.rs:21-34.

Taken as given (they are this build's definitions, docs/SPEC.md sections 4 and 5, because the reference leaves them to a
crate and to libm): the Sobol points and sin/cos.  Both are implemented here from the SPEC's text -- the direction vectors
straight from scipy's Joe-Kuo table, the f64 fused multiply-adds in exact rational arithmetic -- so this file shares no code
with the oracle either.  The camera matrix (camera.rs:10-21: one tan() from libm) and the colour-matching tables
(color.rs:145-1565: data) are inputs.

    python tests/golden/restate_path.py          # writes tests/golden/traces_s1_restated.npz

tests/test_oracle.py checks the oracle against the file, tests/test_gpu_parity.py the GPU: a misreading of the Rust text
would now have to be made the same way twice, by two different programs, to go unnoticed.

Every arithmetic operation below is one IEEE binary32 operation (numpy float32 scalars), in the order the Rust source
writes it; Python floats never enter an expression.
"""
import os
import re
import sys
from fractions import Fraction

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
F = np.float32
M32 = 0xFFFFFFFF
np.seterr(all="ignore")

PI = F(3.14159265358979323846)  # std::f32::consts::PI
ZERO, ONE, TWO, HALF = F(0.0), F(1.0), F(2.0), F(0.5)


def bits(x):
    return int(np.array(x, dtype=np.float32).view(np.uint32))


def from_bits(u):
    return np.array(u & M32, dtype=np.uint32).view(np.float32)[()]


# ------------------------------------------------------------------ docs/SPEC.md section 5: sin / cos
S = [-1.66666666666666324348e-01, 8.33333333332248946124e-03, -1.98412698298579493134e-04, 2.75573137070700676789e-06,
     -2.50507602534068634195e-08, 1.58969099521155010221e-10]   # fdlibm __kernel_sin S1..S6
C = [4.16666666666666019037e-02, -1.38888888888741095749e-03, 2.48015872894767294178e-05, -2.75573143513906633035e-07,
     2.08757232129817482790e-09, -1.13596475577881948265e-11]   # fdlibm __kernel_cos C1..C6


def fma64(a, b, c):
    """binary64 fused multiply-add: exact product and sum, one rounding (int / int true division rounds correctly)"""
    r = Fraction(a) * Fraction(b) + Fraction(c)
    if r == 0:
        return a * b + c  # an exactly zero sum: the float expression is exact too and carries IEEE's sign of zero
    return r.numerator / r.denominator


def sincos(x):
    xd = float(x)
    if not abs(xd) < 1.0e5:
        return F(np.nan), F(np.nan)
    kd = float(round(xd * 0.6366197723675814))  # round-half-even
    r = fma64(-kd, 1.5707963267341256, xd)
    r = fma64(-kd, 6.077100506506192e-11, r)
    z = r * r
    p = fma64(z, S[5], S[4])
    for k in (3, 2, 1, 0):
        p = fma64(z, p, S[k])
    ks = fma64(r * z, p, r)
    p = fma64(z, C[5], C[4])
    for k in (3, 2, 1, 0):
        p = fma64(z, p, C[k])
    kc = fma64(z * z, p, fma64(z, -0.5, 1.0))
    q = int(kd) & 3
    s = kc if q & 1 else ks
    c = ks if q & 1 else kc
    return F(-s if q & 2 else s), F(-c if (q + 1) & 2 else c)


# ------------------------------------------------------------------ docs/SPEC.md section 4: Sobol points
def lowbias32(x):
    x ^= x >> 16
    x = (x * 0x7FEB352D) & M32
    x ^= x >> 15
    x = (x * 0x846CA68B) & M32
    x ^= x >> 16
    return x


def lk(n, s):
    n ^= (n * 0x3D20ADEA) & M32
    n = (n + s) & M32
    n = (n * ((s >> 16) | 1)) & M32
    n ^= (n * 0x05526C56) & M32
    n ^= (n * 0x53A22864) & M32
    return n


def bitreverse32(x):
    return int("{:032b}".format(x & M32)[::-1], 2)


_REV = None


def rev_vectors():
    """REV[dim][b]: the 16-bit direction vector of Joe-Kuo dimension `dim` for index bit b, bit-reversed.  scipy's
    unscrambled sequence is in Gray-code order, so the point at position 2^(b+1) - 1 (Gray code 2^b) IS vector b."""
    global _REV
    if _REV is None:
        from scipy.stats import qmc
        pts = qmc.Sobol(d=256, scramble=False).random(1 << 16)   # positions 0 .. 65535
        _REV = [[0] * 16 for _ in range(256)]
        for b in range(16):
            row = pts[(1 << (b + 1)) - 1]
            for dim in range(256):
                v16 = int(row[dim] * 65536.0)                     # exact: at most b+1 fractional bits
                _REV[dim][b] = int("{:016b}".format(v16)[::-1], 2)
    return _REV


def sobol_sample_4d(index, dset, seed):
    rev = rev_vectors()
    r = lk(bitreverse32(index), lowbias32(seed)) & 0xFFFF0000
    out = []
    for k in range(4):
        x = 0
        for b in range(16):
            if (r >> (31 - b)) & 1:
                x ^= rev[4 * dset + k][b]
        y = bitreverse32(lk(x, lowbias32(seed ^ ((0x9E3779B9 * (4 * dset + k + 1)) & M32))))
        out.append(from_bits((y >> 9) | 0x3F800000) - ONE)
    return out


# ------------------------------------------------------------------ sampling/sampler.rs
def hash_u32(n, seed):  # :74-83
    h = n & M32
    for _ in range(3):
        h = (h * 1936502639) & M32
        h ^= h >> 16
        h ^= seed
    return h


class Sampler:  # :4-72
    def __init__(self, x, y, sample_index, seed):
        self.scramble = hash_u32((x & M32) ^ ((y << 16) & M32), seed)
        self.dimension = 0
        self.index = sample_index & M32
        self.buf = [ZERO] * 4
        self.in_buf = 0
        self.draws = 0

    def gen_0_1(self):
        self.draws += 1
        if self.in_buf > 0:
            self.in_buf -= 1
            return self.buf[self.in_buf]
        assert self.dimension < 64, "past this build's Sobol table (docs/SPEC.md section 4)"
        self.buf = sobol_sample_4d(self.index, self.dimension, hash_u32(self.dimension, self.scramble))
        self.dimension += 1
        self.in_buf = 3
        return self.buf[3]

    def gen_array_index(self, length):  # :52-60: gen_range(0, len - 0.5) as usize
        u = self.gen_0_1() * ((F(length) - HALF) - ZERO) + ZERO
        return int(u)  # u >= 0 here; `as usize` truncates

    def gen_golden_ratio(self):  # :65-71
        u = (((self.index + self.scramble) & M32) * 2654435769) & M32
        return F(u) / F(4294967296.0)  # u32 -> f32 rounds to nearest; u32::MAX as f32 == 2^32


# ------------------------------------------------------------------ math/vec3.rs, point3.rs
class V:
    __slots__ = ("x", "y", "z")

    def __init__(self, x, y, z):
        self.x, self.y, self.z = F(x), F(y), F(z)

    def __add__(self, o): return V(self.x + o.x, self.y + o.y, self.z + o.z)
    def __sub__(self, o): return V(self.x - o.x, self.y - o.y, self.z - o.z)
    def __neg__(self): return V(-self.x, -self.y, -self.z)
    def scale(self, s): return V(self.x * s, self.y * s, self.z * s)        # Vec3 * f32 and f32 * Vec3
    def div(self, s): return V(self.x / s, self.y / s, self.z / s)          # Vec3 / f32: three divisions
    def dot(self, o): return self.x * o.x + self.y * o.y + self.z * o.z     # vec3.rs:45-48
    def cross(self, o):                                                      # vec3.rs:50-57
        return V(self.y * o.z - self.z * o.y, self.z * o.x - self.x * o.z, self.x * o.y - self.y * o.x)
    def len_squared(self): return self.dot(self)
    def len(self): return np.sqrt(self.len_squared())
    def normalize(self): return self.div(self.len())                        # vec3.rs:67-70
    def tolist(self): return [self.x, self.y, self.z]


def coordinate_system_from_unit(s):  # vec3.rs:77-85
    if abs(s.x) > abs(s.y):
        v2 = V(-s.z, ZERO, s.x).div(s.x * s.x + s.z * s.z)
    else:
        v2 = V(ZERO, s.z, -s.y).div(s.y * s.y + s.z * s.z)
    return v2, s.cross(v2)


def fmax(a, b):  # f32::max: a NaN operand is ignored
    if a != a:
        return b
    if b != b:
        return a
    return b if a < b else a


def fmin(a, b):
    if a != a:
        return b
    if b != b:
        return a
    return b if b < a else a


# ------------------------------------------------------------------ math/ray.rs
def as_i32(v):  # Rust `as i32`: saturating, NaN -> 0
    if v != v:
        return 0
    if v >= F(2147483648.0):
        return 2**31 - 1
    if v <= F(-2147483648.0):
        return -2**31
    return int(v)


def offset_origin(p, n):  # :61-85
    out = []
    for pc, nc in zip(p.tolist(), n.tolist()):
        of_i = as_i32(F(256.0) * nc)
        p_i = from_bits((bits(pc) + (-of_i if pc < ZERO else of_i)) & M32)
        out.append(pc + F(1.0 / 65536.0) * nc if abs(pc) < F(1.0 / 32.0) else p_i)
    return V(*out)


class Ray:
    def __init__(self, o, d_normalized):
        self.o, self.d = o, d_normalized

    @staticmethod
    def new(o, d):  # :14-19
        return Ray(o, d.normalize())

    @staticmethod
    def spawn(o, d, normal):  # :21-32
        o2 = offset_origin(o, -normal) if d.dot(normal) < ZERO else offset_origin(o, normal)
        return Ray(o2, d.normalize())

    @staticmethod
    def spawn_to(o, p, normal):  # :34-45: the direction starts at the OFFSET origin
        o2 = offset_origin(o, -normal) if (p - o).dot(normal) < ZERO else offset_origin(o, normal)
        return Ray(o2, (p - o2).normalize())

    def point_at(self, t):
        return self.o + self.d.scale(t)


# ------------------------------------------------------------------ shape/sphere.rs, shape/mod.rs
class Hit:
    pass


class Sphere:
    def __init__(self, c, r):
        self.position, self.radius = V(*c), F(r)

    def intersect(self, ray):  # :25-74
        oc = ray.o - self.position
        a = ray.d.len_squared()
        half_b = ray.d.dot(oc)
        c = oc.len_squared() - self.radius * self.radius
        discrim = half_b * half_b - a * c
        if discrim > ZERO:
            root = np.sqrt(discrim)
            for temp in ((-half_b - root) / a, (-half_b + root) / a):
                if temp > ZERO:
                    h = Hit()
                    h.point = ray.point_at(temp)
                    h.normal = (h.point - self.position).div(self.radius)
                    h.tangeant = V(ZERO, ONE, ZERO).cross(h.normal).normalize()
                    h.bitangeant = h.normal.cross(h.tangeant)
                    h.back_face = bool(h.normal.dot(ray.d) >= ZERO)
                    return h, temp
        return None

    def sample(self, hit, smp):  # :77-135
        u0, u1 = smp.gen_0_1(), smp.gen_0_1()
        point = offset_origin(hit.point, -hit.normal) if hit.back_face else offset_origin(hit.point, hit.normal)
        if (self.position - point).len_squared() <= self.radius * self.radius:
            z = ONE - TWO * u0                               # sampling::unit_sphere, mod.rs:26-31
            r = np.sqrt(fmax(ONE - z * z, ZERO))
            s, c = sincos(TWO * PI * u1)
            light_normal = V(r * c, r * s, z)
            light_point = self.position + light_normal.scale(self.radius)
            area = F(4.0) * PI * (self.radius * self.radius)
            pdf = (light_point - point).len_squared() / (light_normal.dot((light_point - point).normalize()) * area)
            return light_point, fmax(pdf, F(0.001))
        dc = np.sqrt((self.position - hit.point).len_squared())
        wc = (self.position - hit.point).normalize()
        wc_x, wc_y = coordinate_system_from_unit(wc)
        sin_theta_max = self.radius / dc
        sin_theta_max_2 = sin_theta_max * sin_theta_max
        cos_theta_max = np.sqrt(fmax(ONE - sin_theta_max_2, ZERO))
        if sin_theta_max_2 < F(0.00068523):
            sin_theta_2 = sin_theta_max_2 * u0
            cos_theta = np.sqrt(fmax(ONE - sin_theta_2, ZERO))
        else:
            cos_theta = (cos_theta_max - ONE) * u0 + ONE
            sin_theta_2 = ONE - cos_theta * cos_theta
        cos_alpha = sin_theta_2 / sin_theta_max + cos_theta * np.sqrt(fmax(ONE - sin_theta_2 / sin_theta_max_2, ZERO))
        sin_alpha = np.sqrt(fmax(ONE - cos_alpha * cos_alpha, ZERO))
        phi = u1 * TWO * PI
        sphi, cphi = sincos(phi)
        # Vec3::spherical_direction(sin_alpha, cos_alpha, phi, -wc_x, -wc_y, -wc), vec3.rs:87-96
        normal = ((-wc_x).scale(sin_alpha * cphi) + (-wc_y).scale(sin_alpha * sphi)) + (-wc).scale(cos_alpha)
        sampled_local = normal.normalize().scale(self.radius)
        sampled_world = self.position + sampled_local       # local_to_world: position.to_vec() + vec
        return sampled_world, pdf_cone(cos_theta_max)

    def pdf(self, hit, wi):  # :137-162
        point = offset_origin(hit.point, -hit.normal) if hit.back_face else offset_origin(hit.point, hit.normal)
        if (self.position - point).len_squared() <= self.radius * self.radius:
            ray = Ray.spawn(hit.point, wi, hit.normal)
            got = self.intersect(ray)
            if got is None:
                return ZERO
            lh, _ = got
            area = F(4.0) * PI * (self.radius * self.radius)
            pdf = (lh.point - point).len_squared() / (lh.normal.dot((lh.point - point).normalize()) * area)
            return fmax(pdf, F(0.001))
        sin_theta_max_2 = (self.radius * self.radius) / (self.position - hit.point).len_squared()
        cos_theta_max = np.sqrt(fmax(ONE - sin_theta_max_2, ZERO))
        return pdf_cone(cos_theta_max)


def pdf_cone(cos_theta_max):  # sampling/mod.rs:22-24
    return ONE / (TWO * PI * (ONE - cos_theta_max))


def world_to_shading(h, w):  # shape/mod.rs:24-30
    return V(h.bitangeant.dot(w), h.tangeant.dot(w), h.normal.dot(w))


def shading_to_world(h, s):  # shape/mod.rs:32-37
    return V(h.bitangeant.x * s.x + h.tangeant.x * s.y + h.normal.x * s.z,
             h.bitangeant.y * s.x + h.tangeant.y * s.y + h.normal.y * s.z,
             h.bitangeant.z * s.x + h.tangeant.z * s.y + h.normal.z * s.z)


# ------------------------------------------------------------------ 4-wide values (vec4.rs, sample.rs, pdf.rs)
def splat(v): return [F(v)] * 4
def mul4(a, b): return [x * y for x, y in zip(a, b)]
def add4(a, b): return [x + y for x, y in zip(a, b)]
def scale4(a, s): return [x * s for x in a]
def div4s(a, s): return [x / s for x in a]
def sum4(a): return (((F(-0.0) + a[0]) + a[1]) + a[2]) + a[3]   # f32x4::reduce_sum: ordered, from -0.0 (docs/SPEC.md section 1)


def balance_heuristic_1(f): return f[0] / sum4(f)               # mis.rs:3-5
def balance_heuristic_2(f, g): return f[0] / sum4(add4(f, g))   # mis.rs:8-10


# ------------------------------------------------------------------ bsdf/lambertian.rs over spectrum/constant.rs
class Lambertian:
    def __init__(self, albedo):
        self.albedo = F(albedo)

    def evaluate(self):  # :24-31: albedo.evaluate(lambda) / PI; a ConstantSpectrum is its value at every wavelength
        return div4s(splat(self.albedo), PI)

    def pdf(self, wi):   # :33-35 + sampling/mod.rs:56-59
        return splat(abs(wi.z) / PI)

    def sample(self, wo, smp):  # :37-50; arguments are evaluated left to right: the first draw is r1
        r1 = smp.gen_0_1()
        r2 = smp.gen_0_1()
        x_off, y_off = TWO * r1 - ONE, TWO * r2 - ONE       # concentric_disk, sampling/mod.rs:33-47
        if x_off == ZERO and y_off == ZERO:
            x, y = ZERO, ZERO
        else:
            if abs(x_off) > abs(y_off):
                r, theta = x_off, PI / F(4.0) * (y_off / x_off)
            else:
                r, theta = y_off, (PI / TWO) - (PI / F(4.0)) * (x_off / y_off)
            s, c = sincos(theta)
            x, y = r * c, r * s
        z = np.sqrt(fmax(ONE - x * x - y * y, ZERO))        # cosine_unit_hemisphere, :49-54
        wi = V(x, y, z)
        if not (wo.z * wi.z > ZERO):                        # same_hemisphere, vec3.rs:352-354
            wi = -wi
        return wi, self.evaluate(), self.pdf(wi)


# ------------------------------------------------------------------ scene.rs
class Prim:
    def __init__(self, sphere, material, light):
        self.sphere, self.material, self.light = sphere, material, light


def default_scene():  # Scene::dummy, scene.rs:43-66
    prims = [Prim(Sphere((0.0, 2.3, 3.0), 1.0), Lambertian(0.5), F(3.0)),      # add_emissive_material: ConstantSpectrum 3.0
             Prim(Sphere((0.0, -0.2, 3.0), 1.0), Lambertian(0.5), None),
             Prim(Sphere((0.0, -101.5, 2.0), 100.0), Lambertian(0.8), None)]
    lights = [0]  # lights: Vec<PrimIndex<Spectrum>>: one entry, prim_index 0
    return prims, lights


def scan(prims, ray):
    """the loop shared by Scene::intersection / ray_hits_point / ray_hits_object (scene.rs:116-166):
    closest t > 0 in primitive order, strict `<`"""
    closest_t, closest = F(np.inf), None
    for i, p in enumerate(prims):
        got = p.sphere.intersect(ray)
        if got is not None and got[1] < closest_t and got[1] > ZERO:
            closest_t, closest = got[1], (i, got[0])
    return closest_t, closest


# ------------------------------------------------------------------ integrator/hwss_naive.rs
MAX_DEPTH, MIN_DEPTH = 15, 2


def direct_light(prims, lights, bsdf, hit, ray, smp):  # :95-157
    radiance = splat(0.0)
    shading_wo = world_to_shading(hit, -ray.d)
    light_idx = smp.gen_array_index(len(lights))                       # scene.rs:168-172
    light_prim_index = lights[light_idx]
    light_prim = prims[light_prim_index]
    light_pick_weight = F(len(lights))
    light_emission = splat(light_prim.light)
    # sample light
    light_pos, light_pdf = light_prim.sphere.sample(hit, smp)
    ray_to_light = Ray.spawn_to(hit.point, light_pos, hit.normal)
    facing_forward = bool((light_pos - hit.point).dot(hit.normal) > ZERO)
    if light_pdf > ZERO and facing_forward != hit.back_face and (light_pos - hit.point).len_squared() > F(0.00001):
        closest_t, _ = scan(prims, ray_to_light)                       # scene.rs:134-149
        target_t = (light_pos - ray_to_light.o).len() / ray_to_light.d.len()
        if closest_t > target_t - F(0.001):
            shading_wi = world_to_shading(hit, ray_to_light.d)
            bsdf_values, bsdf_pdfs = bsdf.evaluate(), bsdf.pdf(shading_wi)
            cos_theta = abs(shading_wi.z)
            mis_weight = balance_heuristic_2(splat(light_pdf), bsdf_pdfs)
            radiance = add4(radiance, div4s(scale4(mul4(scale4(light_emission, mis_weight), bsdf_values), cos_theta), light_pdf))
    # sample BSDF
    wi, bsdf_values, bsdf_pdfs = bsdf.sample(shading_wo, smp)
    cos_theta = abs(wi.z)
    ray_to_light = Ray.spawn(hit.point, shading_to_world(hit, wi), hit.normal)
    if bsdf_pdfs[0] > ZERO:
        _, closest = scan(prims, ray_to_light)                         # scene.rs:151-166
        if closest is not None and closest[0] == light_prim_index:
            light_pdf = light_prim.sphere.pdf(hit, ray_to_light.d)
            mis_weight = balance_heuristic_2(bsdf_pdfs, splat(light_pdf))
            radiance = add4(radiance, div4s(scale4(mul4(scale4(light_emission, mis_weight), bsdf_values), cos_theta), bsdf_pdfs[0]))
    return scale4(radiance, light_pick_weight)


def radiance(prims, lights, ray, smp, trace):  # :28-93
    rad, throughput, path_pdfs = splat(0.0), splat(1.0), splat(1.0)
    for bounces in range(MAX_DEPTH):
        closest_t, closest = scan(prims, ray)                          # scene.rs:116-132
        if closest is None:
            break
        prim_index, hit = closest
        trace["hit_prim"].append(prim_index)
        trace["hit_t"].append(closest_t)
        prim = prims[prim_index]
        bsdf = prim.material
        if bsdf is None:
            break
        if bounces == 0 and prim.light is not None:
            rad = add4(rad, scale4(mul4(throughput, splat(prim.light)), balance_heuristic_1(path_pdfs)))
        rad = add4(rad, mul4(throughput, direct_light(prims, lights, bsdf, hit, ray, smp)))
        shading_wo = world_to_shading(hit, -ray.d)
        wi, bsdf_values, bsdf_pdfs = bsdf.sample(shading_wo, smp)
        cos_theta = abs(wi.z)
        if bsdf_pdfs[0] == ZERO or cos_theta == ZERO:
            break
        throughput = mul4(throughput, div4s(scale4(bsdf_values, cos_theta), bsdf_pdfs[0]))
        path_pdfs = mul4(path_pdfs, bsdf_pdfs)
        if bounces >= MIN_DEPTH:
            p = fmin(sum4(throughput), F(0.95))
            if smp.gen_0_1() > p:
                break
            throughput = [t / p for t in throughput]
        ray = Ray.spawn(hit.point, shading_to_world(hit, wi), hit.normal)
    return rad


# ------------------------------------------------------------------ spectrum/wavelength.rs, sample.rs, color.rs
def rotate_n(hero, n):  # wavelength.rs:55-64
    lam = hero + (F(470.0) / F(4.0)) * F(n)
    return lam - F(470.0) if lam >= F(830.0) else lam


def load_cmf():
    """CIE_X / CIE_Y / CIE_Z of color.rs:145-1565 -- data, read from the generated table"""
    text = open(os.path.join(ROOT, "oracle", "tables", "cie_cmf.inc")).read()
    out = []
    for name in ("IRIS_CIE_X", "IRIS_CIE_Y", "IRIS_CIE_Z"):
        body = text[text.index(name + "[471]"):]
        body = body[body.index("{") + 1:body.index("}")]
        vals = [F(v.rstrip("f")) for v in re.findall(r"[-+0-9.eE]+f", body)]
        assert len(vals) == 471
        out.append(vals)
    return out


def to_xyz(values, wl, cmf):  # sample.rs:45-52 over color.rs:21-34
    scale = (F(830.0) - F(360.0)) / F(116.661843131358)
    parts = []
    for lam, v in zip(wl, values):
        index = int(lam) - 360
        v = v * scale
        parts.append((cmf[0][index] * v, cmf[1][index] * v, cmf[2][index] * v))
    return [((parts[0][k] + parts[1][k]) + parts[2][k]) + parts[3][k] for k in range(3)]


# ------------------------------------------------------------------ tile.rs:127-166, one sample
def camera_sample(x, y, sample_index, width, height, seed, cam_pos, m):
    cx = ((F(x) + HALF) / F(width) - HALF) * TWO
    cy = ((F(y) + HALF) / F(height) - HALF) * F(-2.0)
    smp = Sampler(x, y, sample_index, seed)
    hero = smp.gen_golden_ratio() * F(470.0) + F(360.0)              # Wavelength::sample, wavelength.rs:26-28
    wl = [hero, rotate_n(hero, 1), rotate_n(hero, 2), rotate_n(hero, 3)]
    j0 = HALF * smp.gen_0_1() / F(width)
    j1 = HALF * smp.gen_0_1() / F(height)
    px, py, pz = cx + j0, cy + j1, ZERO + ZERO
    w = m[3][0] * px + m[3][1] * py + m[3][2] * pz + m[3][3]         # matrix.rs:168-196
    target = V(*[(m[r][0] * px + m[r][1] * py + m[r][2] * pz + m[r][3]) / w for r in range(3)])
    origin = V(*cam_pos)
    return smp, wl, Ray.new(origin, target - origin)


def trace_sample(x, y, sample_index, width, height, seed, cam_pos, m, cmf, scene):
    prims, lights = scene
    smp, wl, ray = camera_sample(x, y, sample_index, width, height, seed, cam_pos, m)
    tr = {"hit_prim": [], "hit_t": []}
    rad = radiance(prims, lights, ray, smp, tr)
    tr.update(hero=wl, ray_d=ray.d.tolist(), radiance=rad, xyz=to_xyz(rad, wl, cmf), draws=smp.draws)
    return tr


def main():
    sys.path.insert(0, ROOT)
    import iris_amd  # host-side camera only (iris_camera_new: camera.rs:10-21); no GPU, no oracle
    width = height = 512
    cam = iris_amd.Camera((0.0, 0.0, 0.0), F(width) / F(height))
    m = [[F(v) for v in row] for row in cam.clip_to_world]
    cam_pos = [F(v) for v in cam.position]
    cmf = load_cmf()
    scene = default_scene()
    rng = np.random.RandomState(20251111)
    xys = rng.randint(0, [width, height, 64], size=(256, 3)).astype(np.uint32)
    xys[16:, 1] = 230 + xys[16:, 1] % 282   # most of them below the horizon: spheres and ground, several bounces each
    xys[0] = (255, 40, 0)     # looks at the light
    xys[1] = (256, 300, 7)    # the middle sphere
    xys[2] = (30, 500, 63)    # the ground
    xys[3] = (5, 5, 1)        # the sky
    out = {k: [] for k in ("hero", "ray_d", "radiance", "xyz", "n_hits", "draws", "hit_prim", "hit_t")}
    for x, y, s in xys:
        tr = trace_sample(int(x), int(y), int(s), width, height, 123456789, cam_pos, m, cmf, scene)
        out["hero"].append(tr["hero"]); out["ray_d"].append(tr["ray_d"]); out["radiance"].append(tr["radiance"]); out["xyz"].append(tr["xyz"])
        out["n_hits"].append(len(tr["hit_prim"])); out["draws"].append(tr["draws"])
        out["hit_prim"].append((tr["hit_prim"] + [-1] * 16)[:16]); out["hit_t"].append((tr["hit_t"] + [ZERO] * 16)[:16])
    np.savez_compressed(os.path.join(HERE, "traces_s1_restated.npz"), xys=xys, width=np.uint32(width), height=np.uint32(height),
                        cam_pos=np.array(cam_pos, np.float32), clip_to_world=np.array(m, np.float32),
                        hero=np.array(out["hero"], np.float32), ray_d=np.array(out["ray_d"], np.float32),
                        radiance=np.array(out["radiance"], np.float32), xyz=np.array(out["xyz"], np.float32),
                        n_hits=np.array(out["n_hits"], np.uint32), draws=np.array(out["draws"], np.uint32),
                        hit_prim=np.array(out["hit_prim"], np.int32), hit_t=np.array(out["hit_t"], np.float32))
    print("wrote traces_s1_restated.npz: %d samples, %d bounce records, mean radiance %.4f" %
          (len(xys), int(np.sum(out["n_hits"])), float(np.mean(out["radiance"]))))


if __name__ == "__main__":
    main()
