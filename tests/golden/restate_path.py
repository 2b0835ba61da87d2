#!/usr/bin/env python3
"""A SECOND, independent restatement of one camera sample, in numpy float32 scalars -- written from the Rust text (64/iris @
2025-01-03, src/), not from oracle/iris_oracle.c and not from the kernels: tile.rs:127-166, integrator/hwss_naive.rs:28-157,
integrator/swss_naive.rs:28-153, scene.rs:43-66,116-172, shape/sphere.rs:25-162, shape/mod.rs:24-37, bsdf/lambertian.rs:23-51,
bsdf/microfacet.rs:32-93, bsdf/specular.rs:25-54, bsdf/fresnel.rs:30-107, sampling/ggx.rs:7-77, math/mod.rs:35-72,
math/ray.rs:14-85, math/vec3.rs:45-111,301-355, math/matrix.rs:168-196, sampling/mod.rs:22-59, sampling/sampler.rs:13-88,
sampling/mis.rs:3-10, spectrum/wavelength.rs:14-64, spectrum/sample.rs:45-52, spectrum/constant.rs:15-23, color.rs:21-34.

Two sets of traces: HwssNaive on the reference's default scene (Lambertian only; traces_s1_restated.npz), and HwssNaive and
SwssNaive on the glossy/dielectric scene S3 at depth 32 (GGX microfacet, mirror, dispersive dielectric; traces_s3_restated.npz).

Taken as given (they are this build's definitions, docs/SPEC.md sections 4 and 5, because the reference leaves them to a
crate and to libm): the Sobol points and sin/cos.  Both are implemented here from the SPEC's text -- the direction vectors
straight from scipy's Joe-Kuo table, the f64 fused multiply-adds in exact rational arithmetic -- so this file shares no code
with the oracle either.  The camera matrix (camera.rs:10-21: one tan() from libm), the GGX alpha of the S3 scene's rough sphere
(ggx.rs:66-73: one ln() from libm, evaluated once when the scene is built) and the colour-matching tables (color.rs:145-1565:
data) are inputs.

    python tests/golden/restate_path.py          # writes tests/golden/traces_s1_restated.npz and traces_s3_restated.npz

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


def hash_u32_to_f32(n, seed):  # :85-88: INV_MAX = 1.0 / u32::MAX as f32 = 2^-32
    return F(hash_u32(n, seed)) * (ONE / F(4294967296.0))


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
        if self.dimension < 64:  # sobol::NUM_DIMENSIONS in units of 4-D sets (docs/SPEC.md section 4)
            self.buf = sobol_sample_4d(self.index, self.dimension, hash_u32(self.dimension, self.scramble))
        else:                    # :36-41
            self.buf = [hash_u32_to_f32(((self.dimension + k) & M32) ^ ((self.index << 16) & M32), self.scramble) for k in range(4)]
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
    def is_zero(self): return bool(self.x == ZERO and self.y == ZERO and self.z == ZERO)   # Vec3 == Vec3::splat(0.0), vec3.rs:139-144
    def face_forward(self, n): return self if self.dot(n) >= ZERO else -self               # vec3.rs:105-111


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


def clamp(x, lo, hi):  # f32::clamp: NaN stays NaN
    return lo if x < lo else (hi if x > hi else x)


# shading-space trigonometry, vec3.rs:301-355
def cos_theta(w): return w.z
def cos_2_theta(w): return w.z * w.z
def sin_2_theta(w): return fmax(ONE - cos_2_theta(w), ZERO)
def sin_theta(w): return np.sqrt(sin_2_theta(w))
def tan_theta(w): return sin_theta(w) / cos_theta(w)
def tan_2_theta(w): return sin_2_theta(w) / cos_2_theta(w)


def cos_phi(w):
    st = sin_theta(w)
    return ZERO if st == ZERO else clamp(w.x / st, F(-1.0), ONE)


def sin_phi(w):
    st = sin_theta(w)
    return ZERO if st == ZERO else clamp(w.y / st, F(-1.0), ONE)


def cos_2_phi(w): return cos_phi(w) * cos_phi(w)
def sin_2_phi(w): return sin_phi(w) * sin_phi(w)
def same_hemisphere(a, b): return bool(a.z * b.z > ZERO)


# ------------------------------------------------------------------ math/mod.rs
def fresnel_dielectric(cos_theta_i, eta_i, eta_t):  # :35-59
    cos_theta_i = clamp(cos_theta_i, F(-1.0), ONE)
    if cos_theta_i > ZERO:
        ei, et, ci = eta_i, eta_t, cos_theta_i
    else:
        ei, et, ci = eta_t, eta_i, abs(cos_theta_i)
    sin_theta_i = np.sqrt(fmax(ONE - ci * ci, ZERO))
    sin_theta_t = ei / et * sin_theta_i
    if sin_theta_t >= ONE:
        return ONE
    cos_theta_t = np.sqrt(fmax(ONE - sin_theta_t * sin_theta_t, ZERO))
    r_par = ((et * ci) - (ei * cos_theta_t)) / ((et * ci) + (ei * cos_theta_t))
    r_perp = ((ei * ci) - (et * cos_theta_t)) / ((ei * ci) + (et * cos_theta_t))
    return (r_par * r_par + r_perp * r_perp) / TWO


def refract(wi, n, eta):  # :61-72
    cos_theta_i = n.dot(wi)
    sin_2_theta_i = fmax(ONE - cos_theta_i * cos_theta_i, ZERO)
    sin_2_theta_t = eta * eta * sin_2_theta_i
    if sin_2_theta_t >= ONE:
        return None
    cos_theta_t = np.sqrt(ONE - sin_2_theta_t)
    return (-wi).scale(eta) + n.scale(eta * cos_theta_i - cos_theta_t)


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


# ------------------------------------------------------------------ bsdf/*.rs over spectrum/constant.rs
# Every spectrum of the two scenes is a ConstantSpectrum (constant.rs:15-23): its value at each of the four wavelengths.
class Lambertian:
    def __init__(self, albedo):
        self.albedo = F(albedo)

    def evaluate(self, wi, wo, wl):  # lambertian.rs:24-31: albedo.evaluate(lambda) / PI
        return div4s(splat(self.albedo), PI)

    def pdf(self, wi, wo):           # :33-35 + sampling/mod.rs:56-59
        return splat(abs(wi.z) / PI)

    def sample(self, wo, wl, smp):   # :37-50; arguments are evaluated left to right: the first draw is r1
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
        return wi, self.evaluate(wi, wo, wl), self.pdf(wi, wo)


# sampling/ggx.rs
def ggx_sample(wo, ax, ay, smp):  # :7-33
    wo = V(ax * wo.x, ay * wo.y, wo.z).normalize()
    len_2 = wo.x * wo.x + wo.y * wo.y
    t1_v = V(-wo.y, wo.x, ZERO).div(np.sqrt(len_2)) if len_2 > ZERO else V(ONE, ZERO, ZERO)
    t2_v = wo.cross(t1_v)
    r = np.sqrt(smp.gen_0_1())
    phi = TWO * PI * smp.gen_0_1()
    sphi, cphi = sincos(phi)
    t1 = r * cphi
    t2 = r * sphi
    s = HALF * (ONE + wo.z)
    t2 = (ONE - s) * np.sqrt(ONE - t1 * t1) + s * t2
    nh = (t1_v.scale(t1) + t2_v.scale(t2)) + wo.scale(np.sqrt(fmax(ONE - t1 * t1 - t2 * t2, ZERO)))
    return V(ax * nh.x, ay * nh.y, fmax(nh.z, ZERO)).normalize()


def ggx_evaluate(wh, ax, ay):  # :35-43
    if np.isinf(tan_2_theta(wh)):
        return ZERO
    e = (cos_2_phi(wh) / (ax * ax) + sin_2_phi(wh) / (ay * ay)) * tan_2_theta(wh)
    c2 = cos_theta(wh) * cos_theta(wh)                       # powi(4) = (x * x) * (x * x)
    return ONE / (PI * ax * ay * (c2 * c2) * ((ONE + e) * (ONE + e)))


def ggx_lambda(w, ax, ay):  # :45-55
    if np.isinf(tan_theta(w)):
        return ZERO
    alpha = np.sqrt(cos_2_phi(w) * (ax * ax) + sin_2_phi(w) * (ay * ay))
    a = alpha * abs(tan_theta(w))
    return (F(-1.0) + np.sqrt(ONE + a * a)) / TWO


def ggx_g(wo, wh, ax, ay): return ONE / (ONE + ggx_lambda(wh, ax, ay) + ggx_lambda(wo, ax, ay))   # :57-59
def ggx_g1(w, ax, ay): return ONE / (ONE + ggx_lambda(w, ax, ay))                                 # :75-77
def ggx_pdf(wo, wh, ax, ay): return ggx_evaluate(wh, ax, ay) * ggx_g1(wo, ax, ay) * abs(wo.dot(wh)) / abs(cos_theta(wo))  # :61-64


class Microfacet:  # bsdf/microfacet.rs
    def __init__(self, reflectance, alpha_x, alpha_y):
        self.reflectance, self.ax, self.ay = F(reflectance), F(alpha_x), F(alpha_y)

    def evaluate(self, wi, wo, wl):  # :33-57
        cos_theta_o, cos_theta_i = abs(cos_theta(wo)), abs(cos_theta(wi))
        wh = wo + wi
        if not same_hemisphere(wi, wo) or cos_theta_o == ZERO or cos_theta_i == ZERO or wh.is_zero():
            return splat(0.0)
        wh = wh.normalize()
        wh_facing = wh.face_forward(V(ZERO, ZERO, ONE))
        d = ggx_evaluate(wh, self.ax, self.ay)
        f = fresnel_dielectric(wi.dot(wh_facing), F(1.5), ONE)
        g = ggx_g(wo, wh, self.ax, self.ay)
        return div4s(scale4(scale4(scale4(splat(self.reflectance), d), g), f), F(4.0) * cos_theta_o * cos_theta_i)

    def pdf(self, wi, wo):           # :59-63
        wh = (wi + wo).normalize()
        return splat(ggx_pdf(wo, wh, self.ax, self.ay) / (F(4.0) * wo.dot(wh)))

    def sample(self, wo, wl, smp):   # :65-88
        wh = ggx_sample(wo, self.ax, self.ay, smp)
        wi = (-wo) + wh.scale(TWO * wo.dot(wh))              # reflect, :91-93: -wo + (2.0 * wo.dot(n) * n)
        if cos_theta(wo) == ZERO or wo.dot(wh) < ZERO or not same_hemisphere(wo, wi):
            return V(ZERO, ZERO, ZERO), splat(0.0), splat(0.0)
        pdf = ggx_pdf(wo, wh, self.ax, self.ay) / (F(4.0) * wo.dot(wh))
        return wi, self.evaluate(wi, wo, wl), splat(pdf)


class Specular:  # bsdf/specular.rs
    def __init__(self, reflected):
        self.reflected = F(reflected)

    def evaluate(self, wi, wo, wl): return splat(0.0)        # :26-33
    def pdf(self, wi, wo): return splat(0.0)                 # :35-37

    def sample(self, wo, wl, smp):   # :39-49: no draw
        wi = V(-wo.x, -wo.y, wo.z)
        return wi, div4s(splat(self.reflected), abs(cos_theta(wi))), splat(1.0)


class Fresnel:  # bsdf/fresnel.rs
    def __init__(self, reflected, transmitted, base_ior, dispersion):
        self.reflected, self.transmitted, self.base_ior, self.dispersion = F(reflected), F(transmitted), F(base_ior), F(dispersion)

    def evaluate(self, wi, wo, wl): return splat(0.0)        # :37-44
    def pdf(self, wi, wo): return splat(0.0)                 # :46-48

    def sample(self, wo, wl, smp):   # :50-102
        eta_a = ONE
        eta_b = [self.base_ior + self.dispersion / ((w * w) * F(1e-6)) for w in wl]   # :30-34
        fresnel = [fresnel_dielectric(cos_theta(wo), eta_a, e) for e in eta_b]
        if smp.gen_0_1() < fresnel[0]:
            wi = V(-wo.x, -wo.y, wo.z)
            bsdf = div4s(splat(self.reflected), abs(cos_theta(wi)))
            return wi, mul4(bsdf, fresnel), list(fresnel)
        eta_i, eta_t = (eta_a, eta_b[0]) if cos_theta(wo) > ZERO else (eta_b[0], eta_a)
        wi = refract(wo, V(ZERO, ZERO, ONE).face_forward(wo), eta_i / eta_t)
        if wi is None:               # total internal reflection
            return V(ZERO, ZERO, ZERO), splat(0.0), splat(0.0)
        ft = div4s(scale4(scale4(splat(self.transmitted), ONE - fresnel[0]), eta_i * eta_i), eta_t * eta_t)
        hero_value = ft[0] / abs(cos_theta(wi))
        return wi, [hero_value, ZERO, ZERO, ZERO], [ONE - fresnel[0], ZERO, ZERO, ZERO]


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


def glossy_dielectric_scene(alpha):
    """S3 (iris_amd.Scene.glossy_dielectric): the default scene with the middle sphere a dispersive dielectric
    (the values in the comment at scene.rs:55-60), a GGX sphere (roughness 0.2 -> `alpha`, ggx.rs:66-73) and a mirror sphere"""
    prims = [Prim(Sphere((0.0, 2.3, 3.0), 1.0), Lambertian(0.5), F(3.0)),
             Prim(Sphere((0.0, -0.2, 3.0), 1.0), Fresnel(1.0, 1.0, 1.5220, 0.00459), None),
             Prim(Sphere((0.0, -101.5, 2.0), 100.0), Lambertian(0.8), None),
             Prim(Sphere((-2.2, -0.2, 3.5), 1.0), Microfacet(0.9, alpha, alpha), None),
             Prim(Sphere((2.2, -0.2, 3.5), 1.0), Specular(0.9), None)]
    return prims, [0]


def enclosed_scene(alpha):
    """iris_amd.Scene.enclosed: the default scene and a GGX sphere inside a second light of radius 40 that has no material"""
    prims = [Prim(Sphere((0.0, 2.3, 3.0), 1.0), Lambertian(0.5), F(3.0)),
             Prim(Sphere((0.0, -0.2, 3.0), 1.0), Lambertian(0.5), None),
             Prim(Sphere((0.0, -101.5, 2.0), 100.0), Lambertian(0.8), None),
             Prim(Sphere((-2.2, -0.2, 3.5), 1.0), Microfacet(0.9, alpha, alpha), None),
             Prim(Sphere((0.0, 0.0, 0.0), 40.0), None, F(0.25))]
    return prims, [0, 4]


def scan(prims, ray):
    """the loop shared by Scene::intersection / ray_hits_point / ray_hits_object (scene.rs:116-166):
    closest t > 0 in primitive order, strict `<`"""
    closest_t, closest = F(np.inf), None
    for i, p in enumerate(prims):
        got = p.sphere.intersect(ray)
        if got is not None and got[1] < closest_t and got[1] > ZERO:
            closest_t, closest = got[1], (i, got[0])
    return closest_t, closest


# ------------------------------------------------------------------ integrator/hwss_naive.rs, integrator/swss_naive.rs
MIN_DEPTH = 2   # MAX_DEPTH = 15 in both files; the S3 traces run the same loop to 32 (BASELINE config E)


def direct_light(prims, lights, bsdf, hit, ray, wl, smp, hwss):  # hwss_naive.rs:95-157 / swss_naive.rs:92-153
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
            bsdf_values, bsdf_pdfs = bsdf.evaluate(shading_wi, shading_wo, wl), bsdf.pdf(shading_wi, shading_wo)
            cos_theta_ = abs(shading_wi.z)
            mis_weight = balance_heuristic_2(splat(light_pdf), bsdf_pdfs) if hwss else light_pdf / (bsdf_pdfs[0] + light_pdf)
            radiance = add4(radiance, div4s(scale4(mul4(scale4(light_emission, mis_weight), bsdf_values), cos_theta_), light_pdf))
    # sample BSDF
    wi, bsdf_values, bsdf_pdfs = bsdf.sample(shading_wo, wl, smp)
    cos_theta_ = abs(wi.z)
    ray_to_light = Ray.spawn(hit.point, shading_to_world(hit, wi), hit.normal)
    if bsdf_pdfs[0] > ZERO:
        _, closest = scan(prims, ray_to_light)                         # scene.rs:151-166
        if closest is not None and closest[0] == light_prim_index:
            light_pdf = light_prim.sphere.pdf(hit, ray_to_light.d)
            mis_weight = balance_heuristic_2(bsdf_pdfs, splat(light_pdf)) if hwss else bsdf_pdfs[0] / (bsdf_pdfs[0] + light_pdf)
            radiance = add4(radiance, div4s(scale4(mul4(scale4(light_emission, mis_weight), bsdf_values), cos_theta_), bsdf_pdfs[0]))
    return scale4(radiance, light_pick_weight)


def radiance(prims, lights, ray, wl, smp, trace, hwss=True, max_depth=15):  # hwss_naive.rs:28-93 / swss_naive.rs:28-90
    rad, throughput, path_pdfs = splat(0.0), splat(1.0), splat(1.0)
    for bounces in range(max_depth):
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
            if hwss:
                rad = add4(rad, scale4(mul4(throughput, splat(prim.light)), balance_heuristic_1(path_pdfs)))
            else:
                rad = add4(rad, mul4(throughput, splat(prim.light)))
        rad = add4(rad, mul4(throughput, direct_light(prims, lights, bsdf, hit, ray, wl, smp, hwss)))
        shading_wo = world_to_shading(hit, -ray.d)
        wi, bsdf_values, bsdf_pdfs = bsdf.sample(shading_wo, wl, smp)
        cos_theta_ = abs(wi.z)
        if bsdf_pdfs[0] == ZERO or cos_theta_ == ZERO:
            break
        throughput = mul4(throughput, div4s(scale4(bsdf_values, cos_theta_), bsdf_pdfs[0]))
        if hwss:
            path_pdfs = mul4(path_pdfs, bsdf_pdfs)
        if bounces >= MIN_DEPTH:
            p = fmin(sum4(throughput), F(0.95))
            if smp.gen_0_1() > p:
                break
            throughput = [t / p for t in throughput]
        ray = Ray.spawn(hit.point, shading_to_world(hit, wi), hit.normal)
    return rad if hwss else [rad[0], ZERO, ZERO, ZERO]                 # swss_naive.rs:87


def radiance_slow(prims, lights, ray, wl, smp, trace, hwss=True, max_depth=15):  # hwss_slow.rs:28-84 / swss_slow.rs:28-82
    """The integrators without next-event estimation: emission is collected wherever a path lands on a light (HWSS: weighted by
    balance_heuristic_1 of the path pdfs), the BSDF sample is the only continuation."""
    rad, throughput, path_pdfs = splat(0.0), splat(1.0), splat(1.0)
    for bounces in range(max_depth):
        closest_t, closest = scan(prims, ray)                          # scene.rs:116-132
        if closest is None:
            break
        prim_index, hit = closest
        trace["hit_prim"].append(prim_index)
        trace["hit_t"].append(closest_t)
        prim = prims[prim_index]
        if prim.light is not None:                                     # hwss_slow.rs:47-49 / swss_slow.rs:46-48: at every bounce
            if hwss:
                rad = add4(rad, scale4(mul4(throughput, splat(prim.light)), balance_heuristic_1(path_pdfs)))
            else:
                rad = add4(rad, mul4(throughput, splat(prim.light)))
        bsdf = prim.material
        if bsdf is None:
            break
        shading_wo = world_to_shading(hit, -ray.d)
        wi, bsdf_values, bsdf_pdfs = bsdf.sample(shading_wo, wl, smp)
        cos_theta_ = abs(wi.z)
        if bsdf_pdfs[0] == ZERO or cos_theta_ == ZERO:
            break
        throughput = mul4(throughput, div4s(scale4(bsdf_values, cos_theta_), bsdf_pdfs[0]))
        if hwss:
            path_pdfs = mul4(path_pdfs, bsdf_pdfs)
        if bounces >= MIN_DEPTH:
            p = fmin(sum4(throughput), F(0.95))
            if smp.gen_0_1() > p:
                break
            throughput = [t / p for t in throughput]
        ray = Ray.spawn(hit.point, shading_to_world(hit, wi), hit.normal)
    return rad if hwss else [rad[0], ZERO, ZERO, ZERO]                 # swss_slow.rs:81


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


def trace_sample(x, y, sample_index, width, height, seed, cam_pos, m, cmf, scene, hwss=True, max_depth=15, slow=False):
    prims, lights = scene
    smp, wl, ray = camera_sample(x, y, sample_index, width, height, seed, cam_pos, m)
    tr = {"hit_prim": [], "hit_t": []}
    rad = (radiance_slow if slow else radiance)(prims, lights, ray, wl, smp, tr, hwss, max_depth)
    tr.update(hero=wl, ray_d=ray.d.tolist(), radiance=rad, xyz=to_xyz(rad, wl, cmf), draws=smp.draws)
    return tr


def trace_set(xys, integrators, width, height, cam_pos, m, cmf, scene, max_depth, pad):
    out = {k: [] for k in ("hero", "ray_d", "radiance", "xyz", "n_hits", "draws", "hit_prim", "hit_t")}
    for (x, y, s), integ in zip(xys, integrators):
        # enum iris_integrator: 0 HwssNaive, 1 HwssSlow, 2 SwssNaive, 3 SwssSlow
        tr = trace_sample(int(x), int(y), int(s), width, height, 123456789, cam_pos, m, cmf, scene, hwss=(integ < 2), max_depth=max_depth, slow=bool(integ & 1))
        out["hero"].append(tr["hero"]); out["ray_d"].append(tr["ray_d"]); out["radiance"].append(tr["radiance"]); out["xyz"].append(tr["xyz"])
        out["n_hits"].append(len(tr["hit_prim"])); out["draws"].append(tr["draws"])
        out["hit_prim"].append((tr["hit_prim"] + [-1] * pad)[:pad]); out["hit_t"].append((tr["hit_t"] + [ZERO] * pad)[:pad])
    return out


def save(name, xys, integrators, width, height, cam_pos, m, out, extra=None):
    np.savez_compressed(os.path.join(HERE, name), xys=xys, width=np.uint32(width), height=np.uint32(height),
                        cam_pos=np.array(cam_pos, np.float32), clip_to_world=np.array(m, np.float32),
                        hero=np.array(out["hero"], np.float32), ray_d=np.array(out["ray_d"], np.float32),
                        radiance=np.array(out["radiance"], np.float32), xyz=np.array(out["xyz"], np.float32),
                        n_hits=np.array(out["n_hits"], np.uint32), draws=np.array(out["draws"], np.uint32),
                        hit_prim=np.array(out["hit_prim"], np.int32), hit_t=np.array(out["hit_t"], np.float32), **(extra or {}))
    print("wrote %s: %d samples, %d bounce records, mean radiance %.4f" %
          (name, len(xys), int(np.sum(out["n_hits"])), float(np.mean(out["radiance"]))))


def main():
    sys.path.insert(0, ROOT)
    import iris_amd  # host side only: the camera (iris_camera_new: camera.rs:10-21) and the GGX alpha (ggx.rs:66-73); no GPU, no oracle
    width = height = 512
    cam = iris_amd.Camera((0.0, 0.0, 0.0), F(width) / F(height))
    m = [[F(v) for v in row] for row in cam.clip_to_world]
    cam_pos = [F(v) for v in cam.position]
    cmf = load_cmf()
    # ---- the default scene, HwssNaive, depth 15
    rng = np.random.RandomState(20251111)
    xys = rng.randint(0, [width, height, 64], size=(256, 3)).astype(np.uint32)
    xys[16:, 1] = 230 + xys[16:, 1] % 282   # most of them below the horizon: spheres and ground, several bounces each
    xys[0] = (255, 40, 0)     # looks at the light
    xys[1] = (256, 300, 7)    # the middle sphere
    xys[2] = (30, 500, 63)    # the ground
    xys[3] = (5, 5, 1)        # the sky
    out = trace_set(xys, [0] * len(xys), width, height, cam_pos, m, cmf, default_scene(), 15, 16)
    save("traces_s1_restated.npz", xys, [0] * len(xys), width, height, cam_pos, m, out)
    # ---- the glossy/dielectric scene, depth 32: 192 samples through HwssNaive, 96 through SwssNaive; two thirds of them aimed at the
    # three non-Lambertian spheres (rows 230-330 of the 512-row film, columns around 256 and 256 -+ 115)
    import ctypes
    alpha = F(iris_amd.ffi.load().iris_roughness_to_alpha(ctypes.c_float(0.2)))
    rng = np.random.RandomState(20261111)
    n3 = 288
    xys3 = rng.randint(0, [width, height, 256], size=(n3, 3)).astype(np.uint32)
    for i in range(n3):
        if i % 3 != 2:
            cx = (256, 141, 371)[i % 3 if i % 6 < 3 else (i + 1) % 3]
            xys3[i, 0] = np.clip(cx + rng.randint(-45, 46), 0, width - 1)
            xys3[i, 1] = 235 + rng.randint(0, 95)
        else:
            xys3[i, 1] = 230 + xys3[i, 1] % 282
    integ3 = np.array([0] * 192 + [2] * 96, np.uint32)   # enum iris_integrator: 0 HwssNaive, 2 SwssNaive
    out3 = trace_set(xys3, integ3, width, height, cam_pos, m, cmf, glossy_dielectric_scene(alpha), 32, 32)
    save("traces_s3_restated.npz", xys3, integ3, width, height, cam_pos, m, out3, extra={"integrator": integ3, "alpha": np.float32(alpha), "max_depth": np.uint32(32)})
    kinds = {}
    for row, n in zip(out3["hit_prim"], out3["n_hits"]):
        for p_ in row[:n]:
            kinds[p_] = kinds.get(p_, 0) + 1
    print("S3 bounce records by primitive (0 light, 1 dielectric, 2 ground, 3 GGX, 4 mirror):", dict(sorted(kinds.items())))
    # ---- the integrators without next-event estimation (hwss_slow.rs, swss_slow.rs): 64 + 64 samples on the default scene at depth
    # 15 and 64 + 64 on the glossy/dielectric scene at depth 32 -- paths find the light only by landing on it, so samples are aimed
    # where that happens more often (near the light's reflection and through the dielectric)
    if "--slow" in sys.argv or "--all" in sys.argv:
        rng = np.random.RandomState(20271111)
        for name, scene, depth, pad in (("traces_s1_slow_restated.npz", default_scene(), 15, 16),
                                        ("traces_s3_slow_restated.npz", glossy_dielectric_scene(alpha), 32, 32)):
            n = 512
            xs = rng.randint(0, [width, height, 256], size=(n, 3)).astype(np.uint32)
            xs[:, 1] = 230 + xs[:, 1] % 282                       # spheres and ground
            near = np.arange(n) % 4 == 0                          # every fourth sample: at and around the light
            xs[near, 0] = np.clip(255 + rng.randint(-40, 41, int(near.sum())), 0, width - 1)
            xs[near, 1] = np.clip(40 + rng.randint(-40, 41, int(near.sum())), 0, height - 1)
            integ = np.array([1] * (n // 2) + [3] * (n // 2), np.uint32)
            o = trace_set(xs, integ, width, height, cam_pos, m, cmf, scene, depth, pad)
            save(name, xs, integ, width, height, cam_pos, m, o, extra={"integrator": integ, "alpha": np.float32(alpha), "max_depth": np.uint32(depth)})
            lit = int(np.sum(np.any(np.array(o["radiance"], np.float32) != 0, axis=1)))
            print("   samples with radiance:", lit, "of", len(xs))
    # ---- two lights, one of them around everything (iris_amd.Scene.enclosed): the uniform-sphere branches of Sphere::sample / pdf
    # (sphere.rs:83-95, :141-153), the pick among several lights (scene.rs:168-172), a light without a material; all four integrators
    if "--enclosed" in sys.argv or "--all" in sys.argv:
        rng = np.random.RandomState(20281111)
        n = 512
        xs = rng.randint(0, [width, height, 256], size=(n, 3)).astype(np.uint32)
        xs[n // 8:, 1] = 200 + xs[n // 8:, 1] % 312              # one eighth anywhere (the sky is the second light), the rest at the spheres and the ground
        integ = np.array([0, 1, 2, 3] * (n // 4), np.uint32)
        o = trace_set(xs, integ, width, height, cam_pos, m, cmf, enclosed_scene(alpha), 15, 16)
        save("traces_enclosed_restated.npz", xs, integ, width, height, cam_pos, m, o, extra={"integrator": integ, "alpha": np.float32(alpha), "max_depth": np.uint32(15)})
        kinds = {}
        for row, k in zip(o["hit_prim"], o["n_hits"]):
            for p_ in row[:k]:
                kinds[p_] = kinds.get(p_, 0) + 1
        print("   bounce records by primitive (0 small light, 1 sphere, 2 ground, 3 GGX, 4 enclosing light):", dict(sorted(kinds.items())))


if __name__ == "__main__":
    main()
