#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/.  Run in the build container:

    python tests/golden/make_golden.py

1. integer_kats.json -- known answers for the integer / exact-f32 parts of the path, computed HERE by an
   independent pure-Python restatement of the reference text (sampling/sampler.rs:13-21,65-88;
   spectrum/wavelength.rs:14-28,55-64; math/ray.rs:61-85).  The oracle (C) and the GPU are both checked
   against these, so a shared misreading would have to be made three times.
2. film_s1_32x32x4.npz, film_s3_24x24x8.npz, traces_s1.npz -- films and per-sample traces produced by the
   oracle in ORC_TRIG_IRIS mode (the mode the GPU matches bit for bit).  They let the GPU tests compare
   against committed data as well as against the live oracle.

3. film_s3env_24x24x8.npz -- the glossy/dielectric scene under a 16x16 environment map (docs/SPEC.md section 6):
   the map's texels and importance, its sampling tables and the oracle's film.

(The Sobol direction-vector fixture sobol_plain_256idx_256dim.npz is written by tools/gen_tables.py.)
"""
import json
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

M32 = 0xFFFFFFFF


def hash_u32(n, seed):  # sampler.rs:74-83
    h = n & M32
    for _ in range(3):
        h = (h * 1936502639) & M32
        h ^= h >> 16
        h ^= seed
    return h


def f32(x):
    return struct.unpack("<f", struct.pack("<f", x))[0]


def f32_bits(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


def from_bits(u):
    return struct.unpack("<f", struct.pack("<I", u & M32))[0]


def golden_ratio(x, y, s, seed):  # sampler.rs:13-21, 65-71
    scramble = hash_u32((x & M32) ^ ((y << 16) & M32), seed)
    u = (((s + scramble) & M32) * 2654435769) & M32
    return f32(f32(float(np.float32(u))) / f32(4294967296.0))


def wavelengths(u):  # wavelength.rs:14-28, 55-64
    hero = float(np.float32(np.float32(u) * np.float32(470.0)) + np.float32(360.0))
    out = [hero]
    for n in (1, 2, 3):
        lam = float(np.float32(hero) + np.float32(117.5) * np.float32(n))
        out.append(float(np.float32(lam) - np.float32(470.0)) if lam >= 830.0 else lam)
    return out


def as_i32(f):  # Rust `as i32`
    if f != f:
        return 0
    if f >= 2147483648.0:
        return 2**31 - 1
    if f <= -2147483648.0:
        return -2**31
    return int(f)


def offset_origin(p, n):  # ray.rs:61-85
    out = []
    for pc, nc in zip(p, n):
        pc, nc = f32(pc), f32(nc)
        of_i = as_i32(float(np.float32(256.0) * np.float32(nc)))
        bits = (f32_bits(pc) + (-of_i if pc < 0.0 else of_i)) & M32
        if abs(pc) < 1.0 / 32.0:
            out.append(float(np.float32(pc) + np.float32(1.0 / 65536.0) * np.float32(nc)))
        else:
            out.append(from_bits(bits))
    return out


def main():
    rng = np.random.RandomState(64)
    kats = {"hash_u32": [], "golden_ratio": [], "wavelengths": [], "offset_origin": []}
    for n, seed in [(0, 0), (1, 2), (123, 123456789), (M32, M32), (0x80000000, 1)] + \
            [(int(a), int(b)) for a, b in rng.randint(0, 2**32, size=(40, 2), dtype=np.uint64)]:
        kats["hash_u32"].append([n, seed, hash_u32(n, seed)])
    for x, y, s in [(0, 0, 0), (511, 511, 99), (1919, 1079, 1023), (3839, 2159, 4095), (13, 700, 65535)] + \
            [tuple(int(v) for v in r) for r in rng.randint(0, 4096, size=(30, 3))]:
        u = golden_ratio(x, y, s, 123456789)
        kats["golden_ratio"].append([x, y, s, 123456789, f32_bits(u)])
        kats["wavelengths"].append([f32_bits(u)] + [f32_bits(w) for w in wavelengths(u)])
    cases = [([0.0, 0.0, 0.0], [0.0, 1.0, 0.0]), ([1.0, -2.0, 3.0], [0.0, 0.0, 1.0]), ([0.01, -0.02, 5.0], [0.6, -0.8, 0.0]),
             ([-100.5, 0.03125, -0.03124], [-1.0, 0.5, 0.25]), ([1e-9, 1e9, -1e9], [0.3, 0.3, -0.9])]
    for _ in range(30):
        p = (rng.standard_normal(3) * (10.0 ** rng.randint(-3, 3))).astype(np.float32)
        nrm = rng.standard_normal(3)
        nrm = (nrm / np.linalg.norm(nrm)).astype(np.float32)
        cases.append(([float(v) for v in p], [float(v) for v in nrm]))
    for p, n in cases:
        kats["offset_origin"].append([[f32_bits(v) for v in p], [f32_bits(v) for v in n], [f32_bits(v) for v in offset_origin(p, n)]])
    json.dump(kats, open(os.path.join(HERE, "integer_kats.json"), "w"), indent=0)

    # ---- oracle-produced films / traces (IRIS trig mode) ----
    import oracle_ffi as O
    from iris_amd import Scene
    O.build()
    O.set_trig_mode(O.TRIG_IRIS)
    cam = O.camera_new([0, 0, 0], 1.0)
    s1 = O.OracleScene(Scene.dummy().flatten())
    p = O.make_params(32, 32, 4)
    film, _, _, st = s1.render(cam, p, nthreads=4)
    np.savez_compressed(os.path.join(HERE, "film_s1_32x32x4.npz"), film=film,
                        stats=np.array([st.samples, st.bounces, st.rays, st.sphere_tests], np.uint64))
    s3 = O.OracleScene(Scene.glossy_dielectric().flatten())
    p3 = O.make_params(24, 24, 8, max_depth=32)
    film3, _, _, st3 = s3.render(cam, p3, nthreads=4)
    np.savez_compressed(os.path.join(HERE, "film_s3_24x24x8.npz"), film=film3,
                        stats=np.array([st3.samples, st3.bounces, st3.rays, st3.sphere_tests], np.uint64))
    pt = O.make_params(512, 512, 64)
    xys = rng.randint(0, [512, 512, 64], size=(64, 3)).astype(np.uint32)
    tr = [s1.trace(cam, pt, int(x), int(y), int(s)) for x, y, s in xys]
    np.savez_compressed(
        os.path.join(HERE, "traces_s1.npz"), xys=xys,
        hero=np.array([list(t.hero) for t in tr], np.float32), ray_d=np.array([list(t.ray_d) for t in tr], np.float32),
        radiance=np.array([list(t.radiance) for t in tr], np.float32), xyz=np.array([list(t.xyz) for t in tr], np.float32),
        n_hits=np.array([t.n_hits for t in tr], np.uint32), draws=np.array([t.draws for t in tr], np.uint32),
        hit_prim=np.array([list(t.hit_prim[:16]) for t in tr], np.int32), hit_t=np.array([list(t.hit_t[:16]) for t in tr], np.float32))
    # ---- environment map (docs/SPEC.md section 6): texels are stored with the film, so the fixture does not
    # depend on how this machine's numpy evaluates the procedural sky ----
    from iris_amd import EnvMap, UpsampleTable
    table = UpsampleTable(os.path.join(HERE, "srgb_res16.coeff"))
    env = EnvMap.sky(table, 16, 16)
    sc = Scene.glossy_dielectric()
    sc.env_map = env
    se = O.OracleScene(sc.flatten())
    filme, _, _, ste = se.render(cam, O.make_params(24, 24, 8, max_depth=32), nthreads=4)
    func, cond, marg, integral = se.env_tables(16, 16)
    np.savez_compressed(os.path.join(HERE, "film_s3env_24x24x8.npz"), film=filme, texels=env.texels, importance=env.importance,
                        func=func, cond_cdf=cond, marg_cdf=marg, integral=np.float32(integral),
                        stats=np.array([ste.samples, ste.bounces, ste.rays, ste.sphere_tests], np.uint64))
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
