"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol include/iris_hip.h
declares, and its host-side helpers (camera, roughness mapping, XYZ->RGB) agree bit for bit with the
oracle.  No compute entry point is called here (there is no GPU in this run)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bits(a):
    return np.asarray(a, np.float32).view(np.uint32)


def test_library_exports_every_declared_symbol():
    from iris_amd import ffi
    header = open(os.path.join(ROOT, "include", "iris_hip.h")).read()
    declared = set(re.findall(r"^\s*(?:const\s+char\s*\*|int|void|float|uint32_t|size_t)\s*\*?\s*(iris_\w+)\s*\(", header, re.M))
    assert declared == set(ffi.EXPORTS), declared ^ set(ffi.EXPORTS)
    for path in (ffi.LIB_PATH, ffi.LIB_RELAXED_PATH, ffi.LIB_DEBUG_PATH):  # the exact build, the opt-in relaxed build and the debug-invariants build: one ABI
        L = C.CDLL(path)
        for name in declared:
            assert hasattr(L, name), (path, name)


def _c_prototypes(header):
    """name -> number of parameters, for every function include/iris_hip.h declares"""
    text = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    out = {}
    for m in re.finditer(r"(?:const\s+char\s*\*|int|void|float|uint32_t|size_t)\s*\*?\s*(iris_\w+)\s*\(([^;{]*?)\)\s*;", text, re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
    return out


def _c_struct_fields(header, name):
    text = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    body = re.search(r"typedef\s+struct\s*\{([^}]*)\}\s*%s\s*;" % name, text, re.S).group(1)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        for part in decl.split(","):
            fields.append(re.sub(r"\[.*?\]", "", part.strip().split()[-1].lstrip("*")))
    return fields


def test_rust_sys_crate_declares_the_same_abi():
    """ffi/iris-hip-sys/src/lib.rs (the binding a maintainer adds to iris; shipped as source, there is no Rust toolchain
    here): every `pub fn` is a symbol of include/iris_hip.h with the same number of parameters and vice versa, and every
    #[repr(C)] struct lists the header's fields in the header's order."""
    header = open(os.path.join(ROOT, "include", "iris_hip.h")).read()
    rust = open(os.path.join(ROOT, "ffi", "iris-hip-sys", "src", "lib.rs")).read()
    rust = re.sub(r"//[^\n]*", "", rust)
    want = _c_prototypes(header)
    got = {}
    for m in re.finditer(r"pub\s+fn\s+(iris_\w+)\s*\(([^)]*)\)", rust, re.S):
        args = m.group(2).strip()
        got[m.group(1)] = len([a for a in args.split(",") if a.strip()]) if args else 0
    assert got == want, {k: (got.get(k), want.get(k)) for k in set(got) | set(want) if got.get(k) != want.get(k)}
    from iris_amd import ffi
    assert set(got) == set(ffi.EXPORTS)
    for name in ("iris_sphere", "iris_spectrum", "iris_bsdf", "iris_light", "iris_envmap_desc", "iris_scene_desc", "iris_camera",
                 "iris_render_params", "iris_stats", "iris_trace"):
        body = re.search(r"pub\s+struct\s+%s\s*\{([^}]*)\}" % name, rust, re.S).group(1)
        fields = re.findall(r"pub\s+(\w+)\s*:", body)
        assert fields == _c_struct_fields(header, name), name
    # the wrapper type and the patch it needs are there, and use only declared symbols
    wrapper = open(os.path.join(ROOT, "ffi", "hip_wavefront.rs")).read()
    for sym in set(re.findall(r"sys::(iris_\w+)\s*\(", wrapper)):
        assert sym in want, sym
    assert os.path.exists(os.path.join(ROOT, "ffi", "iris_patch", "accessors.patch"))
    for f in ("Cargo.toml", "build.rs"):
        assert os.path.exists(os.path.join(ROOT, "ffi", "iris-hip-sys", f))


def test_abi_struct_sizes():
    from iris_amd import ffi
    assert C.sizeof(ffi.Sphere) == 16 and C.sizeof(ffi.Spectrum) == 20 and C.sizeof(ffi.Bsdf) == 20
    assert C.sizeof(ffi.Light) == 8 and C.sizeof(ffi.Camera) == 76 and C.sizeof(ffi.RenderParams) == 36
    assert C.sizeof(ffi.Trace) == 4 * (4 + 3 + 3 + 4 + 3 + 1 + 64 + 64 + 1)


def test_no_cpu_fallback_without_gpu():
    """On a machine without a HIP device the context cannot be created and says so."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import iris_amd
    with pytest.raises(iris_amd.IrisError):
        iris_amd.HipWavefront(device=0)


def test_product_does_not_reference_the_oracle():
    """Nothing under iris_amd/, include/ or bench.py's product leg may use oracle/."""
    for base, _, files in os.walk(os.path.join(ROOT, "iris_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".hpp", ".inc")):
                text = open(os.path.join(base, f)).read()
                assert "oracle_ffi" not in text and "liboracle" not in text and "iris_oracle" not in text, f


@pytest.mark.parametrize("aspect", [1.0, 16.0 / 9.0, 3840.0 / 2160.0, 0.75])
def test_camera_matches_oracle(oracle, aspect):
    """iris_camera_new (camera.rs:10-21) == the oracle's restatement, bit for bit."""
    import iris_amd
    for pos in ([0, 0, 0], [0.5, -1.25, 2.0]):
        cam = iris_amd.Camera(pos, np.float32(aspect))
        ocam = oracle.camera_new(pos, np.float32(aspect))
        assert np.array_equal(bits(cam.clip_to_world.ravel()), bits(np.array(ocam.clip_to_world)))
        assert np.array_equal(cam.position, np.array(ocam.position, np.float32))


def test_roughness_to_alpha_matches_oracle(oracle):
    from iris_amd import ffi
    L = ffi.load()
    for r in (1e-4, 1e-3, 0.05, 0.2, 0.5, 1.0):
        a = L.iris_roughness_to_alpha(C.c_float(r))
        assert a == oracle.lib().orc_roughness_to_alpha(C.c_float(r))
    assert abs(L.iris_roughness_to_alpha(C.c_float(1.0)) - 1.62142) < 1e-6


def test_film_to_rgb_matches_oracle(oracle):
    import iris_amd
    rng = np.random.RandomState(0)
    xyz = rng.uniform(-0.2, 3.0, size=(5, 7, 3)).astype(np.float32)
    xyz[0, 0] = np.nan
    rgb = iris_amd.film_to_rgb(xyz)
    want = np.zeros_like(xyz)
    fp = C.POINTER(C.c_float)
    for i in range(5):
        for j in range(7):
            o = np.zeros(3, np.float32)
            oracle.lib().orc_xyz_to_rgb_hdr(xyz[i, j].ctypes.data_as(fp), o.ctypes.data_as(fp))
            want[i, j] = np.where(np.isnan(o), 0.0, np.maximum(o, 0.0))  # f32::max(0.0), tile.rs:109
    assert np.array_equal(bits(rgb), bits(want))
    assert np.all(rgb >= 0.0)


def test_scene_flattening():
    """Scene.dummy() mirrors scene.rs:43-66: 3 primitives, 3 materials, 1 light on primitive 0."""
    from iris_amd import Scene, scene as S
    flat = Scene.dummy().flatten()
    assert flat.spheres.tolist() == [[0.0, np.float32(2.3), 3.0, 1.0], [0.0, np.float32(-0.2), 3.0, 1.0], [0.0, -101.5, 2.0, 100.0]]
    assert flat.prim_material.tolist() == [0, 1, 2] and flat.prim_light.tolist() == [0, -1, -1]
    assert [b[0] for b in flat.bsdfs] == [S.BSDF_LAMBERTIAN] * 3
    assert len(flat.lights) == 1 and flat.lights[0][1] == 0
    assert flat.spectra[flat.lights[0][0]] == (S.SPEC_CONSTANT, 3.0, 0.0, 0.0, 0.0)
    desc, keep = flat.to_desc()
    assert desc.n_prims == 3 and desc.n_bsdfs == 3 and desc.n_lights == 1 and desc.n_spectra == 4
    s2 = Scene.many_spheres(300)
    f2 = s2.flatten()
    assert len(f2.spheres) == 300 and f2.prim_light.tolist().count(0) == 1
    assert np.array_equal(f2.spheres, Scene.many_spheres(300).flatten().spheres)  # deterministic generator
    s3 = Scene.glossy_dielectric().flatten()
    assert [b[0] for b in s3.bsdfs] == [S.BSDF_LAMBERTIAN, S.BSDF_FRESNEL, S.BSDF_LAMBERTIAN, S.BSDF_MICROFACET, S.BSDF_SPECULAR]


# ---- the C++ host driver (iris_amd/host/: mirror of main.rs + scene.rs above the C ABI) --------------

def _cli():
    from iris_amd import build
    return build.build_cli()


@pytest.mark.parametrize("name,ctor", [("dummy", "dummy"), ("furnace", "furnace"), ("glossy", "glossy_dielectric"), ("spheres10k", "many_spheres")])
def test_cpp_host_scene_tables_equal_python_mirror(name, ctor):
    """The C++ Scene (add_material / add_emissive_material / dummy ...) flattens to byte-identical tables
    as the Python mirror, including the 10k-sphere generator."""
    import subprocess
    from iris_amd import Scene
    out = subprocess.run([_cli(), "--scene", name, "--dump-scene"], check=True, capture_output=True, text=True).stdout.split("\n")
    flat = getattr(Scene, ctor)().flatten()
    hx = lambda v: float(np.float32(v)).hex()
    want = []
    for i, s in enumerate(flat.spheres):
        want.append("prim %d %s %s %s %s %d %d" % (i, hx(s[0]), hx(s[1]), hx(s[2]), hx(s[3]), flat.prim_material[i], flat.prim_light[i]))
    for i, b in enumerate(flat.bsdfs):
        want.append("bsdf %d %d %d %d %s %s" % (i, b[0], b[1], b[2], hx(b[3]), hx(b[4])))
    for i, s in enumerate(flat.spectra):
        want.append("spectrum %d %d %s %s %s %s" % (i, s[0], hx(s[1]), hx(s[2]), hx(s[3]), hx(s[4])))
    for i, l in enumerate(flat.lights):
        want.append("light %d %d %d" % (i, l[0], l[1]))
    # C's %a and Python's float.hex() print the same value differently: compare parsed
    def parse(line):
        return [float.fromhex(t) if "0x" in t else t for t in line.split()]
    got = [parse(l) for l in out if l]
    assert got == [parse(l) for l in want]


def test_cpp_host_driver_fails_loudly_without_gpu(tmp_path):
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = subprocess.run([_cli(), "--width", "8", "--height", "8", "--spp", "1", "--out", str(tmp_path / "o.pfm")], capture_output=True, text=True)
    assert r.returncode != 0 and "no CPU fallback" in r.stderr
    assert not (tmp_path / "o.pfm").exists()
