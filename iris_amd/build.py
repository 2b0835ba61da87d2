"""Builds libiris_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m iris_amd.build [--force]

Flags that matter for correctness:
  -ffp-contract=off   one IEEE-754 f32 operation per source operation (the reference's Rust never
                      contracts a*b+c); the film is compared bit for bit with a scalar CPU evaluation.
  hipcc's default for HIP, -fhip-fp32-correctly-rounded-divide-sqrt, is left ON (IEEE divide/sqrt),
  and f32 denormals are left enabled (the gfx9 default).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libiris_hip.so")
# The opt-in relaxed-arithmetic build of the same sources (-DIRIS_RELAXED=1, iris_device.h): never the default, never what the
# parity tests load; Context(relaxed=True) / bench.py's `relaxed` leg select it.
LIB_RELAXED_PATH = os.path.join(LIB_DIR, "libiris_hip_relaxed.so")
# The debug-invariants build (-DIRIS_DEBUG_INVARIANTS=1, iris_device.h): counts, at fixed check points, where the reference's debug
# build would have panicked; Context(debug=True) selects it, tests/test_gpu_invariants.py compares with the oracle.
LIB_DEBUG_PATH = os.path.join(LIB_DIR, "libiris_hip_debug.so")
SOURCES = [os.path.join(CSRC, "iris_kernels.hip"), os.path.join(CSRC, "iris_host.cpp"), os.path.join(CSRC, "iris_bvh_build.cpp")]
DEPS = SOURCES + [os.path.join(CSRC, "iris_device.h"), os.path.join(CSRC, "iris_exact.h"), os.path.join(CSRC, "iris_bvh_build.h"), os.path.join(CSRC, "tables", "cie_cmf.inc"),
                  os.path.join(CSRC, "tables", "sobol_rev16.inc"), os.path.join(HERE, "..", "include", "iris_hip.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def needs_build(path=LIB_PATH):
    if not os.path.exists(path):
        return True
    t = os.path.getmtime(path)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force=False, extra=()):
    os.makedirs(LIB_DIR, exist_ok=True)
    jobs = []  # the two builds of the library run side by side (each is one long hipcc invocation)
    if force or needs_build(LIB_PATH):
        jobs.append(subprocess.Popen([HIPCC] + FLAGS + list(extra) + ["-o", LIB_PATH] + SOURCES))
    if force or needs_build(LIB_RELAXED_PATH):
        jobs.append(subprocess.Popen([HIPCC] + FLAGS + ["-DIRIS_RELAXED=1"] + list(extra) + ["-o", LIB_RELAXED_PATH] + SOURCES))
    if force or needs_build(LIB_DEBUG_PATH):
        jobs.append(subprocess.Popen([HIPCC] + FLAGS + ["-DIRIS_DEBUG_INVARIANTS=1"] + list(extra) + ["-o", LIB_DEBUG_PATH] + SOURCES))
    codes = [j.wait() for j in jobs]
    if any(codes):
        raise subprocess.CalledProcessError(max(codes), HIPCC)
    build_cli(force)
    return LIB_PATH


def build_experiment(out_path, extra=()):
    """A side build for tuning experiments (IRIS_HIP_LIB=<out_path> selects it): HwssNaive kernels only, which compiles
    four times faster.  Never the shipped library."""
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    subprocess.check_call([HIPCC] + FLAGS + ["-DIRIS_EXPERIMENT_INTEG0"] + list(extra) + ["-o", out_path] + SOURCES)
    return out_path


# The C++ host driver above the C ABI (iris_amd/host/: the mirror of the reference's main.rs + scene.rs).
CLI_PATH = os.path.join(LIB_DIR, "iris_render")
CLI_SOURCES = [os.path.join(HERE, "host", "iris_main.cpp")]
CLI_DEPS = CLI_SOURCES + [os.path.join(HERE, "host", "iris_scene.hpp"), os.path.join(HERE, "..", "include", "iris_hip.h")]


def build_cli(force=False):
    if not force and os.path.exists(CLI_PATH) and all(os.path.getmtime(d) <= os.path.getmtime(CLI_PATH) for d in CLI_DEPS + [LIB_PATH]):
        return CLI_PATH
    cmd = ["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-Wall", "-pthread", "-o", CLI_PATH] + CLI_SOURCES + \
          ["-L" + LIB_DIR, "-liris_hip", "-Wl,-rpath,$ORIGIN"]
    subprocess.check_call(cmd)
    return CLI_PATH


if __name__ == "__main__":
    if "--integ0" in sys.argv:  # python -m iris_amd.build --integ0 <out.so> [-Dflags...]
        i = sys.argv.index("--integ0")
        print(build_experiment(sys.argv[i + 1], extra=[a for a in sys.argv[i + 2:] if a.startswith("-")]))
    else:
        path = build(force="--force" in sys.argv, extra=[a for a in sys.argv[1:] if a.startswith("-") and a != "--force"])
        print(path)
