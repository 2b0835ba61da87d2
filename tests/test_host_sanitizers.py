"""The library's host code (iris_host.cpp: camera, RGB->spectrum table, environment-map tables, .hdr reader; the hierarchy
build of iris_bvh_build.cpp; iris_sample_range needs the .hip file and is exercised through the library elsewhere) and the
oracle, compiled with AddressSanitizer + UndefinedBehaviorSanitizer (gcc, runtimes linked statically so that nothing has to
be preloaded) and driven over their edge cases by tests/host_san/host_san_main.cpp: malformed .hdr and .coeff files,
bvh_leaf 1..7 on one-sphere / all-coincident / random / sub-pixel sphere sets, dark and 1x1 environment maps, small oracle
renders with every integrator.  GPU code is not part of this build (sanitizers are host-only on this pool)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g", "-O1"]


def test_host_code_under_asan_and_ubsan(tmp_path):
    exe = str(tmp_path / "host_san")
    obj = str(tmp_path / "oracle.o")
    subprocess.check_call(["gcc", "-std=c11", "-ffp-contract=off", "-pthread", "-c"] + SAN +
                          ["-o", obj, os.path.join(ROOT, "oracle", "iris_oracle.c")])
    # iris_sample_range lives in the .hip file; the driver gets the same few lines through a tiny shim
    shim = tmp_path / "shim.cpp"
    shim.write_text('#include "%s/include/iris_hip.h"\n'
                    'extern "C" int iris_sample_range(uint32_t c, uint32_t p, uint32_t n, uint32_t *b, uint32_t *k) {\n'
                    '  if (!b || !k || n == 0 || p >= n) return IRIS_E_ARG;\n'
                    '  const uint64_t lo = (uint64_t)c * p / n, hi = (uint64_t)c * (p + 1) / n; *b = (uint32_t)lo; *k = (uint32_t)(hi - lo); return IRIS_OK; }\n' % ROOT)
    subprocess.check_call(["g++", "-std=c++17", "-ffp-contract=off", "-pthread", "-static-libasan", "-static-libubsan"] + SAN +
                          ["-o", exe, os.path.join(ROOT, "tests", "host_san", "host_san_main.cpp"),
                           os.path.join(ROOT, "iris_amd", "csrc", "iris_host.cpp"), os.path.join(ROOT, "iris_amd", "csrc", "iris_bvh_build.cpp"),
                           str(shim), obj, "-lm"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    env.pop("LD_PRELOAD", None) if False else None  # the environment is left as it is
    out = subprocess.run([exe, str(tmp_path), os.path.join(ROOT, "tests", "golden", "srgb_res16.coeff")], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-6000:])
    assert "all checks passed" in out.stdout
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr
