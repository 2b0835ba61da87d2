"""UpsampleTable resident in HBM and get_spectrum_hdr as a GPU batch (iris_upsample_table_upload,
iris_upsample_get_spectrum_hdr_batch): bit-exact against the host look-up iris_upsample_get_spectrum_hdr /
iris_envmap_texels_from_rgb (upsample.rs:46-106), which tests/test_upsample.py pins to the oracle on the CPU."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def rgb_cases(rng, n):
    rgb = rng.uniform(0.0, 1.0, size=(n, 3)).astype(np.float32)
    rgb[: n // 4] *= rng.uniform(1.0, 400.0, size=(n // 4, 1)).astype(np.float32)      # HDR values: the factor is split off
    rgb[n // 4: n // 3] = rng.uniform(0.0, 1.0, size=(n // 3 - n // 4, 1)).astype(np.float32)  # greys: ties in the max-channel rule
    rgb[-8:] = [[1, 1, 1], [1, 0, 0], [0, 1, 0], [0, 0, 1], [1e-6, 2e-6, 3e-6], [0.5, 0.5, 0.25], [50, 60, 70], [1, 0.999999, 1]]
    return rgb


def host_texels(table, rgb, max_hdr):
    from iris_amd import ffi
    n = len(rgb)
    tex, imp = np.zeros((n, 4), np.float32), np.zeros(n, np.float32)
    fp = C.POINTER(C.c_float)
    rc = ffi.load().iris_envmap_texels_from_rgb(table.handle, rgb.ctypes.data_as(fp), n, C.c_float(max_hdr), tex.ctypes.data_as(fp), imp.ctypes.data_as(fp))
    assert rc == 0
    return tex, imp


@pytest.mark.parametrize("max_hdr", [50.0, float("inf")])
def test_batch_equals_the_host_lookup_bit_for_bit(gpu, max_hdr):
    from iris_amd import UpsampleTable
    table = UpsampleTable(os.path.join(GOLD, "srgb_res16.coeff"))
    gpu.upload_upsample_table(table)
    rgb = rgb_cases(np.random.RandomState(31), 200000)
    tex, imp = gpu.get_spectrum_hdr_batch(rgb, max_hdr=max_hdr)
    want, want_imp = host_texels(table, rgb, max_hdr)
    assert not np.isnan(want).any()
    assert np.array_equal(bits(tex), bits(want))
    assert np.array_equal(bits(imp), bits(want_imp))
    # and the single-value entry point the reference's get_spectrum_hdr corresponds to (its clamp is 50)
    if max_hdr == 50.0:
        for k in (0, 1, 1000, len(rgb) - 3, len(rgb) - 2):
            s = table.get_spectrum_hdr(rgb[k])
            assert np.array_equal(bits(list(s.coefficients) + [s.hdr_coefficient]), bits(tex[k]))


def test_batch_equals_the_oracle_lookup_directly(gpu, oracle):
    """One hop, not two: the GPU batch against the ORACLE's get_spectrum_hdr / get_spectrum (oracle/iris_oracle.c's restatement of
    upsample.rs:46-106, on its own copy of the table), bit for bit -- 20 000 HDR, grey, tied and tiny values with the reference's
    clamp at 50, and LDR values through get_spectrum."""
    from iris_amd import UpsampleTable
    from test_upsample import TABLE, oracle_lookup
    table = UpsampleTable(TABLE)
    o = oracle.lib().orc_upsample_table_load(TABLE.encode())
    assert o
    try:
        gpu.upload_upsample_table(table)
        rgb = rgb_cases(np.random.RandomState(4321), 20000)
        tex, _ = gpu.get_spectrum_hdr_batch(rgb, max_hdr=50.0)
        for k in range(len(rgb)):
            c, h = oracle_lookup(oracle, o, rgb[k], hdr=True)
            assert np.array_equal(bits(tex[k]), bits(list(c) + [h])), (k, rgb[k])
        ldr = np.random.RandomState(99).uniform(0.0, 1.0, size=(4000, 3)).astype(np.float32)
        tex, _ = gpu.get_spectrum_hdr_batch(ldr)
        for k in range(len(ldr)):
            c, _ = oracle_lookup(oracle, o, ldr[k])
            assert np.array_equal(bits(tex[k, :3]), bits(c)) and tex[k, 3] == 1.0, (k, ldr[k])
    finally:
        oracle.lib().orc_upsample_table_free(o)


def test_batch_on_ldr_values_is_get_spectrum(gpu):
    """For RGB values in [0, 1] the HDR factor is 1 and the division by it exact, so the batch also serves as the device form of
    UpsampleTable::get_spectrum (upsample.rs:59-106): same three coefficients bit for bit, factor exactly 1."""
    from iris_amd import UpsampleTable
    table = UpsampleTable(os.path.join(GOLD, "srgb_res16.coeff"))
    gpu.upload_upsample_table(table)
    rgb = np.random.RandomState(77).uniform(0.0, 1.0, size=(512, 3)).astype(np.float32)
    rgb[0] = (1.0, 1.0, 1.0)
    tex, _ = gpu.get_spectrum_hdr_batch(rgb)
    assert np.all(tex[:, 3] == 1.0)
    for k in range(0, 512, 7):
        assert np.array_equal(bits(list(table.get_spectrum(rgb[k]).coefficients)), bits(tex[k, :3]))


def test_black_and_nan_values_follow_the_host(gpu):
    """A black texel divides by zero in get_spectrum (upsample.rs:67): NaN coefficients on both sides, same HDR factor."""
    from iris_amd import UpsampleTable
    table = UpsampleTable(os.path.join(GOLD, "srgb_res16.coeff"))
    gpu.upload_upsample_table(table)
    rgb = np.array([[0, 0, 0], [np.nan, 0.5, 0.25], [0.25, 0.5, 0.75], [-1, 0.5, 0.2]], np.float32)
    tex, _ = gpu.get_spectrum_hdr_batch(rgb)
    want, _ = host_texels(table, rgb, 50.0)
    assert np.array_equal(np.isnan(tex), np.isnan(want))
    ok = ~np.isnan(want)
    assert np.array_equal(bits(tex)[ok], bits(want)[ok])


def test_device_pointers_and_env_map_built_on_the_gpu(gpu):
    """on_device = 1 with raw device buffers; and EnvMap.from_rgb(gpu=...) gives the texels of the host path."""
    import hip_mem
    from iris_amd import EnvMap, UpsampleTable
    table = UpsampleTable(os.path.join(GOLD, "srgb_res16.coeff"))
    gpu.upload_upsample_table(table)
    rgb = rgb_cases(np.random.RandomState(5), 4096)
    n = len(rgb)
    d_rgb, d_tex, d_imp = hip_mem.alloc(n * 12), hip_mem.alloc(n * 16), hip_mem.alloc(n * 4)
    try:
        hip_mem.to_device(d_rgb, rgb)
        gpu.get_spectrum_hdr_batch_device(d_rgb, n, d_tex, d_imp, max_hdr=50.0)
        hip_mem.synchronize()
        tex = hip_mem.to_host(d_tex, (n, 4), np.float32)
        imp = hip_mem.to_host(d_imp, (n,), np.float32)
    finally:
        for p in (d_rgb, d_tex, d_imp):
            hip_mem.free(p)
    want, want_imp = host_texels(table, rgb, 50.0)
    assert np.array_equal(bits(tex), bits(want)) and np.array_equal(bits(imp), bits(want_imp))
    sky_rgb = np.random.RandomState(6).uniform(0.01, 3.0, size=(32, 32, 3)).astype(np.float32)
    a, b = EnvMap.from_rgb(sky_rgb, table), EnvMap.from_rgb(sky_rgb, table, gpu=gpu)
    assert np.array_equal(bits(a.texels), bits(b.texels)) and np.array_equal(bits(a.importance), bits(b.importance))


def test_batch_without_a_table_is_a_state_error():
    import iris_amd
    with iris_amd.HipWavefront(device=0) as ctx:
        with pytest.raises(iris_amd.ffi.IrisError):
            ctx.get_spectrum_hdr_batch(np.ones((4, 3), np.float32))
