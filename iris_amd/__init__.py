"""iris_amd -- MI355X (gfx950) implementation of the per-pixel path-tracing hot path of the iris
spectral renderer, behind the C ABI of include/iris_hip.h.

    from iris_amd import HipWavefront, Render, Scene
    gpu = HipWavefront(device=0)
    gpu.upload(Scene.dummy())
    film_xyz = gpu.render_film(Render(512, 512, 64, Scene.dummy()))

The compute path is libiris_hip.so (hand-written HIP); there is no CPU fallback.
"""
from .ffi import IrisError  # noqa: F401
from .render import Camera, HipWavefront, Render, allreduce_films, film_to_rgb, render_film_multi  # noqa: F401
from .scene import (  # noqa: F401
    HWSS_NAIVE, HWSS_SLOW, SWSS_NAIVE, SWSS_SLOW, INTEGRATORS, ConstantSpectrum, EnvMap, FresnelBsdf, LambertianBsdf,
    MicrofacetBsdf, NullBsdf, Scene, SpecularBsdf, Sphere, UpsampledHdrSpectrum, UpsampledSpectrum, UpsampleTable,
)
