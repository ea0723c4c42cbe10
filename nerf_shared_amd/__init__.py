"""nerf_shared_amd -- MI355X (gfx950) native drop-in for nerf_shared's render hot path.

    from nerf_shared_amd import nerf, render_utils, utils      # same module names as the reference

``nerf.NeRF``, ``nerf.get_embedder``, ``render_utils.Renderer`` and
``utils.{get_rays, ndc_rays, sample_pdf, ...}`` keep the reference signatures;
the arithmetic runs in hand-written HIP kernels behind the C ABI of
include/nerf_amd.h (nerf_shared_amd/libnerf_amd.so).  Importing the submodules
fails loudly if that library has not been built -- there is no CPU fallback.
``synth`` (deterministic synthetic weights/cameras) imports without it.
"""
__all__ = ["nerf", "render_utils", "utils", "synth", "dist"]


def __getattr__(name):
    if name in __all__:
        import importlib
        return importlib.import_module("." + name, __name__)
    raise AttributeError(name)
