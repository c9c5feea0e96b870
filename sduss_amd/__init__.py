"""sduss_amd -- MI355X-native denoiser for the sduss/Mixfusion model slot.

Only what the hot path needs: ``csrc/`` (gfx950 HIP kernels + the C ABI of include/mxdenoise.h, built into
``libmxdenoise.so``), the ctypes binding (``lib``), the weight packer (``weights``), and the host-side mirrors of the
reference's interfaces for this path: ``unet.MxUNet`` (PatchUNet.forward), ``esymred_mp`` (the native op),
``pipeline.SDXLDenoiser.denoising_step``.
"""
from .config import UNetConfig  # noqa: F401

__all__ = ["UNetConfig"]
