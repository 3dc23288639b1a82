"""pixart_sigma_amd — MI355X-native (gfx950) implementation of the PixArt-Sigma XL/2 denoiser hot path.

Python host code mirrors the reference's module surface (diffusion.model.nets.PixArtMS / builder / IDDPM / DPMS) and
calls hand-written HIP kernels through the C ABI in include/pixart_hip.h (libpixart_hip.so, built by
`python -m pixart_sigma_amd.build`).  There is no CPU or eager-PyTorch fallback for the token path.
"""
from .diffusion import DPMS, IDDPM, SASolverSampler  # noqa: F401
from .model import MODELS, build_model  # noqa: F401
from .model.nets import PixArt, PixArt_XL_2, PixArtBlock, PixArtMS, PixArtMS_XL_2, PixArtMSBlock  # noqa: F401
