"""SDXL-VAE / SD-VAE encode / decode on the HIP kernel set (csrc/vae.hip + pxa_gemm); mirrors diffusers' AutoencoderKL API."""
from .autoencoder_kl import AutoencoderKL, DiagonalGaussianDistribution  # noqa: F401
