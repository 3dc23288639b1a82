"""BASELINE.json config 5: PixArt-alpha-DMD 512px one-step generator, batch 64 per GPU (the VAE-decode-bound path).
One generation = one PixArtMS forward at t = 400 without CFG (reference app/app_pixart_dmd.py:193-196: timesteps=[400],
guidance_scale=1, 1 step), x0 = (x_t - sqrt(1 - abar_t) eps) / sqrt(abar_t), then SD-VAE decode of x0 / 0.18215.
Random-init weights, synthetic caption features (L = 120).  Parity of this exact step: tests/test_model_gpu.py::test_dmd_one_step_generator_matches_reference
(reference generate.py:20-41); of the decoder: tests/test_vae_gpu.py.  Usage: python tools/bench_dmd.py [--batch 64] [--iters 3]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# like the reference app (fp16 pipeline, app/app_pixart_dmd.py) the default is the fp16-operand build; --dtype bf16 selects the other library
os.environ["PXA_OPERAND_DTYPE"] = "bf16" if ("--dtype" in sys.argv[:-1] and sys.argv[sys.argv.index("--dtype") + 1] == "bf16") else "f16"
import numpy as np  # noqa: E402
import torch  # noqa: E402

from bench import MFMA_PEAK, fwd_flops_per_sample  # noqa: E402
from vae_layer_table import decode_schedule  # noqa: E402
from pixart_sigma_amd import PixArtMS_XL_2  # noqa: E402
from pixart_sigma_amd.vae import AutoencoderKL  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--dtype", choices=["fp16", "bf16"], default="fp16")
    a = ap.parse_args()
    B, lat, L = a.batch, 64, 120
    torch.manual_seed(0)
    m = PixArtMS_XL_2(input_size=lat, pe_interpolation=1.0, model_max_length=L)
    with torch.no_grad():
        for blk in m.blocks:
            blk.cross_attn.proj.weight.normal_(std=0.02)
        m.final_layer.linear.weight.normal_(std=0.02)
    m = m.cuda().eval()
    vae = AutoencoderKL(scaling_factor=0.18215).cuda()        # random-init weights (torch's default Conv2d / Linear init under the seed above; GroupNorm 1 / 0)
    betas = np.linspace(1e-4, 2e-2, 1000, dtype=np.float64)          # the linear schedule of the alpha / DMD checkpoints
    abar = float(np.cumprod(1.0 - betas)[400])
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, 4, lat, lat, generator=g).cuda()
    y = torch.randn(B, 1, L, 4096, generator=g).cuda()
    mask = torch.ones(B, L, dtype=torch.int64)
    t = torch.full((B,), 400, device="cuda", dtype=torch.long)

    def generate():
        eps = m.forward_with_dpmsolver(x, t, y, data_info=None, mask=mask)
        x0 = (x - (1.0 - abar) ** 0.5 * eps) / abar ** 0.5
        return vae.decode(x0 / vae.config.scaling_factor).sample, eps

    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    with torch.no_grad():
        img, _ = generate()
        torch.cuda.synchronize()
        t_dit = t_all = 0.0
        for _ in range(a.iters):
            ev[0].record()
            eps = m.forward_with_dpmsolver(x, t, y, data_info=None, mask=mask)
            ev[1].record()
            x0 = (x - (1.0 - abar) ** 0.5 * eps) / abar ** 0.5
            img = vae.decode(x0 / vae.config.scaling_factor).sample
            ev[2].record()
            ev[2].synchronize()
            t_dit += ev[0].elapsed_time(ev[1])
            t_all += ev[0].elapsed_time(ev[2])
    t_dit, t_all = t_dit / a.iters, t_all / a.iters
    f_dit = fwd_flops_per_sample((lat // 2) ** 2, L=L) * B
    f_vae = sum(e[2] for e in decode_schedule(B, 512))             # 2 m n k of every convolution / projection / attention product of the decoder, unpadded
    print(json.dumps({"workload": "config5: PixArt-alpha-DMD 512px one-step generator + SD-VAE decode", "batch": B, "ms_per_batch": t_all,
                      "images_per_s": B / t_all * 1e3, "ms_dit": t_dit, "ms_vae_decode": t_all - t_dit, "TFLOP_dit": f_dit / 1e12, "TFLOP_vae": f_vae / 1e12,
                      "TFLOP/s": (f_dit + f_vae) / t_all / 1e9, "mfma_frac": (f_dit + f_vae) / t_all / 1e9 / (MFMA_PEAK / 1e12),
                      "finite": bool(torch.isfinite(img).all()), "image_shape": list(img.shape)}))


if __name__ == "__main__":
    main()
