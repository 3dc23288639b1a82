"""VAE decode / encode throughput on the HIP kernel set (config 5 of BASELINE.json is VAE-decode-bound: 512px, batch 64/GPU).
Usage (GPU box): python tools/bench_vae.py [--px 512] [--batch 16] [--iters 3] [--encode] [--cpu-sample]
Prints one JSON line: images/s, ms per batch, algorithmic TFLOP (2*m*n*k of every convolution / projection / attention product
at the UNPADDED sizes) and the achieved TFLOP/s; --cpu-sample times oracle/vae_ref.py (fp32, torch CPU) on one 256px image."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def conv_flops(model, px, decode=True):
    """Algorithmic FLOPs of one image through the decoder (or encoder) of the restated AutoencoderKL."""
    import torch.nn as nn
    total, hooks = [0.0], []

    def conv_hook(m, i, o):
        total[0] += 2.0 * o.shape[2] * o.shape[3] * o.shape[1] * m.weight[0].numel()

    def lin_hook(m, i, o):
        total[0] += 2.0 * o.shape[-2] * m.weight.numel()

    net = model.decoder if decode else model.encoder
    for m in list(net.modules()) + [model.post_quant_conv if decode else model.quant_conv]:
        if isinstance(m, nn.Conv2d):
            hooks.append(m.register_forward_hook(conv_hook))
        elif isinstance(m, nn.Linear):
            hooks.append(m.register_forward_hook(lin_hook))
    with torch.no_grad():
        if decode:
            model.decode(torch.zeros(1, 4, px // 8, px // 8, device="meta"))
        else:
            model.encode_moments(torch.zeros(1, 3, px, px, device="meta"))
    for h in hooks:
        h.remove()
    n, c = (px // 8) ** 2, 512
    return total[0] + 4.0 * n * n * c                      # + QK^T and PV of the mid-block attention


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--px", type=int, default=512)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--encode", action="store_true")
    ap.add_argument("--cpu-sample", action="store_true")
    a = ap.parse_args()
    from pixart_sigma_amd.vae import AutoencoderKL
    torch.manual_seed(0)
    vae = AutoencoderKL().cuda()                               # random-init weights (torch's default init; GroupNorm 1 / 0)
    if a.encode or a.cpu_sample:                               # the encoder's FLOP count and the CPU sample come from the restated reference (tools only)
        from oracle.vae_ref import AutoencoderKLRef, randomize_
        ref = randomize_(AutoencoderKLRef(), seed=0)
        vae.load_state_dict(ref.state_dict())
        fl = conv_flops(AutoencoderKLRef().to("meta"), a.px, decode=not a.encode)
    if not a.encode:
        from vae_layer_table import decode_schedule
        fl = sum(e[2] for e in decode_schedule(1, a.px))
    g = torch.Generator().manual_seed(0)
    if a.encode:
        x = torch.randn(a.batch, 3, a.px, a.px, generator=g).cuda()
        step = lambda: vae.encode(x).latent_dist.mean
    else:
        x = torch.randn(a.batch, 4, a.px // 8, a.px // 8, generator=g).cuda()
        step = lambda: vae.decode(x).sample
    step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        step()
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    out = {"op": "vae_encode" if a.encode else "vae_decode", "px": a.px, "batch": a.batch, "ms_per_batch": ms, "images_per_s": a.batch / ms * 1e3,
           "algorithmic_tflop_per_image": fl / 1e12, "achieved_tflops": fl * a.batch / ms / 1e9, "mfma_peak_frac": fl * a.batch / ms / 1e9 / 2500.0,
           "dtype": "bf16 storage / fp32 accumulate", "peak_mem_gb": torch.cuda.max_memory_allocated() / 2 ** 30}
    if a.cpu_sample:
        torch.set_num_threads(os.cpu_count())
        xs = torch.randn(1, 4, 32, 32, generator=g)
        with torch.no_grad():
            ref.decode(xs)
            t0 = time.time()
            ref.decode(xs)
            dt = time.time() - t0
        out["cpu_baseline"] = {"value": 1.0 / dt, "unit": "images/s (256px decode)", "cores": os.cpu_count(), "kind": "port", "sample": "1 x 256px decode, oracle/vae_ref.py fp32"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
