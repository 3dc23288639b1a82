"""Inference workloads of BASELINE.json on one MI355X (parity-tested elsewhere; this measures them):
  config 2: PixArt-Sigma-XL/2 512px, batch 8, 20-step DPM-Solver++ with CFG 4.5 (model batch 16)
  config 4: PixArt-Sigma-XL/2 2K multi-scale, KV compression (conv, x2) on layers 14-27, batch 2 (model batch 4), 20 steps
Usage: python tools/bench_infer.py [512|2k|both] [--steps 20]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import MFMA_PEAK, fwd_flops_per_sample  # noqa: E402
from pixart_sigma_amd import DPMS, PixArtMS_XL_2  # noqa: E402


def run(name, image_size, bs, steps, kv, graph=False):
    lat = image_size // 8
    N = (lat // 2) ** 2
    kvc = {"sampling": "conv", "scale_factor": 2, "kv_compress_layer": list(range(14, 28))} if kv else None
    torch.manual_seed(0)
    m = PixArtMS_XL_2(input_size=lat, pe_interpolation=image_size / 512, model_max_length=300, kv_compress_config=kvc)
    with torch.no_grad():
        for blk in m.blocks:
            blk.cross_attn.proj.weight.normal_(std=0.02)
        m.final_layer.linear.weight.normal_(std=0.02)
    m = m.cuda().eval()
    g = torch.Generator().manual_seed(1)
    z = torch.randn(bs, 4, lat, lat, generator=g).cuda()
    y = torch.randn(bs, 1, 300, 4096, generator=g).cuda()
    null_y = torch.randn(1, 1, 300, 4096, generator=g).repeat(bs, 1, 1, 1).cuda()
    mask = torch.ones(bs, 300, dtype=torch.int64)      # host mask: no per-step sync

    solver = DPMS(m.forward_with_dpmsolver, condition=y, uncondition=null_y, cfg_scale=4.5, model_kwargs=dict(data_info=None, mask=mask))

    def sample():
        fn = solver.sample_graphed if graph else solver.sample
        return fn(z, steps=steps, order=2, skip_type="time_uniform", method="multistep")
    with torch.no_grad():
        ref = sample() if not graph else solver.sample(z, steps=steps, order=2, skip_type="time_uniform", method="multistep")
        first = sample()
        assert not graph or torch.equal(first, ref), "graph replay must reproduce the eager sample bit for bit"
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = sample()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    n_kv = None
    flops = 0.0
    for l in range(28):
        nk = N // 4 if (kv and l >= 14) else N
        flops += fwd_flops_per_sample(N, n_kv=nk) / 28      # per-layer share with that layer's key count
    flops_nfe = flops * 2 * bs
    res = {"workload": name + (" [HIP graph]" if graph else ""), "image_size": image_size, "batch": bs, "model_batch": 2 * bs, "tokens": N, "steps": steps,
           "seconds": dt, "denoising_steps_per_s": steps / dt, "images_per_s": bs / dt, "ms_per_nfe": dt / steps * 1e3,
           "TFLOP/s": flops_nfe * steps / dt / 1e12, "mfma_frac": flops_nfe * steps / dt / MFMA_PEAK, "finite": bool(torch.isfinite(out).all())}
    print(json.dumps(res))
    del m
    torch.cuda.empty_cache()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="?", default="both")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--graph", action="store_true", help="replay the sampling loop as one HIP graph (DPM_Solver.sample_graphed)")
    a = ap.parse_args()
    if a.what in ("256",):
        run("config1-like: XL/2 256px bs1 20-step DPM-Solver++ CFG", 256, 1, a.steps, kv=False, graph=a.graph)
    if a.what in ("512", "both"):
        run("config2: XL/2 512px bs8 20-step DPM-Solver++ CFG", 512, 8, a.steps, kv=False, graph=a.graph)
    if a.what in ("2k", "both"):
        run("config4: XL/2 2K bs2 KV-compress(14-27) 20-step DPM-Solver++ CFG", 2048, 2, a.steps, kv=True, graph=a.graph)
