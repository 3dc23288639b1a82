#!/usr/bin/env python
"""Headline benchmark: denoising TRAINING steps/sec (fwd + bwd + grad-clip + AdamW) of PixArt-Sigma-XL/2 at 1024px,
batch 16 per GPU, data-parallel over N GPUs of one node (BASELINE.json `metric`, configs[2]).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One JSON line on rank 0.  A "step" = q_sample -> PixArtMS.forward -> IDDPM loss (MSE + VB) -> backward (gradient all-reduce
overlapped) -> global-norm clip -> AdamW, on synthetic latents / caption features already resident in HBM, random-init
weights with the zero-init tensors re-randomised.  `value` is whole-job steps/s x nothing: every
rank does one step of batch 16, so steps/s of the job = 1 / (max-over-ranks step time) and images/s = 16 N x that.

Extra objects: `roofline` (MFMA bound; algorithmic FLOPs of SURVEY.md section 8d / time; the dominant kernel measured live with
events on the launch stream) and `cpu_baseline` (oracle/ = CPU port of the reference, timed on the host cores on a
bounded sample).
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # RCCL across processes needs dmabuf IPC on this driver; must be set before the HIP runtime starts

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

D, H, DFF, DEPTH, LTXT = 1152, 16, 4608, 28, 300
MFMA_PEAK = 2.5e15   # dense bf16, MI355X (MI355X_MICROARCH.md)
def _pmc_files():
    """profiles/*_pmc_attention.json, newest session first (tools/pmc_query.py --json; name order: round, then `final` over numbered / lettered sessions)."""
    import glob
    import re

    def key(path):
        m = re.match(r"r0?(\d+)(final|[a-z]?)_?(\d*)", os.path.basename(path))
        return (int(m.group(1)), 1 if m.group(2) == "final" else 0, m.group(2), int(m.group(3) or 0)) if m else (0, 0, "", 0)
    return sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_attention.json")), key=key, reverse=True)
DOMINANT = "attn_bwd_dkv4_kernel"     # the step's largest kernel by total time (profiles/r4final_step_kernel_stats.csv: 28 self-attention launches; round 4: the
                                      # one-wave-per-SIMD dK/dV kernel, 256 keys per workgroup - attn_bwd_dkv2_kernel<1> until round 3; its 16-row variant dkv5 is
                                      # faster alone and slower in the step, profiles/r4_34_step_ab_attention.txt)


def pmc_traffic(kernel_substr, grid):
    """HBM bytes per launch of a kernel from the committed PMC passes (rocprofv3 cannot run inside the benchmark): separate --pmc
    passes for FETCH_SIZE / WRITE_SIZE (KB per launch); FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md.
    Returns (bytes | None, source)."""
    for path in _pmc_files():
        rel = os.path.relpath(path, ROOT)
        with open(path) as f:
            rows = json.load(f)["kernels"]
        head = ""
        txt = path[:-5] + ".txt"                       # the session's text artefact starts with "# box .. HEAD .. operand build .."
        if os.path.exists(txt):
            with open(txt) as f:
                first = f.readline().strip()
            head = " [" + first.lstrip("# ") + "]" if first.startswith("#") else ""
        for r in rows:
            if kernel_substr in r["kernel"] and r.get("grid") in grid and "FETCH_SIZE" in r["counters"] and "WRITE_SIZE" in r["counters"]:
                return 2 * r["counters"]["FETCH_SIZE"] * 1e3 + r["counters"]["WRITE_SIZE"] * 1e3, f"{rel}{head}: 2 x FETCH_SIZE (gfx950 correction) + WRITE_SIZE, separate --pmc passes, bytes per launch (a committed file, not measured by this run)"
    return None, "no PMC file for this kernel / grid under profiles/"


def fwd_flops_per_sample(N, L=LTXT, n_kv=None):
    """SURVEY.md section 8(d) algorithmic FLOPs (2mnk per GEMM; softmax/LN/GELU excluded)."""
    n_kv = N if n_kv is None else n_kv
    D2 = D * D
    per_layer = 28 * N * D2 + 4 * L * D2 + 4 * N * n_kv * D + 4 * N * L * D
    embed = 2 * N * 16 * D + 2 * 256 * D + 2 * D2 + 12 * D2 + 2 * L * 4096 * D + 2 * L * D2 + 2 * N * D * 32
    return DEPTH * per_layer + embed


def _pad(n, m):
    return (n + m - 1) // m * m


def issued_flops_per_step(B, N, L=LTXT):
    """Matrix FLOPs the step's kernels ISSUE on the MFMA pipe, computed from the launch shapes (not the algorithmic count `roofline.achieved` uses): what
    a power-limited matrix pipe has to get through whatever surrounds it.  Differences from the algorithmic 2mnk:
      * attention backward is two kernels that each recompute S and dP: dQ issues S, dP, dQ; dK/dV issues S, dP, dV, dK  (7 products for the 5 of a
        single-kernel backward; the forward issues its 2);
      * head_dim 72 pads to 80 wherever it is the reduction (S, dP: 5 k-steps of 16) or an output on 16-row tiles (O, dQ), and to 96 where it is an
        output on 32-row tiles (dV, dK of the default dK/dV kernel, attn_bwd_dkv4_kernel);
      * text keys pad to whole 64-key tiles in the keys-resident cross-attention kernels (300 -> 320) and to whole 128-key workgroups in the
        cross-attention dK/dV kernel (300 -> 384);
      * token GEMMs: M = B N is a multiple of 256 and every width a multiple of 128, so NT / NN issue exactly 2mnk; the weight-gradient (TN) kernel pads an
        output side of 1152 to 1280 on the m side (5 tiles of 256; the n side is paired half tiles) - and the 4,800 text rows pad to 4,864."""
    R, hd = B * N, 72
    pr = lambda m, n, k: 2.0 * m * n * k                     # noqa: E731
    # ---- token GEMMs per block: forward NT, dX NN, dW TN (m = output features of the layer, padded to 256; n = its input features)
    lin = [(3 * D, D), (D, D), (D, D), (D, D), (DFF, D), (D, DFF)]     # (out, in): qkv, proj, q_linear, cross proj, fc1, fc2 on R rows
    g = 0.0
    for o, i in lin:
        g += 2 * pr(R, o, i) + pr(_pad(o, 256), i, R)
    Lt = B * L                                                         # packed text rows: kv_linear (forward, dX into the caption gradient, dW)
    g += pr(_pad(Lt, 256), 2 * D, D) * 2 + pr(_pad(2 * D, 256), D, _pad(Lt, 64))
    # ---- self-attention per block and head: 2 N^2 x (sum of the padded depth / width of each product)
    sa = 2.0 * N * N * ((80 + 80) + (80 + 80 + 80) + (80 + 80 + 96 + 96))
    # ---- cross-attention per block and head: forward / dQ on 64-key tiles, dK/dV on 128-key workgroups
    ca = 2.0 * N * (_pad(L, 64) * ((80 + 80) + (80 + 80 + 80)) + _pad(L, 128) * (80 + 80 + 96 + 96))
    per_block = g + B * H * (sa + ca)
    # ---- outside the blocks: caption MLP (4096 -> 1152 -> 1152 on the text rows), final linear, patch embed: forward + dX + dW
    embed = 3 * (pr(_pad(Lt, 256), D, 4096) + pr(_pad(Lt, 256), D, D) + pr(R, 128, D))
    return DEPTH * per_block + embed


def mfma_only_rate(seconds=2.0):
    """Live: the library's MFMA-only probe (include/pixart_hip.h: pxa_mfma_rate_probe - 32 x 32 x 16 MFMAs on N(0,1) operands of the build's type, one wave per
    SIMD, nothing else in the loop) run back to back for `seconds`, the rate taken over the SECOND half (the clock has settled under the power limit by then)."""
    import ctypes
    from pixart_sigma_amd import lib as L_, ops
    lib = L_.load()
    n = lib.pxa_mfma_rate_probe_bytes()
    buf = torch.randn(n // 2, device="cuda").to(ops.BF16)
    sink = torch.zeros(1, device="cuda")
    fl = ctypes.c_double(0.0)
    out = {}
    for shape in (32, 16):
        iters = 4096
        launch = lambda: L_.check(lib.pxa_mfma_rate_probe(L_.ptr(buf), shape, iters, L_.ptr(sink), ctypes.byref(fl), L_.stream()), "pxa_mfma_rate_probe")  # noqa: E731
        t1 = timed(launch, 3, warm=1)                        # size the run: launches for ~seconds/2 per half
        k = max(4, int(seconds / 2 / t1))
        timed(launch, k, warm=0)                             # first half: untimed ramp
        t = timed(launch, k, warm=0)
        out[shape] = fl.value / t
    return out


def timed(fn, iters, warm=2):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def kernel_rooflines(B, N):
    """Live per-kernel measurements (events on the stream the kernels are launched on = torch's current stream).
    >= 15-40 launches each: runs of a few launches under-read by up to 20 % on this part (clock ramp)."""
    from pixart_sigma_amd import ops
    dev = "cuda"
    R = B * N
    x = torch.randn(R, D, device=dev).to(ops.BF16)
    w1 = (torch.randn(DFF, D, device=dev) * D ** -0.5).to(ops.BF16)
    b1 = torch.zeros(DFF, device=dev)
    out, out2 = torch.empty(R, DFF, dtype=ops.BF16, device=dev), torch.empty(R, DFF, dtype=ops.BF16, device=dev)
    res = {}
    t = timed(lambda: ops.gemm(x, w1, ops.NT, bias=b1, act=ops.ACT_GELU, out=out, out2=out2), 40, warm=5)
    res["gemm_nt_fc1_gelu"] = dict(flops=2.0 * R * DFF * D, seconds=t)
    dy = torch.randn(R, DFF, device=dev).to(ops.BF16)
    dxo = torch.empty(R, D, dtype=ops.BF16, device=dev)
    t = timed(lambda: ops.gemm(dy, w1, ops.NN, out=dxo), 40, warm=5)
    res["gemm_nn_fc1_dx"] = dict(flops=2.0 * R * DFF * D, seconds=t)
    dw = torch.zeros(DFF, D, device=dev)
    t = timed(lambda: ops.gemm(dy, x, ops.TN, out_f32=dw, accumulate=True, split_k=0), 40, warm=5)
    res["gemm_tn_fc1_dw"] = dict(flops=2.0 * R * DFF * D, seconds=t)
    # the self-attention operands as the step presents them: q carries scale * log2 e (engine.py folds it into the qkv projection: pxa_attn_args.q_prescaled)
    qkv32 = torch.randn(R, 3 * D, device=dev)
    qkv32[:, :D] *= ops.Q_PRESCALE
    qkv = qkv32.to(ops.BF16)
    del qkv32
    pre = dict(q_prescaled=True)
    a = torch.empty(R, D, dtype=ops.BF16, device=dev)
    lse = torch.empty(B, H, N, device=dev)
    s3 = (N * 3 * D, 3 * D, 72)
    st = (s3, s3, s3, (N * D, D, 72))
    t = timed(lambda: ops.attention_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], a, lse, B, H, N, N, st, **pre), 30, warm=5)
    res["attn_fwd_self"] = dict(flops=4.0 * B * N * N * D, seconds=t)
    da, dqkv, delta = torch.randn(R, D, device=dev).to(ops.BF16), torch.empty_like(qkv), torch.empty(B, H, N, device=dev)
    t = timed(lambda: ops.attention_bwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], a, da, lse, delta, dqkv[:, :D], dqkv[:, D:2 * D],
                                        dqkv[:, 2 * D:], B, H, N, N, st, (s3, s3, s3), **pre), 15, warm=3)
    res["attn_bwd_self"] = dict(flops=10.0 * B * N * N * D, seconds=t)     # delta + dQ + dK/dV kernels, algorithmic 2.5x forward
    # The dominant kernel by itself, one event pair around EACH launch (no subtraction).  PXA_ATTN_BWD_NO_PREPASS makes
    # pxa_attn_bwd skip its delta / stats pre-pass - the workspace still holds this input's rows from the call above - and dq = NULL skips the dQ
    # kernel, so each call is exactly one attn_bwd_dkv4_kernel launch.  Its contract needs S, dP, dV, dK = 4 of the 2 N^2 d products.
    os.environ["PXA_ATTN_BWD_NO_PREPASS"] = "1"
    try:
        one = lambda: ops.attention_bwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], a, da, lse, delta, None, dqkv[:, D:2 * D], dqkv[:, 2 * D:],  # noqa: E731
                                        B, H, N, N, st, (s3, s3, s3), **pre)
        for _ in range(5):
            one()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
        for e0, e1 in evs:
            e0.record()
            one()
            e1.record()
        torch.cuda.synchronize()
        ts = sorted(e0.elapsed_time(e1) * 1e-3 for e0, e1 in evs)
    finally:
        del os.environ["PXA_ATTN_BWD_NO_PREPASS"]
    res["attn_bwd_dkv_kernel"] = dict(flops=8.0 * B * N * N * D, seconds=sum(ts) / len(ts), min_s=ts[0], max_s=ts[-1], launches=len(ts))
    for k, v in res.items():
        v["tflops"] = v["flops"] / v["seconds"] / 1e12
        v["frac"] = v["flops"] / v["seconds"] / MFMA_PEAK
    return res


def cpu_baseline(px=1024, batches=(1, 2)):
    """oracle/ (CPU port of the reference path, fp32, attention through torch's CPU scaled_dot_product_attention) fwd+bwd of the
    FULL-DEPTH XL/2 on the host cores at the benchmark's own resolution: one REAL batch-1 and one REAL batch-2 step,
    ~60 s of CPU work together.  The batch-16 step is extrapolated from the batch-2 point (x8: samples are independent - no cross-sample op - and the
    batch-2 / batch-1 ratio, reported, shows how linear the port is).  Threads are capped at 32: on the 256-thread GPU host an uncapped torch pool ran this op
    mix ~40x slower (measured round 1)."""
    from oracle import pixart_oracle as po
    from oracle.weights import make_inputs, make_state_dict
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    po.SDPA = True
    lat = px // 8
    cfg = po.OracleCfg(depth=DEPTH, input_size=lat, model_max_length=LTXT, pe_interpolation=px / 512)
    sd = make_state_dict(cfg, seed=0)
    sd = {k: (v.requires_grad_(True) if k != "y_embedder.y_embedding" else v) for k, v in sd.items()}
    diff = po.GaussianDiffusionOracle()
    times = {}
    for nb in batches:
        inp = make_inputs(B=nb, Hl=lat, Wl=lat, L=LTXT, seed=1)
        for v in sd.values():
            v.grad = None
        t0 = time.time()
        terms = diff.training_losses(lambda xt, t: po.forward(sd, cfg, xt, t, inp["y"], inp["mask"]), inp["x"], inp["t"], inp["noise"])
        terms["loss"].mean().backward()
        times[nb] = time.time() - t0
    po.SDPA = False
    nb = max(batches)
    per_sample = times[nb] / nb
    desc = (f"oracle (CPU port of the reference path, fp32, SDPA attention) fwd+bwd of the full-depth XL/2 at {px}px on {cores} threads: "
            + ", ".join(f"batch {b}: {t:.1f} s" for b, t in times.items()) + f" (measured, not extrapolated); batch-{nb} time per sample {per_sample:.1f} s")
    return per_sample, cores, desc, times


def torch_rocm_baseline(B, lat, px, steps=3, warmup=1):
    """The 'reference PyTorch path on the same GPU' bar: the restated reference modules
    (oracle/, same arithmetic as the reference's nn.Modules) on this MI355X through stock PyTorch-ROCm - bf16 autocast (hipBLASLt
    linears), F.scaled_dot_product_attention, torch.optim.AdamW(fused), clip_grad_norm_ - on the same synthetic training step.
    Falls back to per-block activation checkpointing (the reference's own 1024px setting, configs/...img1024_internalms.py) if the
    stored activations do not fit."""
    from torch.utils.checkpoint import checkpoint
    from oracle import pixart_oracle as po
    from oracle.weights import make_state_dict
    dev = torch.device("cuda")
    po.SDPA = True
    cfg = po.OracleCfg(depth=DEPTH, input_size=lat, model_max_length=LTXT, pe_interpolation=px / 512)
    sd = {k: v.to(dev) for k, v in make_state_dict(cfg, seed=0).items()}
    params = [v.requires_grad_(True) for k, v in sd.items() if k != "y_embedder.y_embedding"]
    opt = torch.optim.AdamW(params, lr=2e-5, weight_decay=3e-2, eps=1e-10, fused=True)
    diff = po.GaussianDiffusionOracle()
    g = torch.Generator(device="cpu").manual_seed(1234)
    x0, noise = torch.randn(B, 4, lat, lat, generator=g).to(dev), torch.randn(B, 4, lat, lat, generator=g).to(dev)
    y, t = torch.randn(B, 1, LTXT, 4096, generator=g).to(dev), torch.randint(0, 1000, (B,), generator=g).to(dev)
    mask = torch.ones(B, LTXT, dtype=torch.int64, device=dev)
    mode = {"ckpt": False}
    orig_block = po.block_forward

    def block_ckpt(sd_, i, x, yp, t0, yl, HW, cfg_, rp=False):
        return checkpoint(lambda xx, yy, tt: orig_block(sd_, i, xx, yy, tt, yl, HW, cfg_, rp), x, yp, t0, use_reentrant=False)

    def step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            terms = diff.training_losses(lambda xt, tt: po.forward(sd, cfg, xt, tt, y, mask).float(), x0, t, noise)
        terms["loss"].mean().backward()
        torch.nn.utils.clip_grad_norm_(params, 0.01)
        opt.step()

    try:
        try:
            for _ in range(warmup):
                step()
        except torch.OutOfMemoryError:
            torch.cuda.empty_cache()
            mode["ckpt"] = True
            po.block_forward = block_ckpt
            for _ in range(warmup):
                step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
    finally:
        po.block_forward = orig_block
        po.SDPA = False
    del opt, params, sd
    torch.cuda.empty_cache()
    return dt, mode["ckpt"]


def baseline_configs(timeout=900):
    """BASELINE.json configs[1], [3], [4] on this box, each in its own process AFTER the headline's timed region and legs: the 20-step
    DPM-Solver++ CFG-4.5 samplers at 512px / batch 8 and 2K / batch 2 with KV compression (tools/bench_infer.py) and the DMD one-step generator + SD-VAE decode at
    batch 64 (tools/bench_dmd.py), fp16 operands - the reference's inference dtype (scripts/inference.py:188-196).  Parity of exactly these chains:
    tests/test_model_gpu.py (dpms_xl2_512_s20, dpms_xl2_2k_kv_s4, dmd_xl2_512_l120 goldens from the live reference)."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("PXA_OPERAND_DTYPE", "PXA_LIB_PATH", "RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["PXA_OPERAND_DTYPE"] = "f16"
    out = {}

    def lines(cmd):
        r = subprocess.run([sys.executable] + cmd, capture_output=True, text=True, env=env, timeout=timeout, cwd=ROOT)
        js = []
        for ln in r.stdout.splitlines():
            if ln.startswith("{"):
                try:
                    js.append(json.loads(ln))
                except ValueError:
                    pass
        if not js:
            raise RuntimeError((r.stderr or r.stdout)[-300:])
        return js
    try:
        for key, j in zip(("c2", "c4"), lines([os.path.join(ROOT, "tools", "bench_infer.py"), "both"])):
            out[key] = {"workload": j["workload"], "seconds": round(j["seconds"], 4), "unit": "s per batch of images (20 denoiser evaluations)", "images_per_s": round(j["images_per_s"], 3),
                        "ms_per_nfe": round(j["ms_per_nfe"], 3), "TFLOP/s": round(j["TFLOP/s"], 1), "frac": round(j["mfma_frac"], 4), "dtype": "fp16"}
    except Exception as e:   # noqa: BLE001 - a side leg must never take the headline number down
        out["c2_c4_error"] = f"{type(e).__name__}: {e}"[:300]
    try:
        j = lines([os.path.join(ROOT, "tools", "bench_dmd.py")])[-1]
        out["c5"] = {"workload": j["workload"], "ms_per_batch": round(j["ms_per_batch"], 2), "images_per_s": round(j["images_per_s"], 2), "ms_dit": round(j["ms_dit"], 2),
                     "ms_vae_decode": round(j["ms_vae_decode"], 2), "TFLOP/s": round(j["TFLOP/s"], 1), "frac": round(j["mfma_frac"], 4),
                     "frac_dit": round(j["TFLOP_dit"] * 1e12 / (j["ms_dit"] * 1e-3) / MFMA_PEAK, 4),
                     "frac_vae": round(j["TFLOP_vae"] * 1e12 / (j["ms_vae_decode"] * 1e-3) / MFMA_PEAK, 4), "dtype": "fp16"}
    except Exception as e:   # noqa: BLE001
        out["c5_error"] = f"{type(e).__name__}: {e}"[:300]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=16, help="per-GPU batch (BASELINE: 16)")
    ap.add_argument("--image-size", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-roofline", action="store_true")
    ap.add_argument("--grad-checkpoint", action="store_true")
    ap.add_argument("--dtype", choices=["bf16", "fp16"], default="fp16",
                    help="MFMA operand type.  fp16 (default since round 3) = the reference's own mixed precision (configs/PixArt_xl2_internal.py:57: fp16 + "
                         "GradScaler) with dynamic loss scaling on the device - the build whose parity tests assert BASELINE.json's <= 1e-3; bf16 = the "
                         "scaler-free build (one bf16 rounding alone is 1.6e-3: it cannot meet that bound).  One library per operand type and process; "
                         "at N = 1 the other build is timed in a subprocess and reported under `other_dtype`.")
    ap.add_argument("--ragged-text", action="store_true",
                    help="SURVEY.md section 8d secondary: caption lengths drawn from 12..300 per sample (packed varlen cross-attention) instead of all 300")
    ap.add_argument("--no-other-dtype", action="store_true", help="skip the subprocess run of the other operand build (N = 1 only)")
    ap.add_argument("--no-torch-baseline", action="store_true", help="skip the stock-PyTorch-ROCm leg (N = 1 only)")
    ap.add_argument("--no-configs", action="store_true", help="skip the BASELINE configs 2 / 4 / 5 leg (N = 1 only; subprocesses after the timed region)")
    ap.add_argument("--optimizer", choices=["adamw", "came"], default="adamw",
                    help="adamw = the BASELINE config (configs/PixArt_xl2_internal.py); came = the CAMEWrapper of the Sigma configs")
    a = ap.parse_args()
    if a.dtype == "fp16":                      # must be decided before pixart_sigma_amd is imported: one library per operand type
        os.environ["PXA_OPERAND_DTYPE"] = "f16"

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_pg = world > 1 or "RANK" in os.environ            # under torch.distributed.run a one-rank job still builds its RCCL group
    if use_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from pixart_sigma_amd import IDDPM, PixArtMS_XL_2
    from pixart_sigma_amd.dp import FusedAdamW, FusedCAME, LossScaler
    from pixart_sigma_amd.model.utils import set_grad_checkpoint

    lat = a.image_size // 8
    N = (lat // 2) ** 2
    B = a.batch
    torch.manual_seed(0)     # identical init on every rank; model.prepare() below ALSO broadcasts rank 0's weights when a process group exists (DDP semantics)
    model = PixArtMS_XL_2(input_size=lat, pe_interpolation=a.image_size / 512, model_max_length=LTXT, class_dropout_prob=0.0)
    with torch.no_grad():    # re-randomise the zero-init tensors so no branch is numerically dead
        for blk in model.blocks:
            blk.cross_attn.proj.weight.normal_(std=0.02)
        model.final_layer.linear.weight.normal_(std=0.02)
    model = model.to(dev).train()
    if a.grad_checkpoint:
        set_grad_checkpoint(model)
    model.prepare(dev)
    scaler = LossScaler(dev) if a.dtype == "fp16" else None      # GradScaler protocol on the device (accelerate's fp16 mode)
    if a.optimizer == "came":
        opt = FusedCAME(model, lr=2e-5, weight_decay=0.0, betas=(0.9, 0.999, 0.9999), eps=(1e-30, 1e-16), max_grad_norm=0.01, scaler=scaler)
    else:
        opt = FusedAdamW(model, lr=2e-5, weight_decay=3e-2, eps=1e-10, max_grad_norm=0.01, scaler=scaler)
    diff = IDDPM(str(1000), learn_sigma=True, pred_sigma=True, snr=False)

    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    x0 = torch.randn(B, 4, lat, lat, generator=g).to(dev)
    noise = torch.randn(B, 4, lat, lat, generator=g).to(dev)
    y = torch.randn(B, 1, LTXT, 4096, generator=g).to(dev)
    t = torch.randint(0, 1000, (B,), generator=g).to(dev)
    mask = torch.ones(B, LTXT, dtype=torch.int64)            # host-side mask: no device sync for y_lens
    if a.ragged_text:
        lens = torch.randint(12, LTXT + 1, (B,), generator=g)
        lens[0] = LTXT
        for i, n in enumerate(lens.tolist()):
            mask[i, n:] = 0

    def step():
        opt.zero_grad()
        terms = diff.training_losses(model, x0, t, model_kwargs=dict(y=y, mask=mask, data_info=None), noise=noise)
        loss = terms["loss"].mean()
        (scaler.scale(loss) if scaler is not None else loss).backward()
        opt.step()
        return loss

    def barrier():
        if use_pg:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    barrier()
    dt = time.perf_counter() - t0
    if use_pg:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
    sec_per_step = dt / a.steps
    loss_v = float(loss.item())

    if rank == 0:
        flops_step = 3 * fwd_flops_per_sample(N) * B          # per GPU, fwd + bwd, no recompute, no optimizer
        out = {
            "metric": "denoising steps/sec (fwd+bwd) PixArt-Sigma-XL/2 1024px bs16 @1/2/4/8 GPU",
            # whole-job aggregate: batch-16 denoising steps completed per second by ALL ranks (every rank runs one per iteration; the optimizer step of
            # the job is one per iteration: `optimizer_steps_per_s`)
            "value": world / sec_per_step, "value_definition": "batch-16 denoising steps per second summed over all ranks (= n_gpus x optimizer steps per second)",
            "optimizer_steps_per_s": 1.0 / sec_per_step, "unit": "steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": sec_per_step * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": f"PixArt-Sigma-XL/2 {a.image_size}px training step (fwd+bwd+clip+{'CAME' if a.optimizer == 'came' else 'AdamW'}), batch {B}/GPU, L=300 text tokens, "
                                   f"DP={world} RCCL all-reduce", "model": "PixArtMS_XL_2", "global_batch": B * world, "seq_len": N,
                       "parallelism": f"dp{world}", "grad_checkpoint": bool(a.grad_checkpoint), "optimizer": a.optimizer,
                       "text_lens": "ragged 12..300" if a.ragged_text else "all 300"},
            "steps_per_s_per_gpu": 1.0 / sec_per_step, "images_per_s": B * world / sec_per_step, "final_loss": loss_v, "process_group": "nccl" if use_pg else None,
            **({"loss_scale": scaler.value, "steps_skipped": scaler.steps_skipped} if scaler is not None else {}),
            "step_tflops_per_gpu": flops_step / sec_per_step / 1e12,
        }
        roof = {"bound": "mfma", "achieved": flops_step / sec_per_step / 1e12, "peak": MFMA_PEAK / 1e12, "unit": "TFLOP/s",
                "frac": flops_step / sec_per_step / MFMA_PEAK, "traffic": None, "scope": "whole training step (algorithmic FLOPs / wall time)"}
        if not a.no_kernel_roofline and a.image_size == 1024 and world == 1:   # per-kernel measurements and the CPU leg: N = 1 only
            ks = kernel_rooflines(B, N)
            # dominant kernel of the step by total time (profiles/: attn_bwd_dkv_kernel, 56 launches, ~17 % of the step)
            dom = ks["attn_bwd_dkv_kernel"]
            traffic, tsrc = pmc_traffic(DOMINANT, ((N // 256) * H * B,))   # workgroups of the self-attention launch (flat grid, 256 keys per workgroup)
            roof = {"bound": "mfma", "kernel": DOMINANT + " (dK/dV of self-attention, B16 H16 N4096 d72)", "achieved": dom["tflops"], "peak": MFMA_PEAK / 1e12,
                    "unit": "TFLOP/s", "frac": dom["frac"], "traffic": traffic, "traffic_source": tsrc,
                    "flops_per_launch": dom["flops"], "ms_per_launch": dom["seconds"] * 1e3,
                    "timing": f"HIP event pair around each of {dom['launches']} single launches on the launch stream (min {dom['min_s'] * 1e3:.3f} / max {dom['max_s'] * 1e3:.3f} ms)",
                    "step": {"achieved": flops_step / sec_per_step / 1e12, "frac": flops_step / sec_per_step / MFMA_PEAK,
                             "scope": "whole training step (algorithmic FLOPs / wall time)"},
                    "kernels": {k: {"TFLOP/s": round(v["tflops"], 1), "frac": round(v["frac"], 4), "ms": round(v["seconds"] * 1e3, 3)} for k, v in ks.items()}}
        # the ceiling argument, driver-visible: FLOPs the step issues (from the launch shapes) and the rate an MFMA-only stream sustains
        # on THIS box under its power limit, measured now; `peak` stays the sheet's 2.5 PFLOP/s
        issued = issued_flops_per_step(B, N)
        roof["issued_flops_per_step"] = issued
        roof["issued_over_algorithmic"] = issued / flops_step
        if not a.no_kernel_roofline and a.image_size == 1024 and world == 1:
            try:
                mr = mfma_only_rate()
                roof["mfma_only_rate"] = {"TFLOP/s": round(mr[32] / 1e12, 1), "frac_of_peak": round(mr[32] / MFMA_PEAK, 4),
                                          "TFLOP/s_16x16x32": round(mr[16] / 1e12, 1),
                                          "what": "pxa_mfma_rate_probe: v_mfma_f32_32x32x16 only, N(0,1) operands in registers, one wave per SIMD on every CU, "
                                                  "2 s back to back, rate of the second half (second figure: the same FLOPs as 16x16x32)"}
                roof["issued_time_floor_ms"] = issued / mr[32] * 1e3          # the issued matrix work alone at that rate
                roof["step_frac_of_mfma_only_rate"] = (issued / sec_per_step) / mr[32]     # pipe OCCUPANCY: issued FLOPs include head_dim / key padding and the
                #   S / dP recomputation of the two backward kernels - it rises if the kernels pad more; the EFFICIENCY figure is the next one
                roof["step_algorithmic_frac_of_mfma_only_rate"] = (flops_step / sec_per_step) / mr[32]
            except Exception as e:   # noqa: BLE001 - a measurement leg must never take the headline number down
                roof["mfma_only_rate"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        out["roofline"] = roof
        if getattr(opt.reducer, "last_trace", None) is not None:     # PXA_DP_TRACE=1: the last step's per-bucket record (ready / passed / exposed communication)
            out["dp_trace"] = opt.reducer.last_trace
        if world == 1 and not (a.no_other_dtype and a.no_torch_baseline and a.no_configs):
            del opt
            model._store = model._engine = None
            del model
            torch.cuda.empty_cache()
        if world == 1 and not a.no_other_dtype:
            # the other operand build on the same box, same step, same K / W, as its own process (one library per operand type and process)
            import subprocess
            other = "bf16" if a.dtype == "fp16" else "fp16"
            cmd = [sys.executable, os.path.abspath(__file__), "--dtype", other, "--steps", str(a.steps), "--warmup", str(a.warmup), "--batch", str(B),
                   "--image-size", str(a.image_size), "--optimizer", a.optimizer, "--no-other-dtype", "--no-cpu-baseline", "--no-torch-baseline", "--no-kernel-roofline", "--no-configs"]
            if a.grad_checkpoint:
                cmd.append("--grad-checkpoint")
            env = {k: v for k, v in os.environ.items() if k not in ("PXA_OPERAND_DTYPE", "PXA_LIB_PATH", "RANK", "WORLD_SIZE", "LOCAL_RANK")}
            try:
                r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
                oj = json.loads(r.stdout.strip().splitlines()[-1])
                out["other_dtype"] = {k: oj[k] for k in ("dtype", "value", "unit", "ms_per_step", "steps", "warmup", "final_loss", "step_tflops_per_gpu") if k in oj}
                out["other_dtype"].update({k: oj[k] for k in ("loss_scale", "steps_skipped") if k in oj})
                out["other_dtype"]["frac"] = oj["roofline"]["frac"]
            except Exception as e:   # noqa: BLE001
                out["other_dtype"] = {"dtype": other, "value": None, "error": f"{type(e).__name__}: {e}"[:300]}
        if not a.no_torch_baseline and world == 1:
            try:
                tdt, ckpt = torch_rocm_baseline(B, lat, a.image_size)
                out["torch_rocm_baseline"] = {"value": 1.0 / tdt, "unit": "steps/s", "ms_per_step": tdt * 1e3, "speedup_of_this_repo": tdt / sec_per_step,
                                              "what": "restated reference modules (oracle/) on this GPU through stock PyTorch-ROCm: bf16 autocast linears (hipBLASLt), "
                                                      "F.scaled_dot_product_attention, fused torch AdamW, clip_grad_norm_; same synthetic step, batch "
                                                      f"{B}" + ("; per-block activation checkpointing (activations did not fit)" if ckpt else "; no recompute")}
            except Exception as e:   # noqa: BLE001 - the baseline leg must never take the headline number down with it
                out["torch_rocm_baseline"] = {"value": None, "error": f"{type(e).__name__}: {e}"[:300]}
        if not a.no_configs and world == 1:
            out["configs"] = baseline_configs()
        if not a.no_cpu_baseline and world == 1:
            cdt, cores, desc, ctimes = cpu_baseline(a.image_size)
            out["cpu_baseline"] = {"value": 1.0 / (cdt * B), "unit": "steps/s", "cores": cores, "kind": "port",
                                   "sample": desc + f"; x{B} per sample to the {a.image_size}px batch-{B} step (samples are independent)",
                                   "measured_seconds": {f"batch_{b}": round(t, 2) for b, t in ctimes.items()}}
        print(json.dumps(out))
    if use_pg:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
