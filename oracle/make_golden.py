"""TEST INFRASTRUCTURE ONLY — generates tests/golden/*.pt by running the UNMODIFIED reference
(/root/reference, imported under oracle/ref_stubs.py) on seeded inputs and the deterministic
weights of oracle/weights.py.  Run in the build container (the reference does not exist on the
GPU box):   python -m oracle.make_golden [--only NAME]

Each fixture stores the case description (enough to rebuild weights+inputs from seeds) and the
reference outputs only, so files stay small.
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_stubs  # noqa: E402
from oracle.pixart_oracle import OracleCfg  # noqa: E402
from oracle.weights import make_inputs, make_state_dict  # noqa: E402

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

CASES = {
    # name: (cfg kwargs, input kwargs)
    "fwd_d2_sq": (dict(depth=2, input_size=16, model_max_length=20), dict(B=2, Hl=16, Wl=16, L=20, lens=[20, 7])),
    "fwd_d2_kvconv": (dict(depth=2, input_size=16, model_max_length=20, kv_sampling="conv", kv_scale_factor=2, kv_layers=(1,), pe_interpolation=1.0),
                      dict(B=2, Hl=16, Wl=24, L=20, lens=[13, 20])),
    "fwd_d2_kvuniform": (dict(depth=2, input_size=16, model_max_length=20, kv_sampling="uniform", kv_scale_factor=2, kv_layers=(0, 1)),
                         dict(B=2, Hl=16, Wl=16, L=20, lens=[20, 20])),
    "fwd_d2_kvave": (dict(depth=2, input_size=16, model_max_length=20, kv_sampling="ave", kv_scale_factor=2, kv_layers=(0, 1)),
                     dict(B=2, Hl=16, Wl=16, L=20, lens=[5, 20])),
    "fwd_d2_nomask": (dict(depth=2, input_size=16, model_max_length=20, pe_interpolation=0.5), dict(B=3, Hl=8, Wl=8, L=20, lens=None)),
    "train_d2": (dict(depth=2, input_size=16, model_max_length=20, kv_sampling="conv", kv_scale_factor=2, kv_layers=(1,)),
                 dict(B=2, Hl=16, Wl=16, L=20, lens=[20, 9])),
    "train_d2_plain": (dict(depth=2, input_size=16, model_max_length=20), dict(B=2, Hl=16, Wl=24, L=20, lens=[20, 9])),
    # qk_norm=True (off in every shipped config, but an option of AttentionKVCompress): forward with compressed K/V, and a training step
    "fwd_d2_qknorm": (dict(depth=2, input_size=16, model_max_length=20, qk_norm=True, kv_sampling="conv", kv_scale_factor=2, kv_layers=(1,)),
                      dict(B=2, Hl=16, Wl=16, L=20, lens=[20, 7])),
    "train_d2_qknorm": (dict(depth=2, input_size=16, model_max_length=20, qk_norm=True), dict(B=2, Hl=16, Wl=16, L=20, lens=[20, 9])),
    "dpms_d2": (dict(depth=2, input_size=16, model_max_length=20), dict(B=2, Hl=16, Wl=16, L=20, lens=[20, 11])),
    # ---- BASELINE.json configs[1..4] at their real token geometry, depth 2 (round 2) ----
    # configs[2]: 1024px training shapes: N = 4096 tokens, L = 300, pe_interpolation 2 (forward and the full training step)
    "fwd_1024_b2": (dict(depth=2, input_size=128, model_max_length=300, pe_interpolation=2.0), dict(B=2, Hl=128, Wl=128, L=300, lens=[300, 77])),
    "train_1024_b2": (dict(depth=2, input_size=128, model_max_length=300, pe_interpolation=2.0), dict(B=2, Hl=128, Wl=128, L=300, lens=[300, 131])),
    # configs[3]: 2K latent with KV compression (conv, x2) on one of the two blocks: N = 16384 -> N_kv = 4096 (scripts/inference.py:157-172,
    # configs/pixart_sigma_config/PixArt_sigma_xl2_img2K_internalms_kvcompress.py:44-49)
    "fwd_2k_kv": (dict(depth=2, input_size=256, model_max_length=300, pe_interpolation=4.0, kv_sampling="conv", kv_scale_factor=2, kv_layers=(1,)),
                  dict(B=1, Hl=256, Wl=256, L=300, lens=[211])),
    # configs[1] / configs[4]: 512px, N = 1024; L = 300 (Sigma) and L = 120 (alpha-DMD), multi-aspect latent for the second
    "fwd_512_l300": (dict(depth=2, input_size=64, model_max_length=300, pe_interpolation=1.0), dict(B=2, Hl=64, Wl=64, L=300, lens=[300, 12])),
    "fwd_512_l120": (dict(depth=2, input_size=64, model_max_length=120, pe_interpolation=1.0), dict(B=3, Hl=48, Wl=80, L=120, lens=[120, 120, 33])),
    # alpha-1024 micro-conditioning (SizeEmbedder on img_hw / aspect_ratio, PixArtMS.py:187-191): forward and a training step
    "fwd_d2_micro": (dict(depth=2, input_size=16, model_max_length=20, micro_condition=True), dict(B=2, Hl=16, Wl=24, L=20, lens=[20, 7])),
    "train_d2_micro": (dict(depth=2, input_size=16, model_max_length=20, micro_condition=True), dict(B=2, Hl=16, Wl=16, L=20, lens=[20, 9])),
    # forward_with_cfg (PixArtMS.py:221-234): batch = [cond half ; uncond half] over one latent half, guidance on 3 channels
    "cfg_d2": (dict(depth=2, input_size=16, model_max_length=20), dict(B=4, Hl=16, Wl=16, L=20, lens=[20, 9, 20, 20])),
    # scripts/inference.py:89-101 (--sampling_algo iddpm): IDDPM(str(steps)).p_sample_loop over forward_with_cfg, batch = [cond ; null] on a repeated
    # latent, clip_denoised=False as the script passes it (and True as the method's default); 5 respaced steps keep the CPU run short
    "iddpm_d2": (dict(depth=2, input_size=16, model_max_length=20), dict(B=4, Hl=16, Wl=16, L=20, lens=[20, 9, 20, 20])),
    # scripts/inference.py:119-133 (--sampling_algo sa-solver): SASolverSampler(...).sample(S, eta=1, CFG 4.5) - stochastic Adams predictor-corrector,
    # PEC, orders 2 / 2; 6 steps keep the CPU run short and still cover a deterministic (t > 0.8), three stochastic and a low-t deterministic step
    "sasolver_d2": (dict(depth=2, input_size=16, model_max_length=20), dict(B=2, Hl=16, Wl=16, L=20, lens=[20, 11])),
    # BASELINE.json configs[0]: XL/2 256px, batch 2, 2 DPM-Solver steps, CFG 4.5, random-init, CPU
    "cfg1_xl2_256": (dict(depth=28, input_size=32, model_max_length=300, pe_interpolation=0.5), dict(B=2, Hl=32, Wl=32, L=300, lens=[300, 77])),
    # ---- round 3: the model bench.py times, FULL DEPTH at the headline geometry (PixArtMS_XL_2, depth 28, PixArtMS.py:291-293; 1024px: N = 4096,
    # L = 300, pe_interpolation 2: configs/pixart_sigma_config/PixArt_sigma_xl2_img1024_internalms.py).  Batch 1 keeps the CPU run to minutes.
    "fwd_xl2_1024_b1": (dict(depth=28, input_size=128, model_max_length=300, pe_interpolation=2.0), dict(B=1, Hl=128, Wl=128, L=300, lens=[213])),
    "train_xl2_1024_b1": (dict(depth=28, input_size=128, model_max_length=300, pe_interpolation=2.0), dict(B=1, Hl=128, Wl=128, L=300, lens=[187])),
    # 'uniform_every' KV sampling (every scale_factor-th token of the flattened sequence, PixArt_blocks.py:82,102): forward and a training step
    "fwd_d2_kvevery": (dict(depth=2, input_size=16, model_max_length=20, kv_sampling="uniform_every", kv_scale_factor=2, kv_layers=(0, 1)),
                       dict(B=2, Hl=16, Wl=24, L=20, lens=[20, 11])),
    "train_d2_kvevery": (dict(depth=2, input_size=16, model_max_length=20, kv_sampling="uniform_every", kv_scale_factor=4, kv_layers=(1,)),
                         dict(B=2, Hl=16, Wl=16, L=20, lens=[20, 9])),
    # full-depth 2K with the shipped KV-compression layout (conv x2 on blocks 14..27: PixArt_sigma_xl2_img2K_internalms_kvcompress.py:44-49)
    "fwd_xl2_2k_kv_b1": (dict(depth=28, input_size=256, model_max_length=300, pe_interpolation=4.0, kv_sampling="conv", kv_scale_factor=2,
                              kv_layers=tuple(range(14, 28))), dict(B=1, Hl=256, Wl=256, L=300, lens=[300])),
    # ---- round 6: the inference configs END TO END at their configured step count (VERDICT r05 missing #2) ----
    # BASELINE.json configs[1]: the full-depth XL/2 at 512px, 20-step DPM-Solver++(2M) with CFG 4.5 (scripts/inference.py:107-118,
    # diffusion/model/dpm_solver.py:1196-1241), batch 2 (model batch 4), ragged captions; every intermediate x_t is kept (error against step index)
    "dpms_xl2_512_s20": (dict(depth=28, input_size=64, model_max_length=300, pe_interpolation=1.0), dict(B=2, Hl=64, Wl=64, L=300, lens=[300, 77])),
    # BASELINE.json configs[3]: 2K latent, the shipped KV-compression layout, a 4-step chain (8 full-depth N = 16384 forwards on the CPU)
    "dpms_xl2_2k_kv_s4": (dict(depth=28, input_size=256, model_max_length=300, pe_interpolation=4.0, kv_sampling="conv", kv_scale_factor=2,
                               kv_layers=tuple(range(14, 28))), dict(B=1, Hl=256, Wl=256, L=300, lens=[211])),
    # BASELINE.json configs[4]: the one-step generator of PixArt-alpha-DMD: eps at t = 400 without guidance, then eps -> x0 through the reference's own
    # eps_to_mu (scripts/DMD/transformer_train/generate.py:34-41; the diffusers pipeline takes the same pred_original_sample, scripts/diffusers_patches.py:448-449)
    "dmd_xl2_512_l120": (dict(depth=28, input_size=64, model_max_length=120, pe_interpolation=1.0), dict(B=2, Hl=64, Wl=64, L=120, lens=[120, 41])),
}
# steps of the DPM-Solver chains above (2 for the round-1 cases)
DPMS_STEPS = {"dpms_xl2_512_s20": 20, "dpms_xl2_2k_kv_s4": 4}


def build_reference(cfg, sd):
    ref_stubs.install()
    from diffusion.model.nets.PixArtMS import PixArtMS
    kvc = None
    if cfg.kv_sampling is not None:
        kvc = {"sampling": cfg.kv_sampling, "scale_factor": cfg.kv_scale_factor, "kv_compress_layer": list(cfg.kv_layers)}
    m = PixArtMS(depth=cfg.depth, hidden_size=cfg.hidden_size, patch_size=cfg.patch_size, num_heads=cfg.num_heads,
                 input_size=cfg.input_size, pe_interpolation=cfg.pe_interpolation, model_max_length=cfg.model_max_length,
                 class_dropout_prob=0.0, qk_norm=cfg.qk_norm, kv_compress_config=kvc, micro_condition=cfg.micro_condition)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert [k for k in missing if k != "pos_embed"] == [] and unexpected == [], (missing, unexpected)
    return m


def gen_case(name):
    ckw, ikw = CASES[name]
    cfg = OracleCfg(**ckw)
    sd = make_state_dict(cfg, seed=0)
    inp = make_inputs(seed=1, **ikw)
    m = build_reference(cfg, sd).eval()
    hw = torch.tensor([[inp["x"].shape[-2] * 8.0, inp["x"].shape[-1] * 8.0]] * inp["x"].shape[0])
    data_info = {"img_hw": hw, "aspect_ratio": torch.ones(inp["x"].shape[0], 1)}
    if cfg.micro_condition:       # per-sample original sizes / aspect ratios that differ, so a swapped or broadcast embedding shows
        Bn = inp["x"].shape[0]
        data_info = {"img_hw": torch.tensor([[1024.0, 768.0], [512.0, 1536.0], [640.0, 640.0], [2048.0, 1024.0]])[:Bn],
                     "aspect_ratio": torch.tensor([[1.33], [0.33], [1.0], [2.0]])[:Bn]}
    mask = inp["mask"] if ikw.get("lens") is not None else None
    out = {"cfg": ckw, "inputs": ikw, "weights_seed": 0, "inputs_seed": 1}
    if cfg.micro_condition:
        out["data_info"] = data_info
    t0 = time.time()
    if name.startswith("cfg_"):
        with torch.no_grad():
            out["cfg_scale"] = 4.5
            out["y"] = m.forward_with_cfg(inp["x"], inp["t"], inp["y"], 4.5, data_info, mask=mask).clone()
    elif name.startswith("fwd"):
        with torch.no_grad():
            out["y"] = m(inp["x"], inp["t"], inp["y"], mask=mask, data_info=data_info).clone()
    elif name.startswith("train"):
        from diffusion import IDDPM
        m.train()
        diff = IDDPM(str(1000), learn_sigma=True, pred_sigma=True, snr=False)
        t = inp["t"].clone()
        if t.numel() > 1:
            t[0] = 0  # exercise the decoder-NLL branch (gaussian_diffusion.py:741)
        terms = diff.training_losses(m, inp["x"], t, model_kwargs=dict(y=inp["y"], mask=mask[:, None, None, :], data_info=data_info), noise=inp["noise"])
        loss = terms["loss"].mean()
        loss.backward()
        out.update(t=t, loss=terms["loss"].detach().clone(), mse=terms["mse"].detach().clone(), vb=terms["vb"].detach().clone())
        grads = {}
        full_max, nsample = (8192, 4096) if cfg.depth <= 2 else (1152, 1024)   # full depth: 28 x the tensors, keep the fixture small
        for k, p in m.named_parameters():
            g = p.grad
            grads[k] = {"norm": g.norm().item(), "head": g.flatten()[:16].clone(), "sum": g.double().sum().item()}
            if g.numel() > full_max:       # evenly strided sample of the big tensors (every row / column region is hit)
                grads[k]["stride"] = g.numel() // nsample
                grads[k]["sample"] = g.flatten()[:: g.numel() // nsample].clone()
            if g.numel() <= full_max:
                grads[k]["full"] = g.clone()
        out["grads"] = grads
    elif name.startswith("iddpm"):
        from diffusion import IDDPM
        z = torch.cat([inp["x"][:2], inp["x"][:2]], dim=0)          # inference.py:91: randn(n, ...).repeat(2, 1, 1, 1)
        kw = dict(y=inp["y"], cfg_scale=4.5, data_info=data_info, mask=mask)
        out.update(steps=5, cfg_scale=4.5, noise_seed=11)
        with torch.no_grad():
            for key, clip in (("sample", False), ("sample_clip", True)):
                torch.manual_seed(11)                                # the per-step th.randn_like draws (gaussian_diffusion.py:438) come from here
                out[key] = IDDPM(str(5)).p_sample_loop(m.forward_with_cfg, z.shape, z, clip_denoised=clip, model_kwargs=kw, progress=False,
                                                       device="cpu").clone()
    elif name.startswith("sasolver"):
        from diffusion import SASolverSampler

        class _CpuSampler(SASolverSampler):          # the reference's register_buffer insists on torch.device("cuda") (sa_sampler.py:26-30): placement only
            def register_buffer(self, name_, attr):
                setattr(self, name_, attr)
        g = torch.Generator().manual_seed(7)
        null_y = torch.randn(1, 1, ikw["L"], 4096, generator=g).repeat(inp["x"].shape[0], 1, 1, 1)
        draws, real_randn_like = [], torch.randn_like

        def recording_randn_like(t, *a, **k):        # the solver's own Gaussian draws, in order (sa_solver.py:786, 813, 853): the GPU tests replay them
            d = real_randn_like(t, *a, **k)
            draws.append(d.clone())
            return d
        out.update(steps=6, eta=1, cfg_scale=4.5, null_seed=7, noise_seed=13)
        with torch.no_grad():
            torch.manual_seed(13)
            torch.randn_like = recording_randn_like
            try:
                s_, _ = _CpuSampler(m.forward_with_dpmsolver, device="cpu").sample(
                    S=6, batch_size=inp["x"].shape[0], shape=tuple(inp["x"].shape[1:]), eta=1, conditioning=inp["y"], unconditional_conditioning=null_y,
                    unconditional_guidance_scale=4.5, model_kwargs=dict(data_info=data_info, mask=mask), x_T=inp["x"].clone())
            finally:
                torch.randn_like = real_randn_like
        out["sample"], out["draws"] = s_.clone(), draws
    elif name.startswith("dmd"):
        import importlib.util
        import types
        from diffusion.model import gaussian_diffusion as gd
        spec = importlib.util.spec_from_file_location("_ref_dmd_generate", os.path.join(ref_stubs.REFERENCE_ROOT, "scripts/DMD/transformer_train/generate.py"))
        gen_mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(gen_mod)
        # the scheduler of the alpha / DMD pipelines: linear betas 1e-4 .. 2e-2 over 1000 steps; eps_to_mu reads .alphas_cumprod only
        betas = torch.tensor(gd.get_named_beta_schedule("linear", 1000))
        sched = types.SimpleNamespace(alphas_cumprod=torch.cumprod(1.0 - betas, dim=0))
        t = torch.full((inp["x"].shape[0],), 400, dtype=torch.long)          # app/app_pixart_dmd.py:193-196: timesteps=[400], one step, no guidance
        with torch.no_grad():
            eps = m.forward_with_dpmsolver(inp["x"], t, inp["y"], data_info=data_info, mask=mask)
            out["eps"] = eps.clone()
            out["x0"] = gen_mod.eps_to_mu(sched, eps, inp["x"], t).clone()
        out["t"], out["abar_t"] = 400, float(sched.alphas_cumprod[400])
    elif name.startswith("dpms") or name.startswith("cfg1"):
        from diffusion import DPMS
        g = torch.Generator().manual_seed(7)
        null_y = torch.randn(1, 1, ikw["L"], 4096, generator=g).repeat(inp["x"].shape[0], 1, 1, 1)
        steps = DPMS_STEPS.get(name, 2)
        with torch.no_grad():
            if steps == 2:
                out["fwd"] = m(inp["x"], inp["t"], inp["y"], mask=mask, data_info=data_info).clone()
            dpms = DPMS(m.forward_with_dpmsolver, condition=inp["y"], uncondition=null_y, cfg_scale=4.5,
                        model_kwargs=dict(data_info=data_info, mask=mask))
            if steps == 2:
                out["sample"] = dpms.sample(inp["x"], steps=2, order=2, skip_type="time_uniform", method="multistep").clone()
            else:       # x_t after every solver step (intermediates[0] is the initial latent, [-1] the sample: dpm_solver.py:1207-1234)
                s_, inter = dpms.sample(inp["x"], steps=steps, order=2, skip_type="time_uniform", method="multistep", return_intermediate=True)
                out["sample"], out["steps"] = s_.clone(), steps
                out["intermediates"] = torch.stack([v.clone() for v in inter])
        out["null_seed"] = 7
    out["ref_seconds"] = time.time() - t0
    return out


def gen_tables():
    ref_stubs.install()
    from diffusion.model.nets.PixArt import get_2d_sincos_pos_embed
    from diffusion.model.nets.PixArt_blocks import TimestepEmbedder
    out = {"pos": {}, "temb": {}}
    for (h, w, pe, base) in [(16, 16, 0.5, 16), (8, 12, 1.0, 8), (64, 64, 2.0, 64), (32, 48, 1.0, 32)]:
        tab = get_2d_sincos_pos_embed(1152, (h, w), pe_interpolation=pe, base_size=base)
        out["pos"][(h, w, pe, base)] = torch.from_numpy(tab[:: max(1, (h * w) // 37)]).clone()  # strided rows, float64
    ts = torch.tensor([0.0, 1.0, 17.5, 499.0, 999.0, 998.999])
    out["temb"]["t"] = ts
    out["temb"]["emb"] = TimestepEmbedder.timestep_embedding(ts, 256).clone()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    torch.manual_seed(0)
    names = [a.only] if a.only else list(CASES) + ["tables"]
    for n in names:
        t0 = time.time()
        obj = gen_tables() if n == "tables" else gen_case(n)
        path = os.path.join(GOLDEN_DIR, n + ".pt")
        torch.save(obj, path)
        print(f"{n}: {os.path.getsize(path) / 1024:.1f} KiB in {time.time() - t0:.1f}s")


if __name__ == "__main__":
    main()
