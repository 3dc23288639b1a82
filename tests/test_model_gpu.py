"""Model-level parity of the HIP path (pixart_sigma_amd.PixArtMS on MI355X) against
  (a) tests/golden/*.pt - outputs of the unmodified reference in fp32 (oracle/make_golden.py), and
  (b) oracle/pixart_oracle.py evaluated on the host with 16-bit rounding at the HIP path's rounding points (rp=True).

The file runs under either operand build (PXA_OPERAND_DTYPE, one library per type; tests/test_f16_parity_gpu.py re-runs it in a
subprocess under f16).  Bounds, rel-L2, by build:

  fp16 operands (the reference's own mixed precision, configs/PixArt_xl2_internal.py:57; the build BASELINE.json's <= 1e-3 is
  stated for):   forward <= 1e-3 vs the fp32 reference; loss <= 1e-3; every parameter gradient <= 1.2e-3 (loss-scaled backward; all but the
  cross-attention query projection measure <= 9.8e-4 - DESIGN.md section 2 says why that one sits at 1.08e-3).
  bf16 operands (default training build): one bf16 rounding of a tensor is 1.6e-3 by itself, so <= 1e-3 is unreachable by
  construction: forward <= 3e-3 vs the same-rounding-point oracle and <= 1e-2 vs fp32; gradients <= 3e-2; loss <= 5e-3.
Every test prints the measured error next to its bound."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import record_parity, rel_l2  # noqa: E402
from oracle import pixart_oracle as po  # noqa: E402
from oracle.weights import make_inputs, make_state_dict  # noqa: E402
from pixart_sigma_amd import lib as _lib  # noqa: E402

F16 = _lib.OPERAND == "f16"
po.RP_DTYPE = _lib.OPERAND_DTYPE
FWD_RP_TOL, FWD_F32_TOL = (1e-3, 1e-3) if F16 else (3e-3, 1e-2)
FWD_DEEP_TOL = 1e-3 if F16 else 2e-2            # depth-28 XL/2 (error grows with depth)
SAMPLE_TOL = 2e-3 if F16 else 2e-2              # 2-step CFG-4.5 sampler amplifies the forward error
LOSS_TOL = 1e-3 if F16 else 5e-3
GRAD_TOL = 1.2e-3 if F16 else 3e-2          # fp16: worst tensor measured 1.08e-3 (cross-attention q gradient: dP - delta cancellation)
# Full depth (train_xl2_1024_b1): rounding noise accumulates over 28 blocks of backward.  The yardstick is the REFERENCE's own mixed-precision path
# (oracle/ref_fp16_noise.py -> profiles/r03_reference_fp16_autocast_grad_noise.txt: the unmodified reference under torch.autocast(float16) against its own
# fp32 run on the same inputs): worst tensor 1.19e-3 (y_embedder fc1), cross_attn.q_linear 1.09e-3, at depth 2 up to 1.29e-3.  This path measures
# 1.26e-3 worst (blocks.5.mlp.fc1.weight), q_linear 1.06e-3 - the same noise floor; the bound sits just above both.
GRAD_TOL_DEEP = 1.5e-3 if F16 else 3e-2
REF_NOISE_RATIO = 1.1                           # fp16 build: NO parameter gradient's error exceeds 1.1 x the error the reference's own fp16-autocast run makes on the SAME tensor


def _ref_fp16_noise():
    import json
    import os
    p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_fp16_noise.json")
    return json.load(open(p)) if os.path.exists(p) else {}


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def _build(g, train=False):
    from pixart_sigma_amd import build_model
    cfg = po.OracleCfg(**g["cfg"])
    sd = make_state_dict(cfg, seed=g["weights_seed"])
    inp = make_inputs(seed=g["inputs_seed"], **g["inputs"])
    kvc = None
    if cfg.kv_sampling is not None:
        kvc = {"sampling": cfg.kv_sampling, "scale_factor": cfg.kv_scale_factor, "kv_compress_layer": list(cfg.kv_layers)}
    m = build_model("PixArtMS", depth=cfg.depth, hidden_size=1152, num_heads=16, input_size=cfg.input_size,
                    pe_interpolation=cfg.pe_interpolation, model_max_length=cfg.model_max_length, class_dropout_prob=0.0,
                    kv_compress_config=kvc, qk_norm=cfg.qk_norm, micro_condition=cfg.micro_condition)
    m.load_state_dict(sd)
    m = m.cuda()
    m.train(train)
    mask = inp["mask"] if g["inputs"].get("lens") is not None else None
    return cfg, sd, inp, mask, m


def _backward_scaled(make_loss, m):
    """forward + loss.backward() for either build; returns (terms, scale).  fp16 operands: the loss is scaled by the largest power of
    two <= 2^16 whose gradients stay finite (what the dynamic LossScaler converges to; the step is re-run on overflow exactly as a
    skipped step would be) and the flat gradient buffer is unscaled afterwards."""
    scale = 65536.0 if F16 else 1.0
    while True:
        if m._store is not None:
            m._store.grad.zero_()
        terms = make_loss()
        (terms["loss"].mean() * scale).backward()
        if not F16 or torch.isfinite(m._store.grad).all() or scale <= 1.0:
            break
        scale /= 2
    if scale != 1.0:
        m._store.grad.div_(scale)
    return terms, scale


@pytest.mark.parametrize("name", ["fwd_d2_sq", "fwd_d2_nomask", "fwd_d2_kvconv", "fwd_d2_kvuniform", "fwd_d2_kvave", "fwd_d2_kvevery", "fwd_d2_qknorm", "fwd_d2_micro"])
def test_forward_matches_reference_and_oracle(golden, name):
    g = golden(name)
    cfg, sd, inp, mask, m = _build(g)
    di = g.get("data_info")       # micro-conditioning (SizeEmbedder on img_hw / aspect_ratio, PixArtMS.py:187-191)
    with torch.no_grad():
        y = m(inp["x"].cuda(), inp["t"].cuda(), inp["y"].cuda(), mask=None if mask is None else mask.cuda(), data_info=di).cpu()
        y_rp = po.forward(sd, cfg, inp["x"], inp["t"], inp["y"], mask, rp=True, data_info=di)
    e_rp, e_f32 = rel_l2(y, y_rp), rel_l2(y, g["y"])
    print(f"\n[{name}] rel-L2 vs rounding-point oracle {e_rp:.2e}, vs fp32 reference {e_f32:.2e} (oracle-rp vs fp32 {rel_l2(y_rp, g['y']):.2e})")
    record_parity(f"{name}: forward vs fp32 reference", e_f32, FWD_F32_TOL); record_parity(f"{name}: forward vs rounding-point oracle", e_rp, FWD_RP_TOL)
    assert y.shape == g["y"].shape and torch.isfinite(y).all()
    assert e_rp < FWD_RP_TOL
    assert e_f32 < FWD_F32_TOL


@pytest.mark.parametrize("name", ["fwd_512_l300", "fwd_512_l120", "fwd_1024_b2", "fwd_2k_kv"])
def test_forward_at_baseline_token_geometries(golden, name):
    """BASELINE.json configs[1..4] at their real token counts (depth 2): N = 1024 (512px, L = 300 and the multi-aspect L = 120
    alpha-DMD shape), N = 4096 (1024px training shape), N = 16384 with KV compression to 4096 on one block (2K).  These are the
    grids the benchmarks time: attention with 32-128 key tiles per query tile, persistent GEMMs with M = B*N >= 2048."""
    g = golden(name)
    cfg, sd, inp, mask, m = _build(g)
    with torch.no_grad():
        y = m(inp["x"].cuda(), inp["t"].cuda(), inp["y"].cuda(), mask=mask.cuda()).cpu()
    e = rel_l2(y, g["y"])
    print(f"\n[{name}] N={(inp['x'].shape[-1] // 2) * (inp['x'].shape[-2] // 2)} rel-L2 vs fp32 reference {e:.2e} (bound {FWD_F32_TOL:.0e})")
    record_parity(f"{name}: forward vs fp32 reference", e, FWD_F32_TOL)
    assert y.shape == g["y"].shape and torch.isfinite(y).all()
    assert e < FWD_F32_TOL


@pytest.mark.parametrize("name", ["fwd_xl2_1024_b1", "fwd_xl2_2k_kv_b1"])
def test_forward_full_depth_at_headline_geometry(golden, name):
    """Round 3 (VERDICT r02 missing #2): PixArtMS_XL_2 at FULL depth (28 blocks, PixArtMS.py:291-293) on the benchmark's own geometry - 1024px, N = 4096,
    L = 300 - and on the 2K latent (N = 16384) with the shipped KV-compression layout (conv x2 on blocks 14..27), against the fp32 reference output."""
    g = golden(name)
    cfg, sd, inp, mask, m = _build(g)
    with torch.no_grad():
        y = m(inp["x"].cuda(), inp["t"].cuda(), inp["y"].cuda(), mask=mask.cuda()).cpu()
    e = rel_l2(y, g["y"])
    print(f"\n[{name}] depth {cfg.depth} N={(inp['x'].shape[-1] // 2) * (inp['x'].shape[-2] // 2)} rel-L2 vs fp32 reference {e:.2e} (bound {FWD_DEEP_TOL:.0e})")
    record_parity(f"{name}: forward (depth {cfg.depth}) vs fp32 reference", e, FWD_DEEP_TOL)
    assert y.shape == g["y"].shape and torch.isfinite(y).all()
    assert e < FWD_DEEP_TOL


def test_forward_with_cfg_matches_reference(golden):
    """PixArtMS.forward_with_cfg (reference PixArtMS.py:221-234)."""
    g = golden("cfg_d2")
    cfg, sd, inp, mask, m = _build(g)
    with torch.no_grad():
        y = m.forward_with_cfg(inp["x"].cuda(), inp["t"].cuda(), inp["y"].cuda(), g["cfg_scale"], None, mask=mask.cuda()).cpu()
    e = rel_l2(y, g["y"])
    print(f"\nforward_with_cfg rel-L2 vs fp32 reference {e:.2e}")
    record_parity("cfg_d2: forward_with_cfg vs fp32 reference", e, FWD_F32_TOL)
    assert e < FWD_F32_TOL
    assert torch.equal(y[:2, :3], y[2:, :3]) and not torch.equal(y[:2, 3:], y[2:, 3:])


@pytest.mark.parametrize("key,clip", [("sample", False), ("sample_clip", True)])
def test_iddpm_ancestral_sampling_matches_reference(golden, key, clip):
    """`--sampling_algo iddpm` (scripts/inference.py:89-101): IDDPM(str(5)).p_sample_loop over forward_with_cfg on the HIP path against the reference's
    own 5-step chain; the per-step noise is the reference's (same CPU generator draws, gaussian_diffusion.py:438)."""
    from pixart_sigma_amd import IDDPM
    g = golden("iddpm_d2")
    cfg, sd, inp, mask, m = _build(g)
    z = torch.cat([inp["x"][:2], inp["x"][:2]], dim=0).cuda()
    torch.manual_seed(g["noise_seed"])
    kw = dict(y=inp["y"].cuda(), cfg_scale=g["cfg_scale"], data_info=None, mask=mask.cuda())
    out = IDDPM(str(g["steps"])).p_sample_loop(m.forward_with_cfg, z.shape, z, clip_denoised=clip, model_kwargs=kw, device="cuda",
                                               step_noise=lambda x: torch.randn(x.shape).to(x.device)).cpu()
    e = rel_l2(out, g[key])
    print(f"\nIDDPM 5-step ancestral sample ({key}) rel-L2 vs fp32 reference {e:.2e}")
    record_parity(f"iddpm_d2: 5-step ancestral sample ({key})", e)
    # clip_denoised=True (the method's default; the script passes False) clamps x0_hat = 150 (x_t - ...) at the first steps: with random-init weights
    # most of it saturates at +-1 and the elements near zero flip side on a 1e-3 change of eps - the chain is ill-conditioned there (measured 5.7e-3 fp16 /
    # 2.0e-2 bf16 against 7.2e-4 / 5.6e-3 without the clamp).  The clamp logic itself is pinned to 5e-5 on the CPU (tests/test_host_logic.py).
    assert e < (5 * SAMPLE_TOL if clip else SAMPLE_TOL)


def test_fixed_resolution_pixart_forward(golden):
    """The `PixArt` registry class (PixArt.py:62-143) on a square latent equals PixArtMS on the same weights (golden fwd_d2_sq)."""
    from pixart_sigma_amd import build_model
    g = golden("fwd_d2_sq")
    cfg = po.OracleCfg(**g["cfg"])
    sd = make_state_dict(cfg, seed=g["weights_seed"])
    inp = make_inputs(seed=g["inputs_seed"], **g["inputs"])
    m = build_model("PixArt", depth=cfg.depth, hidden_size=1152, num_heads=16, input_size=cfg.input_size, pe_interpolation=cfg.pe_interpolation,
                    model_max_length=cfg.model_max_length, class_dropout_prob=0.0)
    sd["pos_embed"] = m.pos_embed.clone()
    m.load_state_dict(sd)
    m = m.cuda().eval()
    with torch.no_grad():
        y = m(inp["x"].cuda(), inp["t"].cuda(), inp["y"].cuda(), mask=inp["mask"].cuda()).cpu()
        eps = m.forward_with_dpmsolver(inp["x"].cuda(), inp["t"].cuda(), inp["y"].cuda(), mask=inp["mask"].cuda()).cpu()
    assert rel_l2(y, g["y"]) < FWD_F32_TOL and torch.equal(eps, y[:, :4])
    with pytest.raises(AssertionError):
        m(inp["x"][..., :8].cuda(), inp["t"].cuda(), inp["y"].cuda())


@pytest.mark.parametrize("gname", ["train_d2_plain", "train_d2", "train_d2_qknorm", "train_d2_micro", "train_d2_kvevery", "train_1024_b2", "train_xl2_1024_b1"])
def test_training_step_loss_and_grads(golden, gname):
    """IDDPM training_losses + backward vs the reference's loss and EVERY parameter gradient.  train_xl2_1024_b1 (round 3) is the model bench.py
    times - PixArtMS_XL_2, depth 28, 1024px (N = 4096, L = 300) - batch 1: the gradient of every one of its 611 M parameters (norm, leading
    elements and a strided 1,024-element sample of each large tensor; small tensors in full) against the reference's.  train_d2 has KV compression ('conv', x2)
    on block 1 (kv_compress_bwd, shared sr/norm gradients); train_d2_micro the SizeEmbedders; train_1024_b2 is the benchmark's token
    geometry (N = 4096, L = 300: the dW GEMMs reduce over K = 8192 tokens, attention backward runs 32 key blocks x 64 query tiles)."""
    from pixart_sigma_amd import IDDPM
    g = golden(gname)
    cfg, sd, inp, mask, m = _build(g, train=True)
    diff = IDDPM(str(1000), learn_sigma=True, pred_sigma=True, snr=False)
    kw = dict(y=inp["y"].cuda(), mask=mask[:, None, None, :].cuda(), data_info=g.get("data_info"))
    terms, scale = _backward_scaled(lambda: diff.training_losses(m, inp["x"].cuda(), g["t"].cuda(), model_kwargs=kw, noise=inp["noise"].cuda()), m)
    e_loss = rel_l2(terms["loss"].cpu(), g["loss"])
    print(f"\n[{gname}] loss {terms['loss'].tolist()} ref {g['loss'].tolist()} rel-L2 {e_loss:.2e} (bound {LOSS_TOL:.0e}); loss scale {scale:g}")
    record_parity(f"{gname}: loss", e_loss, LOSS_TOL)
    assert e_loss < LOSS_TOL
    worst = []
    gmax = max(r["norm"] for r in g["grads"].values())
    for k, p in m.named_parameters():
        ref = g["grads"][k]
        gr = p.grad.detach().float().cpu()
        if ref["norm"] < 1e-9:            # mathematically zero gradient (k_norm.bias cancels in the softmax): only rounding noise may remain
            assert gr.norm().item() < 1e-3 * gmax, k
            continue
        e_norm = abs(gr.norm().item() - ref["norm"]) / (ref["norm"] + 1e-12)
        if "full" in ref:
            e_el = rel_l2(gr, ref["full"])
        elif "sample" in ref:
            e_el = rel_l2(gr.flatten()[:: ref["stride"]], ref["sample"])
        else:
            e_el = 0.0                    # round-1 goldens keep 16 leading elements only: checked loosely below
            assert rel_l2(gr.flatten()[:16], ref["head"]) < 10 * GRAD_TOL or ref["head"].norm() < 1e-6 * ref["norm"], k
        worst.append((max(e_norm, e_el), e_norm, e_el, k))
    worst.sort(reverse=True)
    for w in worst[:8]:
        print("grad err (max, norm, elementwise) %.2e %.2e %.2e %s" % w)
    # Side by side with the REFERENCE's own mixed-precision noise on the same inputs (tests/golden/ref_fp16_noise.json: the unmodified reference under
    # torch.autocast(float16) + 65536 loss scale against its own fp32 run, oracle/ref_fp16_noise.py) - VERDICT r05 item 7.  Per family of tensors: this path, the
    # reference's fp16 path, the ratio.
    noise = _ref_fp16_noise().get(gname)
    if noise is not None:
        fam = {}
        for e, _, _, k in worst:
            f = ".".join(p_ for p_ in k.split(".") if not p_.isdigit())
            r = noise["grad_rel"].get(k)
            if r is None:
                continue
            o = fam.setdefault(f, [0.0, 0.0, 0.0, k])
            o[0], o[1] = max(o[0], e), max(o[1], r)
            if e / r > o[2]:
                o[2], o[3] = e / r, k
        print(f"  {'family':44s} {'this path':>10s} {'ref fp16':>10s} {'worst ratio of one tensor':>26s}")
        for f, (e, r, q, k) in sorted(fam.items(), key=lambda kv: -kv[1][0])[:14]:
            print(f"  {f:44s} {e:10.2e} {r:10.2e} {q:10.2f}  ({k})")
        ours_worst, ref_worst = worst[0][0], noise["worst"]
        ratio_worst = max(v[2] for v in fam.values())
        print(f"  worst tensor: this path {ours_worst:.2e}, reference fp16 path {ref_worst:.2e}; worst per-tensor ratio {ratio_worst:.2f}; loss: {e_loss:.2e} vs {noise['loss_rel']:.2e}")
        record_parity(f"{gname}: worst gradient / the reference's own fp16-autocast worst gradient", ours_worst / ref_worst, REF_NOISE_RATIO if F16 else None)
        record_parity(f"{gname}: worst per-tensor ratio to the reference's fp16-autocast error of the same tensor", ratio_worst, REF_NOISE_RATIO if F16 else None)
        if F16:      # measured (profiles/r6_07_parity_summary_f16.json): worst-of-all ratio 0.49 ... 1.00, worst per-tensor ratio 0.83 ... 1.01
            assert ours_worst <= REF_NOISE_RATIO * ref_worst, (ours_worst, ref_worst)
            assert ratio_worst <= REF_NOISE_RATIO, (ratio_worst, [v for v in fam.values() if v[2] > REF_NOISE_RATIO])
    record_parity(f"{gname}: worst parameter gradient ({worst[0][3]})", worst[0][0], GRAD_TOL_DEEP if cfg.depth > 2 else GRAD_TOL)
    assert worst[0][0] < (GRAD_TOL_DEEP if cfg.depth > 2 else GRAD_TOL), worst[0]
    # gradients live in the flat buffer the fused optimizer / all-reduce work on
    st = m._store
    assert all(p.grad.data_ptr() == st.grad.data_ptr() + 4 * st.offset[n] for n, p in st.params.items())


def test_grad_checkpointing_matches_saved_activations(golden):
    """auto_grad_checkpoint semantics (diffusion/model/utils.py:38-45): recompute-in-backward gives the same gradients."""
    from pixart_sigma_amd import IDDPM
    from pixart_sigma_amd.model.utils import set_grad_checkpoint
    g = golden("train_d2_plain")
    cfg, sd, inp, mask, m = _build(g, train=True)
    diff = IDDPM(str(1000))
    kw = dict(y=inp["y"].cuda(), mask=mask.cuda())
    grads = []
    for ck in (False, True):
        if ck:
            set_grad_checkpoint(m)
        if m._store is not None:
            m._store.grad.zero_()
        loss = diff.training_losses(m, inp["x"].cuda(), g["t"].cuda(), model_kwargs=kw, noise=inp["noise"].cuda())["loss"].mean()
        (loss * (1024.0 if F16 else 1.0)).backward()
        grads.append(m._store.grad.clone())
    assert rel_l2(grads[1], grads[0]) < 1e-4


def test_sa_solver_sampling_matches_reference(golden):
    """`--sampling_algo sa-solver` on the HIP denoiser against the reference's 6-step chain (tests/golden/sasolver_d2.pt), replaying the reference's
    Gaussian draws (the device generator's stream differs from the CPU's)."""
    from pixart_sigma_amd import SASolverSampler
    g = golden("sasolver_d2")
    cfg, sd, inp, mask, m = _build(g)
    gen = torch.Generator().manual_seed(g["null_seed"])
    null_y = torch.randn(1, 1, g["inputs"]["L"], 4096, generator=gen).repeat(inp["x"].shape[0], 1, 1, 1).cuda()
    s, _ = SASolverSampler(m.forward_with_dpmsolver, device="cuda").sample(
        S=g["steps"], batch_size=inp["x"].shape[0], shape=tuple(inp["x"].shape[1:]), eta=g["eta"], conditioning=inp["y"].cuda(), unconditional_conditioning=null_y,
        unconditional_guidance_scale=g["cfg_scale"], model_kwargs=dict(data_info=None, mask=mask.cuda()), x_T=inp["x"].cuda(), normals_sequence=g["draws"])
    e = rel_l2(s.cpu(), g["sample"])
    print(f"\n6-step SA-Solver sample rel-L2 vs reference {e:.2e}")
    record_parity("sasolver_d2: 6-step SA-Solver sample", e, SAMPLE_TOL if F16 else FWD_F32_TOL)
    assert e < (SAMPLE_TOL if F16 else FWD_F32_TOL)


def test_dpm_solver_sampling_matches_reference(golden):
    from pixart_sigma_amd import DPMS
    g = golden("dpms_d2")
    cfg, sd, inp, mask, m = _build(g)
    gen = torch.Generator().manual_seed(g["null_seed"])
    null_y = torch.randn(1, 1, g["inputs"]["L"], 4096, generator=gen).repeat(inp["x"].shape[0], 1, 1, 1).cuda()
    s = DPMS(m.forward_with_dpmsolver, condition=inp["y"].cuda(), uncondition=null_y, cfg_scale=4.5,
             model_kwargs=dict(data_info=None, mask=mask.cuda())).sample(inp["x"].cuda(), steps=2, order=2, skip_type="time_uniform", method="multistep")
    e = rel_l2(s.cpu(), g["sample"])
    print(f"\n2-step DPM-Solver++ sample rel-L2 vs reference {e:.2e}")
    record_parity("dpms_d2: 2-step DPM-Solver++ sample", e, SAMPLE_TOL if F16 else FWD_F32_TOL)
    assert e < (SAMPLE_TOL if F16 else FWD_F32_TOL)


# ---- round 6: the inference configs end to end at their configured step count (VERDICT r05 missing #2).  Measured (profiles/r6_03_pytest_sel.txt): the error of x_t
# against the reference chain is made in the first two solver steps (t = 1 -> 0.9: sigma_t ~ 1, the whole eps error lands in x) and then STAYS - fp16 operands
# 4.9e-4 after step 1, 5.7e-4 after step 20; bf16 4.0e-3 -> 4.6e-3 - the later steps contract (sigma_t / sigma_s < 1) about as fast as they add.  So the fp16
# build meets north_star's 1e-3 on the 20-step sample itself; the bounds are the measured end-of-chain errors with ~1.7x headroom.
# The 2K chain (N = 16384, KV compression on blocks 14..27, 4 steps) settles at 1.0e-3 (fp16) / 7.9e-3 (bf16) after its FIRST step: classifier-free guidance at
# 4.5 multiplies the error of eps_c - eps_u, and the 2K forward alone measures 5.8e-4 against 5.3e-4 ... 5.6e-4 at 512 / 1024px (profiles/r6_07_pytest_gpu.txt).
CHAIN_TOL = {"dpms_xl2_512_s20": (1e-3, 1e-2), "dpms_xl2_2k_kv_s4": (1.5e-3, 1.2e-2)}      # (fp16, bf16)


@pytest.mark.parametrize("name", ["dpms_xl2_512_s20", "dpms_xl2_2k_kv_s4"])
def test_dpm_solver_chain_at_configured_steps_full_depth(golden, name):
    """BASELINE configs[1] (XL/2 512px, 20-step DPM-Solver++(2M), CFG 4.5: scripts/inference.py:107-118, model/dpm_solver.py:1196-1241) and configs[3]
    (2K, KV compression on blocks 14..27, 4 steps) at FULL DEPTH against the reference's own chain: x_t after EVERY solver step, error printed per step."""
    from pixart_sigma_amd import DPMS
    g = golden(name)
    cfg, sd, inp, mask, m = _build(g)
    gen = torch.Generator().manual_seed(g["null_seed"])
    null_y = torch.randn(1, 1, g["inputs"]["L"], 4096, generator=gen).repeat(inp["x"].shape[0], 1, 1, 1).cuda()
    solver = DPMS(m.forward_with_dpmsolver, condition=inp["y"].cuda(), uncondition=null_y, cfg_scale=4.5, model_kwargs=dict(data_info=None, mask=mask))
    s, inter = solver.sample(inp["x"].cuda(), steps=g["steps"], order=2, skip_type="time_uniform", method="multistep", return_intermediate=True)
    ref = g["intermediates"]
    assert len(inter) == ref.shape[0] == g["steps"] + 1 and torch.equal(inter[0].cpu(), ref[0])
    errs = [rel_l2(a.cpu(), b) for a, b in zip(inter, ref)]
    print(f"\n[{name}] rel-L2 of x_t vs the reference chain, by solver step: " + " ".join(f"{i}:{e:.1e}" for i, e in enumerate(errs)))
    e = rel_l2(s.cpu(), g["sample"])
    tol = CHAIN_TOL[name][0 if F16 else 1]
    record_parity(f"{name}: {g['steps']}-step DPM-Solver++ sample (depth 28)", e, tol)
    record_parity(f"{name}: worst x_t over the chain (step {max(range(len(errs)), key=errs.__getitem__)})", max(errs), tol)
    assert torch.isfinite(s).all() and max(errs) < tol, errs
    if name == "dpms_xl2_512_s20":          # ... and the same 20-step loop captured as ONE HIP graph (20 x ~700 launches) reproduces the eager sample bit for bit
        sg = solver.sample_graphed(inp["x"].cuda(), steps=g["steps"], order=2, skip_type="time_uniform", method="multistep")
        assert torch.equal(sg, s)


def test_dmd_one_step_generator_matches_reference(golden):
    """BASELINE configs[4]: the PixArt-alpha-DMD generator step - eps at t = 400 without guidance, x0 = (x_t - sqrt(1 - abar_t) eps) / sqrt(abar_t)
    (scripts/DMD/transformer_train/generate.py:20-41, scripts/diffusers_patches.py:448-449) - full depth, 512px, L = 120, ragged captions."""
    g = golden("dmd_xl2_512_l120")
    cfg, sd, inp, mask, m = _build(g)
    x = inp["x"].cuda()
    t = torch.full((x.shape[0],), g["t"], device="cuda", dtype=torch.long)
    with torch.no_grad():
        eps = m.forward_with_dpmsolver(x, t, inp["y"].cuda(), data_info=None, mask=mask)
    abar = g["abar_t"]
    x0 = (x - (1.0 - abar) ** 0.5 * eps) / abar ** 0.5                      # what tools/bench_dmd.py times in front of the VAE decode
    e_eps, e_x0 = rel_l2(eps.cpu(), g["eps"]), rel_l2(x0.cpu(), g["x0"])
    print(f"\nDMD one-step generator (depth 28, N = 1024, L = 120): eps rel-L2 {e_eps:.2e}, x0 rel-L2 {e_x0:.2e} vs the reference")
    record_parity("dmd_xl2_512_l120: eps at t = 400", e_eps, FWD_DEEP_TOL); record_parity("dmd_xl2_512_l120: one-step x0", e_x0, FWD_DEEP_TOL)
    assert e_eps < FWD_DEEP_TOL and e_x0 < FWD_DEEP_TOL


def test_graphed_sampler_follows_weight_updates(golden):
    """a captured sampling graph must read CURRENT weights after they change.  The q-prescaled copy of the qkv projection is a buffer derived from
    the weights; it lives at a fixed address and is rewritten in place with the shadow (ParamStore.bump), so a replay after load_state_dict equals a fresh
    eager sample - with the per-generation allocation it replaces the replay read the old block."""
    from pixart_sigma_amd import DPMS
    g = golden("dpms_d2")
    cfg, sd, inp, mask, m = _build(g)
    gen = torch.Generator().manual_seed(g["null_seed"])
    null_y = torch.randn(1, 1, g["inputs"]["L"], 4096, generator=gen).repeat(inp["x"].shape[0], 1, 1, 1).cuda()
    solver = DPMS(m.forward_with_dpmsolver, condition=inp["y"].cuda(), uncondition=null_y, cfg_scale=4.5, model_kwargs=dict(data_info=None, mask=mask))
    kw = dict(steps=3, order=2, skip_type="time_uniform", method="multistep")
    x = inp["x"].cuda()
    g1 = solver.sample_graphed(x, **kw)
    assert torch.equal(g1, solver.sample(x, **kw))
    graph = solver._graph
    qs_ptr = m._engine._qs[0].data_ptr()
    m.load_state_dict({k: (v * 1.1 if ".attn.qkv." in k else v) for k, v in m.state_dict().items()})
    g2 = solver.sample_graphed(x, **kw)
    assert solver._graph is graph and m._engine._qs[0].data_ptr() == qs_ptr          # the same graph, the same derived buffer
    assert torch.equal(g2, solver.sample(x, **kw)) and not torch.equal(g2, g1)


def test_inference_text_cache_is_exact_and_invalidated(golden, monkeypatch):
    """engine.Engine._text_cache (round 3): the caption MLP and the 28 cross-attention kv_linear outputs depend on the text alone, so a sampler's
    steps reuse them.  Same sample bit for bit with the cache off; a different caption tensor, an in-place edit of the same one, and a training
    forward in between all miss - and so does a change of the WEIGHTS under the same caption (the key carries ParamStore.generation)."""
    from pixart_sigma_amd import DPMS
    g = golden("dpms_d2")
    cfg, sd, inp, mask, m = _build(g)
    gen = torch.Generator().manual_seed(g["null_seed"])
    null_y = torch.randn(1, 1, g["inputs"]["L"], 4096, generator=gen).repeat(inp["x"].shape[0], 1, 1, 1).cuda()
    y = inp["y"].cuda()
    kw = dict(steps=3, order=2, skip_type="time_uniform", method="multistep")
    x = inp["x"].cuda()
    with torch.no_grad():
        solver = DPMS(m.forward_with_dpmsolver, condition=y, uncondition=null_y, cfg_scale=4.5, model_kwargs=dict(data_info=None, mask=mask))
        a = solver.sample(x, **kw)
        assert m._engine._text_cache is not None and all(k is not None for k in m._engine._text_cache["kvc"])
        monkeypatch.setenv("PXA_TEXT_CACHE", "0")
        b = DPMS(m.forward_with_dpmsolver, condition=y, uncondition=null_y, cfg_scale=4.5, model_kwargs=dict(data_info=None, mask=mask)).sample(x, **kw)
        monkeypatch.delenv("PXA_TEXT_CACHE")
        assert torch.equal(a, b)
        key0 = m._engine._text_cache["key"]
        y.mul_(0.5)                                           # in-place edit of the caption tensor: version bump -> miss -> different sample
        c = solver.sample(x, **kw)
        assert m._engine._text_cache["key"] != key0 and not torch.equal(c, a)
        # weights change under the same caption tensor: load_state_dict of another checkpoint between two no-grad samples must miss the cache
        key1 = m._engine._text_cache["key"]
        sd2 = {k_: (v_ * 1.25 if k_.endswith("cross_attn.kv_linear.weight") else v_) for k_, v_ in m.state_dict().items()}
        m.load_state_dict(sd2)
        d = solver.sample(x, **kw)
        assert m._engine._text_cache["key"] != key1 and not torch.equal(d, c)
        monkeypatch.setenv("PXA_TEXT_CACHE", "0")
        e = solver.sample(x, **kw)
        monkeypatch.delenv("PXA_TEXT_CACHE")
        assert torch.equal(d, e)                              # ... and what it recomputes is what a cache-less run computes
    m.train()
    from pixart_sigma_amd import IDDPM
    IDDPM(str(1000)).training_losses(m, x, inp["t"].cuda(), model_kwargs=dict(y=y, mask=mask.cuda()), noise=inp["noise"].cuda())["loss"].mean().backward()
    assert m._engine._text_cache is None                      # any training forward drops it


def test_dpm_solver_graphed_loop_replays_bit_exact(golden):
    """the sampling loop captured as one HIP graph reproduces the eager loop bit for bit, also for new latents."""
    from pixart_sigma_amd import DPMS
    g = golden("dpms_d2")
    cfg, sd, inp, mask, m = _build(g)
    gen = torch.Generator().manual_seed(g["null_seed"])
    null_y = torch.randn(1, 1, g["inputs"]["L"], 4096, generator=gen).repeat(inp["x"].shape[0], 1, 1, 1).cuda()
    solver = DPMS(m.forward_with_dpmsolver, condition=inp["y"].cuda(), uncondition=null_y, cfg_scale=4.5, model_kwargs=dict(data_info=None, mask=mask))
    kw = dict(steps=3, order=2, skip_type="time_uniform", method="multistep")
    x1 = inp["x"].cuda()
    x2 = torch.randn(x1.shape, generator=gen).cuda()
    e1, e2 = solver.sample(x1, **kw), solver.sample(x2, **kw)
    g1, g2 = solver.sample_graphed(x1, **kw), solver.sample_graphed(x2, **kw)
    assert torch.equal(g1, e1) and torch.equal(g2, e2) and not torch.equal(g1, g2)


def test_block_api_matches_oracle_block():
    """PixArtMSBlock.forward(x, y, t, mask=y_lens, HW) used stand-alone, forward + input gradients."""
    from pixart_sigma_amd.model.nets import PixArtMSBlock
    cfg = po.OracleCfg(depth=1, input_size=16, model_max_length=20)
    sd = make_state_dict(cfg, seed=3)
    B, N, D, lens = 2, 96, 1152, [20, 6]
    gen = torch.Generator().manual_seed(5)
    x, y, t0 = torch.randn(B, N, D, generator=gen), torch.randn(1, sum(lens), D, generator=gen), torch.randn(B, 6 * D, generator=gen) * 0.3
    blk = PixArtMSBlock(D, 16)
    blk.load_state_dict({k[len("blocks.0."):]: v for k, v in sd.items() if k.startswith("blocks.0.")})
    blk = blk.cuda()
    xg, yg, tg = (v.cuda().requires_grad_(True) for v in (x, y, t0))
    out = blk(xg, yg, tg, mask=lens, HW=(8, 12))
    xr, yr, tr = (v.clone().requires_grad_(True) for v in (x, y, t0))
    sd = dict(sd)
    sd["blocks.0.scale_shift_table"] = sd["blocks.0.scale_shift_table"].clone().requires_grad_(True)
    ref = po.block_forward(sd, 0, xr, yr.view(-1, D), tr, lens, (8, 12), cfg, rp=True)
    assert rel_l2(out.detach().cpu(), ref.detach()) < FWD_RP_TOL
    w = torch.randn(B, N, D, generator=gen)
    (out * w.cuda()).sum().backward()
    (ref * w).sum().backward()
    for a, b, nm in ((xg, xr, "dx"), (yg, yr, "dy"), (tg, tr, "dt")):
        e = rel_l2(a.grad.cpu(), b.grad)
        print(f"block {nm} rel-L2 {e:.2e}")
        assert e < (3e-3 if F16 else 2e-2), nm
    assert rel_l2(blk.scale_shift_table.grad.cpu(), sd["blocks.0.scale_shift_table"].grad) < (3e-3 if F16 else 2e-2)


def test_config1_xl2_256_full_depth(golden):
    """BASELINE.json configs[0] on the HIP path: XL/2 (depth 28) 256px, batch 2, CFG 4.5, 2-step DPM-Solver++."""
    from pixart_sigma_amd import DPMS
    g = golden("cfg1_xl2_256")
    cfg, sd, inp, mask, m = _build(g)
    with torch.no_grad():
        y = m(inp["x"].cuda(), inp["t"].cuda(), inp["y"].cuda(), mask=mask.cuda()).cpu()
    e = rel_l2(y, g["fwd"])
    print(f"\nXL/2 256px forward rel-L2 vs fp32 reference {e:.2e}")
    record_parity("cfg1_xl2_256: forward (depth 28) vs fp32 reference", e, FWD_DEEP_TOL)
    assert e < FWD_DEEP_TOL
    gen = torch.Generator().manual_seed(g["null_seed"])
    null_y = torch.randn(1, 1, 300, 4096, generator=gen).repeat(2, 1, 1, 1).cuda()
    s = DPMS(m.forward_with_dpmsolver, condition=inp["y"].cuda(), uncondition=null_y, cfg_scale=4.5,
             model_kwargs=dict(data_info=None, mask=mask.cuda())).sample(inp["x"].cuda(), steps=2, order=2)
    e = rel_l2(s.cpu(), g["sample"])
    print(f"XL/2 256px 2-step sample rel-L2 vs fp32 reference {e:.2e}")
    record_parity("cfg1_xl2_256: 2-step DPM-Solver++ sample", e)
    assert e < SAMPLE_TOL
