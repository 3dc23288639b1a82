"""CPU tests of the host-side logic around the HIP path: the diffusion loss and DPM-Solver loop (run here with the
oracle denoiser as the model callable, against the reference-generated goldens), the positional table, the registry,
state-dict compatibility, and the flat parameter store."""
import os

import numpy as np
import pytest
import torch

from conftest import rel_l2
from oracle import pixart_oracle as po
from oracle.weights import make_inputs, make_state_dict, param_shapes


def _setup(g):
    cfg = po.OracleCfg(**g["cfg"])
    sd = make_state_dict(cfg, seed=g["weights_seed"])
    inp = make_inputs(seed=g["inputs_seed"], **g["inputs"])
    return cfg, sd, inp, inp["mask"] if g["inputs"].get("lens") is not None else None


def test_training_losses_host_code_matches_reference(golden):
    from pixart_sigma_amd.diffusion import IDDPM
    g = golden("train_d2")
    cfg, sd, inp, mask = _setup(g)
    diff = IDDPM(str(1000), learn_sigma=True, pred_sigma=True, snr=False)
    model = lambda x, timestep, **kw: po.forward(sd, cfg, x, timestep, inp["y"], mask)
    with torch.no_grad():
        terms = diff.training_losses(model, inp["x"], g["t"], model_kwargs={}, noise=inp["noise"])
    for k in ("loss", "mse", "vb"):
        assert torch.allclose(terms[k], g[k], rtol=2e-5, atol=1e-6), k


def test_dpm_solver_host_code_matches_reference(golden):
    from pixart_sigma_amd.diffusion import DPMS
    g = golden("dpms_d2")
    cfg, sd, inp, mask = _setup(g)
    gen = torch.Generator().manual_seed(g["null_seed"])
    null_y = torch.randn(1, 1, g["inputs"]["L"], 4096, generator=gen).repeat(inp["x"].shape[0], 1, 1, 1)
    model = lambda x, t, y, mask=None, **kw: po.forward_with_dpmsolver(sd, cfg, x, t, y, mask)
    s = DPMS(model, condition=inp["y"], uncondition=null_y, cfg_scale=4.5, model_kwargs=dict(mask=mask)).sample(
        inp["x"], steps=2, order=2, skip_type="time_uniform", method="multistep")
    assert rel_l2(s, g["sample"]) < 5e-5


def test_sa_solver_host_code_matches_reference(golden):
    """`--sampling_algo sa-solver` (scripts/inference.py:119-133): pixart_sigma_amd.diffusion.sa_solver around the oracle's denoiser (CPU) reproduces the
    reference's 6-step SASolverSampler chain of tests/golden/sasolver_d2.pt when fed the reference's own Gaussian draws - host-side predictor /
    corrector coefficients, the tau window, warm-up and final-step orders."""
    from pixart_sigma_amd.diffusion import SASolverSampler
    g = golden("sasolver_d2")
    cfg, sd, inp, mask = _setup(g)
    gen = torch.Generator().manual_seed(g["null_seed"])
    null_y = torch.randn(1, 1, g["inputs"]["L"], 4096, generator=gen).repeat(inp["x"].shape[0], 1, 1, 1)
    model = lambda x, t, y, mask=None, **kw: po.forward_with_dpmsolver(sd, cfg, x, t, y, mask)
    assert len(g["draws"]) == g["steps"] + 1                     # one unused draw in front of the first evaluation + one per step
    s, _ = SASolverSampler(model, device="cpu").sample(S=g["steps"], batch_size=inp["x"].shape[0], shape=tuple(inp["x"].shape[1:]), eta=g["eta"],
                                                       conditioning=inp["y"], unconditional_conditioning=null_y, unconditional_guidance_scale=g["cfg_scale"],
                                                       model_kwargs=dict(mask=mask), x_T=inp["x"].clone(), normals_sequence=g["draws"])
    e = rel_l2(s, g["sample"])
    assert e < 5e-5, e
    # the draws matter (three of the six steps are stochastic): zero noise gives a different sample
    z, _ = SASolverSampler(model, device="cpu").sample(S=g["steps"], batch_size=inp["x"].shape[0], shape=tuple(inp["x"].shape[1:]), eta=g["eta"],
                                                       conditioning=inp["y"], unconditional_conditioning=null_y, unconditional_guidance_scale=g["cfg_scale"],
                                                       model_kwargs=dict(mask=mask), x_T=inp["x"].clone(), normals_sequence=[torch.zeros_like(d) for d in g["draws"]])
    assert rel_l2(z, g["sample"]) > 1e-2


def test_sa_solver_25_steps_runs_and_is_finite():
    from pixart_sigma_amd.diffusion import SASolverSampler
    model = lambda x, t, y, **kw: 0.1 * x + 0.01 * y.mean()
    torch.manual_seed(0)
    s, _ = SASolverSampler(model, device="cpu").sample(S=25, batch_size=2, shape=(4, 8, 8), eta=1, conditioning=torch.ones(2, 1),
                                                       unconditional_conditioning=torch.zeros(2, 1), unconditional_guidance_scale=4.5)
    assert s.shape == (2, 4, 8, 8) and torch.isfinite(s).all()


def test_dpm_solver_20_steps_runs_and_is_finite():
    from pixart_sigma_amd.diffusion import DPMS
    model = lambda x, t, y, **kw: 0.1 * x + 0.01 * y.mean()
    z = torch.randn(2, 4, 8, 8, generator=torch.Generator().manual_seed(0))
    s = DPMS(model, condition=torch.ones(2, 1), uncondition=torch.zeros(2, 1), cfg_scale=4.5).sample(z, steps=20, order=2)
    assert torch.isfinite(s).all()
    # return_intermediate (reference model/dpm_solver.py:1175-1176, 1207-1234): (x0, [initial latent, x_t after every solver step]); the oracle's solver, stopped
    # after k steps of the same 20-step schedule, gives the same x_t
    from oracle import pixart_oracle as po
    s2, inter = DPMS(model, condition=torch.ones(2, 1), uncondition=torch.zeros(2, 1), cfg_scale=4.5).sample(z, steps=20, order=2, return_intermediate=True)
    assert torch.equal(s2, s) and len(inter) == 21 and torch.equal(inter[0], z) and torch.equal(inter[-1], s)
    eps_model = lambda x, t_in, c: 0.1 * x + 0.01 * c.mean()          # noqa: E731 - the same toy denoiser behind the oracle's (x, t, cond) interface
    for k in (1, 2, 7, 20):
        xk = po.dpm_solver_sample(eps_model, z, torch.ones(2, 1), torch.zeros(2, 1), 4.5, steps=20, order=2, stop_after=k)
        assert rel_l2(inter[k], xk) < 2e-5, k


def test_pos_table_matches_reference(golden):
    from pixart_sigma_amd.engine import sincos_pos_embed
    g = golden("tables")
    for (h, w, pe, base), ref in g["pos"].items():
        tab = sincos_pos_embed(1152, h, w, pe, base)
        assert np.array_equal(tab[:: max(1, (h * w) // 37)], ref.numpy())


def test_registry_and_state_dict_keys_match_reference():
    """Constructor surface + state-dict wire format (tools/convert_pixart_to_diffusers.py:29-155)."""
    from pixart_sigma_amd import MODELS, build_model
    kv = {"sampling": "conv", "scale_factor": 2, "kv_compress_layer": [1]}
    m = build_model("PixArtMS", depth=2, hidden_size=1152, num_heads=16, input_size=16, model_max_length=20, kv_compress_config=kv,
                    use_grad_checkpoint=True, use_fp32_attention=True, gc_step=1)
    assert MODELS.get("PixArtMS_XL_2") is not None
    cfg = po.OracleCfg(depth=2, input_size=16, model_max_length=20, kv_sampling="conv", kv_scale_factor=2, kv_layers=(1,))
    want = {k: tuple(s) for k, s in param_shapes(cfg).items()}
    want["pos_embed"] = (1, 64, 1152)
    got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert got == want
    assert m.grad_checkpointing and m.blocks[0].fp32_attention
    m.load_state_dict(make_state_dict(cfg, seed=0))
    # reference init invariants (PixArtMS.py:278-285, PixArt_blocks.py:86-88)
    m2 = build_model("PixArtMS", depth=1, hidden_size=1152, num_heads=16, input_size=16, kv_compress_config={"sampling": "conv", "scale_factor": 2, "kv_compress_layer": [0]})
    assert m2.final_layer.linear.weight.abs().max() == 0 and m2.blocks[0].cross_attn.proj.weight.abs().max() == 0
    assert torch.all(m2.blocks[0].attn.sr.weight == 0.25)


def test_fixed_resolution_pixart_class_and_qk_norm_keys(golden):
    """PixArt / PixArt_XL_2 registry names (PixArt.py:62,313): pos_embed is a real buffer of the wire format and equals the
    reference's table; qk_norm adds q_norm / k_norm LayerNorm parameters (PixArt_blocks.py:90-92)."""
    from pixart_sigma_amd import MODELS, build_model
    assert MODELS.get("PixArt") is not None and MODELS.get("PixArt_XL_2") is not None
    m = build_model("PixArt", depth=1, hidden_size=1152, num_heads=16, input_size=32, pe_interpolation=0.5, model_max_length=20, qk_norm=True)
    ref = golden("tables")["pos"][(16, 16, 0.5, 16)]
    assert np.array_equal(m.pos_embed[0].numpy().astype(np.float32)[:: max(1, 256 // 37)], ref.numpy().astype(np.float32))
    cfg = po.OracleCfg(depth=1, input_size=32, model_max_length=20, qk_norm=True)
    want = {k: tuple(s) for k, s in param_shapes(cfg).items()}
    want["pos_embed"] = (1, 256, 1152)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == want
    sd = make_state_dict(cfg, seed=0)
    with pytest.raises(RuntimeError):
        m.load_state_dict(sd)                      # this class's wire format includes pos_embed
    sd["pos_embed"] = m.pos_embed.clone()
    m.load_state_dict(sd)
    assert m.out_channels == 8 and build_model("PixArt", depth=1, hidden_size=1152, num_heads=16, pred_sigma=False).out_channels == 4


def test_xl2_parameter_count():
    from pixart_sigma_amd.model.nets import PixArtMS
    with torch.device("meta"):
        m = PixArtMS(depth=28, hidden_size=1152, patch_size=2, num_heads=16, input_size=128)
    assert sum(p.numel() for p in m.parameters()) == 610856096  # reference notebook known answer


def test_param_store_flat_views_cpu():
    from pixart_sigma_amd.engine import ParamStore
    lin1, lin2 = torch.nn.Linear(8, 6), torch.nn.Linear(6, 3)
    named = [("a.weight", lin1.weight), ("a.bias", lin1.bias), ("b.weight", lin2.weight), ("b.bias", lin2.bias)]
    w0 = lin1.weight.detach().clone()
    st = ParamStore(named, torch.device("cpu"))
    assert torch.equal(lin1.weight, w0) and lin1.weight.data_ptr() == st.master.data_ptr()
    assert st.range_of("a.") == (0, 128) and st.range_of("b.") == (128, 256)
    lin2(lin1(torch.ones(2, 8))).sum().backward()            # autograd accumulates into the flat views
    assert st.grad[: 48].abs().sum() > 0 and lin1.weight.grad.data_ptr() == st.grad.data_ptr()
    lin1.weight.grad = None
    assert st.attach_grads() and lin1.weight.grad.data_ptr() == st.grad.data_ptr()


def test_model_refuses_cpu_forward():
    from pixart_sigma_amd.model.nets import PixArtMS
    m = PixArtMS(depth=1, hidden_size=1152, num_heads=16, input_size=8, model_max_length=4)
    with pytest.raises(AssertionError, match="HIP kernels only"):
        m(torch.zeros(1, 4, 8, 8), torch.zeros(1), torch.zeros(1, 1, 4, 4096))


def test_came_tables_cover_every_parameter_exactly_once():
    """Host tables of pxa_came_step (dp.came_tables): tiles partition the rows / elements of every tensor, factored views follow
    came_pytorch (last two dims; leading dims are a batch of matrices), state offsets do not overlap, column state is 16-byte aligned."""
    from pixart_sigma_amd.dp import came_tables
    shapes = {"w": (300, 200), "b": (300,), "table": (6, 1152), "conv": (48, 4, 2, 2), "odd": (33, 7), "long": (70000,), "fc": (1152, 4608)}
    names, offset, numel, off = list(shapes), {}, {}, 0
    for n, s in shapes.items():
        numel[n] = int(torch.tensor(s).prod())
        offset[n] = off
        off += (numel[n] + 63) // 64 * 64
    tb = came_tables(names, offset, shapes, numel, tile_elems=4096)
    T = tb["tensors"]
    assert [t["factored"] for t in T] == [1, 0, 1, 1, 1, 0, 1]
    conv = T[names.index("conv")]
    assert (conv["batch"], conv["R"], conv["C"]) == (192, 2, 2)
    covered = {i: [] for i in range(len(T))}
    for ti, first, count in tb["tiles"]:
        covered[ti].append((first, count))
    for i, t in enumerate(T):
        total = t["batch"] * t["R"] if t["factored"] else t["C"]
        runs = sorted(covered[i])
        assert runs[0][0] == 0 and sum(c for _, c in runs) == total
        assert all(a[0] + a[1] == b[0] for a, b in zip(runs, runs[1:]))
        if t["factored"]:
            assert all(c * t["C"] <= max(4096, 4 * t["C"]) for _, c in runs) and t["col_off"] % 4 == 0
    rows = [(t["row_off"], t["batch"] * t["R"]) for t in T if t["factored"]]
    assert all(a[0] + a[1] == b[0] for a, b in zip(rows, rows[1:])) and rows[-1][0] + rows[-1][1] == tb["n_row"]
    assert len(tb["col_inv_r"]) == tb["n_col"] and tb["n_rm"] == sum(t["batch"] for t in T if t["factored"])
    assert tb["n_nf"] == 300 + 70000 and tb["col_inv_r"][T[0]["col_off"]] == 1.0 / 300


def test_came_tables_keep_scalar_path_tiles_short():
    """Tensors the CAME kernel walks on its scalar path (csrc/came.hip, MAXJ == 0: batched matrices, row lengths that are no multiple of 4 or longer than
    4,608) get tiles of at most 64 rows: a wave takes those rows one after the other, each a chain of dependent loads and atomics, and the patch-embed
    weight - [4608][2][2] as the package views it - used to be ONE tile of 9,216 rows that every pass of the step waited ~2 ms for
    (profiles/r03u_came_kernel_stats.csv: 9.7 -> 4.7 ms per step).  Vector-path tensors keep their large tiles."""
    from pixart_sigma_amd.dp import came_tables
    shapes = {"x_embedder.proj.weight": (1152, 4, 2, 2), "fc1": (4608, 1152), "odd": (4000, 7), "wide": (8, 4612)}
    names, offset, numel, off = list(shapes), {}, {}, 0
    for n, sh in shapes.items():
        numel[n] = int(torch.tensor(sh).prod())
        offset[n] = off
        off += numel[n]
    tb = came_tables(names, offset, shapes, numel, tile_elems=262144)
    rows_of = {i: [c for ti, _, c in tb["tiles"] if ti == i] for i in range(len(names))}
    assert max(rows_of[0]) <= 64 and sum(rows_of[0]) == 4608 * 2 and len(rows_of[0]) == 144        # conv weight: 144 tiles instead of 1
    assert max(rows_of[1]) == 224 and sum(rows_of[1]) == 4608                                         # C = 1152: 262,144 // 1152 rounded down to a multiple of 4
    assert max(rows_of[2]) <= 64 and sum(rows_of[2]) == 4000                                          # C % 4 != 0
    assert max(rows_of[3]) <= 64 and sum(rows_of[3]) == 8                                             # C > 4,608


def test_store_notices_parameters_moved_out_by_a_standalone_block(monkeypatch):
    """a block used stand-alone re-points ITS parameters into a private flat store; the model's store must notice (every
    parameter is checked, not just the first) and rebuild from the current values instead of training stale copies."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import fake_ops
    from pixart_sigma_amd import engine
    monkeypatch.setattr(engine, "ops", fake_ops)
    from pixart_sigma_amd.model.nets.PixArtMS import PixArtMS
    m = PixArtMS(depth=2, input_size=8, model_max_length=8)
    m._prepare(torch.device("cpu"))
    st = m._store
    assert st.owns_all(m._ordered_named_params())
    blk = m.blocks[1]
    blk._engine_for_standalone()                                  # moves blocks.1.* into the block's own store
    assert st.owns(m.x_embedder.proj.weight, "x_embedder.proj.weight")      # the first parameter alone would not have shown it
    assert not st.owns_all(m._ordered_named_params())
    with torch.no_grad():
        blk.attn.proj.weight.fill_(0.25)
    m._prepare(torch.device("cpu"))                               # what the next forward does
    assert m._store is not st and m._store.owns_all(m._ordered_named_params())
    assert float(m._store.f("blocks.1.attn.proj.weight").mean()) == 0.25


def test_iddpm_ancestral_sampler_host_math_matches_reference():
    """pixart_sigma_amd.diffusion.iddpm.SpacedDiffusion.p_sample_loop is host-side elementwise math around the denoiser call: with the oracle's
    forward_with_cfg standing in for the denoiser (CPU) it must reproduce the reference's 5-step chain of tests/golden/iddpm_d2.pt."""
    import os
    import torch
    from oracle import pixart_oracle as po
    from oracle.weights import make_inputs, make_state_dict
    from pixart_sigma_amd.diffusion.iddpm import IDDPM
    g = torch.load(os.path.join(os.path.dirname(__file__), "golden", "iddpm_d2.pt"), weights_only=False)
    cfg = po.OracleCfg(**g["cfg"])
    sd = make_state_dict(cfg, seed=g["weights_seed"])
    inp = make_inputs(seed=g["inputs_seed"], **g["inputs"])
    z = torch.cat([inp["x"][:2], inp["x"][:2]], dim=0)
    diff = IDDPM(str(g["steps"]))
    assert diff.timestep_map == [0, 250, 500, 749, 999] and diff.num_timesteps == 5
    seen = []

    def model(x, timestep, **kw):
        seen.append(int(timestep[0]))
        return po.forward_with_cfg(sd, cfg, x, timestep, inp["y"], g["cfg_scale"], inp["mask"])
    for key, clip in (("sample", False), ("sample_clip", True)):
        torch.manual_seed(g["noise_seed"])
        with torch.no_grad():
            out = diff.p_sample_loop(model, z.shape, z, clip_denoised=clip, device="cpu", step_noise=lambda x: torch.randn(x.shape))
        assert ((out - g[key]).norm() / g[key].norm()).item() < 5e-5, key
    assert seen[:5] == [999, 749, 500, 250, 0]          # the denoiser is called at the ORIGINAL timesteps (respace.py:128-134)


def test_keys_resident_attention_refuses_a_sample_longer_than_max_kv_len():
    """the keys-resident cross-attention kernels clamp to max_kv_len; the engine now hands its host copy of the lengths to ops.attention_fwd / _bwd,
    whose guard fires before anything is launched (so this needs no GPU)."""
    import pytest
    from pixart_sigma_amd import ops
    t = torch.zeros(1)
    with pytest.raises(AssertionError, match="max_kv_len 64 < longest sample 77"):
        ops.attention_fwd(t, t, t, t, t, 1, 1, 1, 64, ((0, 0, 0),) * 4, max_kv_len=64, kv_len_host=(12, 77))
    with pytest.raises(AssertionError, match="max_kv_len"):
        ops.attention_bwd(t, t, t, t, t, t, t, t, t, t, 1, 1, 1, 64, ((0, 0, 0),) * 4, ((0, 0, 0),) * 3, max_kv_len=64, kv_len_host=(65,))
    import inspect
    from pixart_sigma_amd import engine
    src = inspect.getsource(engine.Engine)
    assert src.count("kv_len_host=ctx.get(\"lens_host\")") == 2          # both cross-attention call sites pass the host lengths
