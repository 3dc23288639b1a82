"""TEST INFRASTRUCTURE ONLY — deterministic "random-init" weights shared bit-for-bit by the
reference (golden generation), the oracle and the HIP path.

The reference's own init (PixArtMS.py:250-285) zero-initialises cross_attn.proj and
final_layer.linear, so a freshly built model outputs exactly 0 and any parity test on it is
vacuous.  make_state_dict() therefore draws EVERY tensor (including the
zero-init ones and all biases) from a seeded CPU generator, tensor by tensor in sorted key
order, with magnitudes of the reference's init scheme:
  Linear weights ~ U(+-sqrt(6/(fan_in+fan_out))) (xavier), small-MLP/zero-init weights ~ N(0,0.02),
  biases ~ N(0,0.02), scale_shift_table ~ N(0,1)/sqrt(D), y_embedding ~ N(0,1)/sqrt(4096),
  KV-compress conv = 1/sr^2 + N(0,0.05), LN affine = 1 + N(0,0.05) / N(0,0.05).
"""
import math

import torch


def param_shapes(cfg, caption_channels=4096):
    D, p, C = cfg.hidden_size, cfg.patch_size, cfg.in_channels
    Dff = int(D * cfg.mlp_ratio)
    s = {
        "x_embedder.proj.weight": (D, C, p, p), "x_embedder.proj.bias": (D,),
        "t_embedder.mlp.0.weight": (D, 256), "t_embedder.mlp.0.bias": (D,),
        "t_embedder.mlp.2.weight": (D, D), "t_embedder.mlp.2.bias": (D,),
        "t_block.1.weight": (6 * D, D), "t_block.1.bias": (6 * D,),
        "y_embedder.y_embedding": (cfg.model_max_length, caption_channels),
        "y_embedder.y_proj.fc1.weight": (D, caption_channels), "y_embedder.y_proj.fc1.bias": (D,),
        "y_embedder.y_proj.fc2.weight": (D, D), "y_embedder.y_proj.fc2.bias": (D,),
        "final_layer.scale_shift_table": (2, D),
        "final_layer.linear.weight": (p * p * cfg.out_channels, D), "final_layer.linear.bias": (p * p * cfg.out_channels,),
    }
    if getattr(cfg, "micro_condition", False):               # SizeEmbedder(hidden_size // 3) x 2 (PixArtMS.py:141-143)
        d3 = D // 3
        for e in ("csize_embedder", "ar_embedder"):
            s.update({e + ".mlp.0.weight": (d3, 256), e + ".mlp.0.bias": (d3,), e + ".mlp.2.weight": (d3, d3), e + ".mlp.2.bias": (d3,)})
    for i in range(cfg.depth):
        b = f"blocks.{i}."
        s.update({
            b + "scale_shift_table": (6, D),
            b + "attn.qkv.weight": (3 * D, D), b + "attn.qkv.bias": (3 * D,),
            b + "attn.proj.weight": (D, D), b + "attn.proj.bias": (D,),
            b + "cross_attn.q_linear.weight": (D, D), b + "cross_attn.q_linear.bias": (D,),
            b + "cross_attn.kv_linear.weight": (2 * D, D), b + "cross_attn.kv_linear.bias": (2 * D,),
            b + "cross_attn.proj.weight": (D, D), b + "cross_attn.proj.bias": (D,),
            b + "mlp.fc1.weight": (Dff, D), b + "mlp.fc1.bias": (Dff,),
            b + "mlp.fc2.weight": (D, Dff), b + "mlp.fc2.bias": (D,),
        })
        if i in cfg.kv_layers and cfg.kv_scale_factor > 1 and cfg.kv_sampling == "conv":
            sr = cfg.kv_scale_factor
            s.update({b + "attn.sr.weight": (D, 1, sr, sr), b + "attn.sr.bias": (D,),
                      b + "attn.norm.weight": (D,), b + "attn.norm.bias": (D,)})
        if cfg.qk_norm:
            for n in ("q_norm", "k_norm"):
                s.update({b + f"attn.{n}.weight": (D,), b + f"attn.{n}.bias": (D,)})
    return s


def make_state_dict(cfg, seed=0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    D = cfg.hidden_size
    sd = {}
    for k, shp in sorted(param_shapes(cfg).items()):
        n = lambda std: torch.randn(shp, generator=g) * std
        if k.endswith("scale_shift_table"):
            v = n(1.0 / math.sqrt(D))
        elif k == "y_embedder.y_embedding":
            v = n(1.0 / math.sqrt(shp[1]))
        elif ".sr.weight" in k:
            v = 1.0 / (shp[-1] * shp[-2]) + n(0.05)
        elif ".norm.weight" in k or "_norm.weight" in k:
            v = 1.0 + n(0.05)
        elif ".norm.bias" in k or "_norm.bias" in k or ".sr.bias" in k:
            v = n(0.05)
        elif k.endswith(".bias"):
            v = n(0.02)
        elif any(t in k for t in ("t_embedder", "csize_embedder", "ar_embedder", "t_block", "y_proj", "cross_attn.proj", "final_layer.linear")):
            v = n(0.02)
        else:  # xavier-uniform on the (out, fan_in) matrix view
            fan_out, fan_in = shp[0], int(torch.tensor(shp[1:]).prod())
            a = math.sqrt(6.0 / (fan_in + fan_out))
            v = (torch.rand(shp, generator=g) * 2 - 1) * a
        sd[k] = v.to(dtype)
    return sd


def make_inputs(B, Hl, Wl, L, seed=1, lens=None, caption_channels=4096, in_channels=4):
    """Seeded synthetic batch: latent, caption features, mask, timesteps."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, in_channels, Hl, Wl, generator=g)
    y = torch.randn(B, 1, L, caption_channels, generator=g)
    t = torch.randint(0, 1000, (B,), generator=g)
    noise = torch.randn(B, in_channels, Hl, Wl, generator=g)
    mask = torch.ones(B, L, dtype=torch.int64)
    if lens is not None:
        for b, n in enumerate(lens):
            mask[b, n:] = 0
    return {"x": x, "y": y, "t": t, "noise": noise, "mask": mask}
