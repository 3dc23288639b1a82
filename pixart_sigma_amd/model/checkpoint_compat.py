"""Checkpoint wire formats of the denoiser: the reference's `.pth` state dict <-> the diffusers `Transformer2DModel` key layout
that `tools/convert_pixart_to_diffusers.py:29-155` of the reference writes (and that the published
`PixArt-alpha/PixArt-Sigma-XL-2-*-MS` `transformer/` folders use).  Table-driven and bidirectional, so a diffusers-format
checkpoint loads into `PixArtMS` and a trained model exports back; fused projections are split / concatenated on the way
(`attn.qkv` rows are [q; k; v], `cross_attn.kv_linear` rows are [k; v]: converter lines 89, 135).
Keys that only one side has: the reference drops `pos_embed` and `y_embedder.y_embedding` on export; on import they are left
to the model's own buffers (load with strict=False)."""
import re

import torch

# reference key (regex on the part after an optional "blocks.{i}.")  ->  diffusers key
_TOP = [
    ("x_embedder.proj", "pos_embed.proj"),
    ("y_embedder.y_proj.fc1", "caption_projection.linear_1"),
    ("y_embedder.y_proj.fc2", "caption_projection.linear_2"),
    ("t_embedder.mlp.0", "adaln_single.emb.timestep_embedder.linear_1"),
    ("t_embedder.mlp.2", "adaln_single.emb.timestep_embedder.linear_2"),
    ("csize_embedder.mlp.0", "adaln_single.emb.resolution_embedder.linear_1"),
    ("csize_embedder.mlp.2", "adaln_single.emb.resolution_embedder.linear_2"),
    ("ar_embedder.mlp.0", "adaln_single.emb.aspect_ratio_embedder.linear_1"),
    ("ar_embedder.mlp.2", "adaln_single.emb.aspect_ratio_embedder.linear_2"),
    ("t_block.1", "adaln_single.linear"),
    ("final_layer.linear", "proj_out"),
]
_BLOCK = [
    ("attn.proj", "attn1.to_out.0"),
    ("attn.q_norm", "attn1.q_norm"),
    ("attn.k_norm", "attn1.k_norm"),
    ("mlp.fc1", "ff.net.0.proj"),
    ("mlp.fc2", "ff.net.2"),
    ("cross_attn.q_linear", "attn2.to_q"),
    ("cross_attn.proj", "attn2.to_out.0"),
]
_SPLIT = [  # fused reference tensor -> diffusers parts (split along dim 0)
    ("attn.qkv", ("attn1.to_q", "attn1.to_k", "attn1.to_v")),
    ("cross_attn.kv_linear", ("attn2.to_k", "attn2.to_v")),
]
_DROPPED = ("pos_embed", "y_embedder.y_embedding")


def to_diffusers(state_dict):
    """Reference `.pth` state dict (the dict under "state_dict", or the bare dict) -> diffusers Transformer2DModel keys."""
    sd = dict(state_dict.get("state_dict", state_dict))
    out = {}
    for k in _DROPPED:
        sd.pop(k, None)
    for src, dst in _TOP:
        for suffix in ("weight", "bias"):
            if f"{src}.{suffix}" in sd:
                out[f"{dst}.{suffix}"] = sd.pop(f"{src}.{suffix}")
    if "final_layer.scale_shift_table" in sd:
        out["scale_shift_table"] = sd.pop("final_layer.scale_shift_table")
    for k in [k for k in sd if k.startswith("blocks.")]:
        if k not in sd:
            continue
        _, i, rest = k.split(".", 2)
        pre = f"transformer_blocks.{i}."
        if rest == "scale_shift_table":
            out[pre + rest] = sd.pop(k)
            continue
        name, suffix = rest.rsplit(".", 1)
        hit = dict(_BLOCK).get(name)
        if hit is not None:
            out[f"{pre}{hit}.{suffix}"] = sd.pop(k)
            continue
        parts = dict(_SPLIT).get(name)
        if parts is not None:
            for p, chunk in zip(parts, torch.chunk(sd.pop(k), len(parts), dim=0)):
                out[f"{pre}{p}.{suffix}"] = chunk
    if sd:
        raise KeyError(f"to_diffusers: no mapping for {sorted(sd)[:8]} (KV-compression `attn.sr` / `attn.norm` have no diffusers counterpart)")
    return out


def from_diffusers(state_dict):
    """diffusers Transformer2DModel state dict -> reference keys (load into PixArtMS with strict=False: the `pos_embed` /
    `y_embedding` buffers are not part of the diffusers format)."""
    sd = dict(state_dict)
    out = {}
    for src, dst in _TOP:
        for suffix in ("weight", "bias"):
            if f"{dst}.{suffix}" in sd:
                out[f"{src}.{suffix}"] = sd.pop(f"{dst}.{suffix}")
    if "scale_shift_table" in sd:
        out["final_layer.scale_shift_table"] = sd.pop("scale_shift_table")
    depth = 1 + max([int(m.group(1)) for k in sd if (m := re.match(r"transformer_blocks\.(\d+)\.", k))], default=-1)
    for i in range(depth):
        pre, dst = f"transformer_blocks.{i}.", f"blocks.{i}."
        if pre + "scale_shift_table" in sd:
            out[dst + "scale_shift_table"] = sd.pop(pre + "scale_shift_table")
        for name, hit in _BLOCK:
            for suffix in ("weight", "bias"):
                if f"{pre}{hit}.{suffix}" in sd:
                    out[f"{dst}{name}.{suffix}"] = sd.pop(f"{pre}{hit}.{suffix}")
        for name, parts in _SPLIT:
            for suffix in ("weight", "bias"):
                keys = [f"{pre}{p}.{suffix}" for p in parts]
                if all(k in sd for k in keys):
                    out[f"{dst}{name}.{suffix}"] = torch.cat([sd.pop(k) for k in keys], dim=0)
    if sd:
        raise KeyError(f"from_diffusers: no mapping for {sorted(sd)[:8]}")
    return out
