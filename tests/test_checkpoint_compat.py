"""Checkpoint wire-format mapping: reference `.pth` keys <-> the diffusers layout written by the
reference's tools/convert_pixart_to_diffusers.py:29-155.  CPU only."""
import pytest
import torch

from pixart_sigma_amd import PixArtMS
from pixart_sigma_amd.model.checkpoint_compat import from_diffusers, to_diffusers


@pytest.mark.parametrize("kw", [dict(), dict(micro_condition=True), dict(qk_norm=True)])
def test_roundtrip_and_converter_layout(kw):
    torch.manual_seed(0)
    m = PixArtMS(depth=2, input_size=8, model_max_length=12, **kw)
    sd = {k: torch.randn_like(v) for k, v in m.state_dict().items()}
    d = to_diffusers({"state_dict": sd})
    # the key set the reference converter produces for these options (converter lines 29-155)
    assert d["pos_embed.proj.weight"].shape == (1152, 4, 2, 2) and d["adaln_single.linear.weight"].shape == (6 * 1152, 1152)
    assert d["transformer_blocks.1.attn1.to_k.weight"].shape == (1152, 1152) and d["transformer_blocks.0.attn2.to_v.bias"].shape == (1152,)
    assert d["transformer_blocks.0.ff.net.0.proj.weight"].shape == (4608, 1152) and d["proj_out.weight"].shape == (32, 1152)
    assert d["scale_shift_table"].shape == (2, 1152) and d["transformer_blocks.0.scale_shift_table"].shape == (6, 1152)
    assert ("adaln_single.emb.resolution_embedder.linear_1.weight" in d) == bool(kw.get("micro_condition"))
    assert ("transformer_blocks.0.attn1.q_norm.weight" in d) == bool(kw.get("qk_norm"))
    assert not any(k.startswith(("pos_embed.", "y_embedder")) and "proj" not in k for k in d)
    # q / k / v are the row chunks of the fused projection, in that order (converter line 89); k / v of kv_linear (line 135)
    q, k, v = sd["blocks.1.attn.qkv.weight"].chunk(3)
    assert torch.equal(d["transformer_blocks.1.attn1.to_q.weight"], q) and torch.equal(d["transformer_blocks.1.attn1.to_v.weight"], v)
    assert torch.equal(d["transformer_blocks.0.attn2.to_k.bias"], sd["blocks.0.cross_attn.kv_linear.bias"].chunk(2)[0])
    back = from_diffusers(d)
    dropped = {"pos_embed", "y_embedder.y_embedding"}
    assert set(back) == set(sd) - dropped
    assert all(torch.equal(back[k], sd[k]) for k in back)
    missing, unexpected = m.load_state_dict(back, strict=False)
    assert set(missing) <= dropped and not unexpected


def test_kv_compression_weights_are_refused():
    m = PixArtMS(depth=2, input_size=8, model_max_length=12, kv_compress_config={"sampling": "conv", "scale_factor": 2, "kv_compress_layer": [1]})
    with pytest.raises(KeyError):
        to_diffusers(m.state_dict())
