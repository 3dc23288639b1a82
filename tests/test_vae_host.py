"""CPU-side checks of the VAE row: the oracle restatement has the published AutoencoderKL parameter
inventory, the product module exposes the same state-dict keys (diffusers' names) and the diffusers-facing plumbing
(config, save / load, deprecated attention keys, DiagonalGaussianDistribution) behaves like the class the reference imports."""
import pytest
import torch

from oracle.vae_ref import AutoencoderKLRef, randomize_
from pixart_sigma_amd.vae import AutoencoderKL, DiagonalGaussianDistribution


def test_oracle_matches_published_parameter_count_and_shapes():
    ref = AutoencoderKLRef()
    assert sum(p.numel() for p in ref.parameters()) == 83_653_863      # SD / SDXL AutoencoderKL (public model card: 83.7 M)
    sd = ref.state_dict()
    assert sd["encoder.conv_in.weight"].shape == (128, 3, 3, 3) and sd["decoder.conv_out.weight"].shape == (3, 128, 3, 3)
    assert sd["encoder.conv_out.weight"].shape == (8, 512, 3, 3) and sd["quant_conv.weight"].shape == (8, 8, 1, 1)
    assert sd["decoder.up_blocks.2.resnets.0.conv_shortcut.weight"].shape == (256, 512, 1, 1)
    assert sd["decoder.mid_block.attentions.0.to_q.weight"].shape == (512, 512)
    assert "decoder.up_blocks.3.upsamplers.0.conv.weight" not in sd and "decoder.up_blocks.2.upsamplers.0.conv.weight" in sd
    assert "encoder.down_blocks.3.downsamplers.0.conv.weight" not in sd and len([k for k in sd if ".resnets." in k and k.endswith("conv1.weight")]) == 8 + 12 + 4


def test_oracle_shapes_and_downsample_convention():
    ref = randomize_(AutoencoderKLRef(block_out_channels=(128, 256), layers_per_block=1))
    x = torch.randn(2, 3, 16, 24)
    mean, logvar = ref.encode_moments(x)
    assert mean.shape == (2, 4, 8, 12) and logvar.shape == mean.shape
    assert ref.decode(mean).shape == (2, 3, 16, 24)
    # Downsample2D = pad (0,1,0,1) + stride-2 conv without padding: output (y, x) only sees inputs at rows 2y..2y+2, cols 2x..2x+2
    d = ref.encoder.down_blocks[0].downsamplers[0]
    t = torch.zeros(1, 128, 8, 8)
    t[0, :, 0, 0] = 1.0
    out = d(t) - d(torch.zeros_like(t))
    assert out[0, :, 0, 0].abs().sum() > 0 and out[0, :, 1:, :].abs().sum() == 0 and out[0, :, :, 1:].abs().sum() == 0


def _delta_conv(conv, tap):
    """weights of an (identity-over-channels, single-tap) convolution: out[:, c] = in[:, c] shifted by the tap; zero bias."""
    with torch.no_grad():
        conv.weight.zero_()
        conv.bias.zero_()
        for c in range(conv.weight.shape[0]):
            conv.weight[c, c, tap[0], tap[1]] = 1.0


def test_known_answers_of_the_restatement_resampling_and_attention():
    """known-answer checks of oracle/vae_ref.py that do not lean on its author's reading of the architecture - the expected tensors are written
    out by hand from the PUBLISHED definitions (diffusers Downsample2D: pad right / bottom by one, then 3x3 stride 2 without padding; Upsample2D: nearest 2x, then
    3x3 padding 1; Attention of the mid block: GroupNorm, one head of width C, softmax(q k^T / sqrt(C)) over the KEY axis, output projection, residual)."""
    from oracle.vae_ref import Attention, Downsample2D, Upsample2D
    C = 32
    x = torch.arange(1.0, 17.0).view(1, 1, 4, 4).repeat(1, C, 1, 1) * torch.linspace(1.0, 2.0, C).view(1, C, 1, 1)   # x[c, y, x] = (4 y + x + 1) * s_c
    v = lambda y, xx: (4 * y + xx + 1) if (0 <= y < 4 and 0 <= xx < 4) else 0.0                                       # noqa: E731 - channel 0 (s = 1); 0 outside
    # ---- Downsample2D, delta tap (ky, kx): out[y, x] = in[2 y + ky, 2 x + kx], zero where that falls on the ONE padded row / column (bottom / right)
    d = Downsample2D(C)
    for tap in ((0, 0), (2, 2), (1, 2), (2, 0)):
        _delta_conv(d.conv, tap)
        out = d(x)
        assert out.shape == (1, C, 2, 2)
        want = torch.tensor([[v(2 * y + tap[0], 2 * xx + tap[1]) for xx in range(2)] for y in range(2)])
        assert torch.equal(out[0, 0], want), (tap, out[0, 0], want)
        assert torch.allclose(out[0, C - 1], want * 2.0)
    # ---- Upsample2D, delta tap: out[Y, X] = up[Y + ky - 1, X + kx - 1] with up[Y, X] = in[Y // 2, X // 2], zero outside the 8 x 8 upsampled image
    u = Upsample2D(C)
    for tap in ((1, 1), (0, 0), (2, 1), (1, 2)):
        _delta_conv(u.conv, tap)
        out = u(x)
        assert out.shape == (1, C, 8, 8)
        up = lambda Y, X: v(Y // 2, X // 2) if (0 <= Y < 8 and 0 <= X < 8) else 0.0                                     # noqa: E731
        want = torch.tensor([[up(Y + tap[0] - 1, X + tap[1] - 1) for X in range(8)] for Y in range(8)])
        assert torch.equal(out[0, 0], want), tap
    # ---- Attention: x pre-normalised per GroupNorm group (two pixel classes +p / -p, every 16-channel... here 1 group of 32 channels: 16 of +1, 16 of -1), so that
    # GroupNorm (weight 1, bias 0) is the identity up to 1 / sqrt(1 + eps); to_v = to_out = I
    a = Attention(C, groups=1)
    with torch.no_grad():
        for lin in (a.to_q, a.to_k, a.to_v, a.to_out[0]):
            lin.weight.zero_()
            lin.bias.zero_()
        a.to_v.weight.copy_(torch.eye(C))
        a.to_out[0].weight.copy_(torch.eye(C))
    pat = torch.tensor([1.0, -1.0] * (C // 2))
    t = torch.stack([pat, pat, -pat, pat * 0 + pat, -pat, -pat], dim=1).view(1, C, 2, 3)            # pixels 0, 1, 3 of class +p; 2, 4, 5 of class -p: mean 0, variance 1
    r = 1.0 / (1.0 + 1e-6) ** 0.5
    out = a(t)                                                                                       # q = k = 0: uniform attention -> every pixel gets the MEAN value = 0
    assert torch.allclose(out, t, atol=1e-6)
    with torch.no_grad():                                                                            # q = k = alpha * GN(x): scores +-alpha^2 r^2 C / sqrt(C); large alpha -> a pixel
        a.to_q.weight.copy_(torch.eye(C) * 3.0)                                                      # attends to its own class only, uniformly -> value = its own GN(x)
        a.to_k.weight.copy_(torch.eye(C) * 3.0)
    out = a(t)
    assert torch.allclose(out, t * (1.0 + r), atol=1e-5)                                             # residual + attended value
    s_same, s_other = 9.0 * r * r * C / C ** 0.5, -9.0 * r * r * C / C ** 0.5                        # and with a finite alpha the softmax weights are the hand-computed ones:
    with torch.no_grad():
        a.to_q.weight.copy_(torch.eye(C) * 0.3)
        a.to_k.weight.copy_(torch.eye(C) * 0.3)
    import math
    e_same, e_other = math.exp(s_same / 100.0), math.exp(s_other / 100.0)                            # 0.3^2 = 9 / 100
    w_same = 3 * e_same / (3 * e_same + 3 * e_other)                                                 # three pixels per class; softmax over the KEY axis
    assert torch.allclose(a(t), t * (1.0 + r * (2 * w_same - 1.0)), atol=1e-5)


def test_product_module_has_diffusers_state_dict_and_config():
    vae, ref = AutoencoderKL(), AutoencoderKLRef()
    a, b = vae.state_dict(), ref.state_dict()
    assert list(a.keys()) == list(b.keys()) and all(a[k].shape == b[k].shape for k in a)
    assert vae.config.scaling_factor == 0.13025 and AutoencoderKL(scaling_factor=0.18215).config.scaling_factor == 0.18215
    assert not any(p.requires_grad for p in vae.parameters())


def test_save_load_roundtrip_and_deprecated_attention_keys(tmp_path):
    vae = AutoencoderKL(block_out_channels=(128, 256), layers_per_block=1, scaling_factor=0.18215)
    randomize_(vae, seed=3)
    vae.save_pretrained(str(tmp_path))
    back = AutoencoderKL.from_pretrained(str(tmp_path), torch_dtype=torch.float16)
    assert back.config.scaling_factor == 0.18215 and back.config.block_out_channels == (128, 256) and back.dtype == torch.float16
    for k, v in vae.state_dict().items():
        assert torch.equal(back.state_dict()[k].float(), v.half().float()), k
    old = {}
    for k, v in vae.state_dict().items():                     # checkpoint written by an old diffusers: query / key / value / proj_attn as 1x1 convs
        for new, o in (("to_q", "query"), ("to_k", "key"), ("to_v", "value"), ("to_out.0", "proj_attn")):
            if f".{new}." in k:
                k, v = k.replace(f".{new}.", f".{o}."), (v[:, :, None, None] if v.dim() == 2 else v)
        old[k] = v
    fresh = AutoencoderKL(block_out_channels=(128, 256), layers_per_block=1)
    fresh.load_state_dict(old)
    assert all(torch.equal(fresh.state_dict()[k], v) for k, v in vae.state_dict().items())


def test_diagonal_gaussian_distribution():
    m = torch.randn(2, 8, 4, 4)
    m[:, 4:] *= 30
    d = DiagonalGaussianDistribution(m)
    assert torch.equal(d.mode(), m[:, :4]) and d.logvar.min() >= -30 and d.logvar.max() <= 20
    g = torch.Generator().manual_seed(1)
    s = d.sample(generator=g)
    g = torch.Generator().manual_seed(1)
    assert torch.allclose(s, d.mean + d.std * torch.randn(d.mean.shape, generator=g))
    assert d.kl().shape == (2,)


def test_vae_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(AssertionError):
        AutoencoderKL(block_out_channels=(128, 256), layers_per_block=1).decode(torch.zeros(1, 4, 4, 4))


def test_conv_weight_packing_follows_the_abi_k_order():
    """The B operand of the implicit convolution (include/pixart_hip.h, pxa_gemm_args.k_tap): K ordered [Cin/64][ky][kx][64] for
    the stride-1 convolutions with Cin % 64 == 0; patch-matrix order [ky][kx][Cin (padded to 8)] for the stems and the stride-2 ones."""
    vae = AutoencoderKL(block_out_channels=(128, 256), layers_per_block=1)
    randomize_(vae, seed=1)
    vae._prepare()                                             # pure tensor plumbing: runs on the CPU
    conv = vae.decoder.up_blocks[0].resnets[0].conv1           # 256 -> 256, stride 1: tap-interleaved
    w, b, co = vae._packed[id(conv)]
    assert w.shape == (256, 9 * 256) and co == 256 and b.shape == (256,)
    ref = conv.weight.detach()
    for k in (0, 63, 64, 200, 575, 576, 1000, 2303):
        cc, r = divmod(k, 576)
        ky, r = divmod(r, 192)
        kx, c = divmod(r, 64)
        assert torch.equal(w[:, k].float(), ref[:, cc * 64 + c, ky, kx].to(w.dtype).float()), k
    stem = vae.decoder.conv_in                                 # 4 -> 256: explicit patch matrix, channels padded to 8
    w, b, co = vae._packed[id(stem)]
    assert w.shape == (256, 72)
    for k in (0, 3, 4, 8, 35, 71):
        tap, c = divmod(k, 8)
        want = stem.weight.detach()[:, c, tap // 3, tap % 3] if c < 4 else torch.zeros(256)
        assert torch.equal(w[:, k].float(), want.to(w.dtype).float()), k
    down = vae.encoder.down_blocks[0].downsamplers[0].conv     # stride 2: patch-matrix order
    w, _, _ = vae._packed[id(down)]
    assert torch.equal(w[:, 128 * 5 + 7].float(), down.weight.detach()[:, 7, 1, 2].to(w.dtype).float())
    qkv = vae._packed[("qkv", id(vae.decoder.mid_block.attentions[0]))]
    assert qkv[0].shape == (3 * 256, 256) and qkv[2] == 3 * 256


def test_analytic_decoder_flop_count_equals_the_hooked_count():
    """tools/vae_layer_table.py:decode_schedule (what tools/bench_vae.py, tools/bench_dmd.py and bench.py's `configs` leg price the decoder with, so that no
    product-side measurement imports oracle/) counts exactly the 2 m n k that forward hooks on the restated decoder count; the phase-decomposed upsampling
    convolutions keep the reference's algorithmic count."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from bench_vae import conv_flops
    from vae_layer_table import decode_schedule
    for px in (256, 512):
        hooked = conv_flops(AutoencoderKLRef().to("meta"), px)
        assert sum(e[2] for e in decode_schedule(1, px)) == hooked
        assert abs(sum(e[2] for e in decode_schedule(1, px, phases=True)) - hooked) < 1e-6 * hooked
        assert sum(e[2] for e in decode_schedule(3, px)) == 3 * hooked
