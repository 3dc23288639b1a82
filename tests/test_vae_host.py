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
