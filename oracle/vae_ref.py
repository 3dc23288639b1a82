"""TEST INFRASTRUCTURE ONLY - CPU restatement (plain torch, fp32) of the VAE the reference calls.

**Parity unpinned.**  The reference never defines this network: it imports `diffusers.models.AutoencoderKL`
(`train_scripts/train.py:17,85,353`, `scripts/inference.py:193-196`) from an UNPINNED dependency
(`requirements.txt:2`: `git+https://github.com/huggingface/diffusers`) and loads hub weights
(`PixArt-alpha/pixart_sigma_sdxlvae_T5_diffusers/vae`, `sd-vae-ft-ema`).  diffusers is not installed in this image, the
weights are not on disk and the reference holds no test or golden vector for it.  What follows restates
the published architecture of that class for the public SDXL-VAE / SD-VAE config:

    in/out_channels 3, latent_channels 4, block_out_channels (128, 256, 512, 512), layers_per_block 2,
    down_block_types 4 x DownEncoderBlock2D, up_block_types 4 x UpDecoderBlock2D, norm_num_groups 32, act_fn silu,
    mid-block attention with ONE head (attention_head_dim = 512), GroupNorm eps 1e-6, resnet output_scale_factor 1,
    scaling_factor 0.13025 (SDXL-VAE) / 0.18215 (SD-VAE)

with diffusers' state-dict key names, so a real checkpoint would load into it unchanged.  Call sites the product path has
to serve: `vae.encode(x).latent_dist.sample()` (`train.py:149-153`), `vae.decode(z / vae.config.scaling_factor).sample`
(`inference.py:136`, `train.py:88`).

Only tests/, __graft_entry__.smoke() and bench/tool baselines may import this module.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, groups=32, eps=1e-6):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))        # dropout p = 0
        return (x if self.conv_shortcut is None else self.conv_shortcut(x)) + h


class Attention(nn.Module):
    """Mid-block self-attention over the H*W pixels: one head of width C, residual connection, GroupNorm in front."""

    def __init__(self, c, groups=32, eps=1e-6):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, c, eps=eps)
        self.to_q, self.to_k, self.to_v = nn.Linear(c, c), nn.Linear(c, c), nn.Linear(c, c)
        self.to_out = nn.ModuleList([nn.Linear(c, c), nn.Identity()])   # [Linear, Dropout(0)]

    def forward(self, x):
        B, C, H, W = x.shape
        t = self.group_norm(x).view(B, C, H * W).transpose(1, 2)
        q, k, v = self.to_q(t), self.to_k(t), self.to_v(t)
        p = torch.softmax(q @ k.transpose(1, 2) * C ** -0.5, dim=-1)
        o = self.to_out[0](p @ v)
        return x + o.transpose(1, 2).reshape(B, C, H, W)


class Downsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=0)

    def forward(self, x):
        return self.conv(F.pad(x, (0, 1, 0, 1)))


class Upsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class Block(nn.Module):
    def __init__(self, cin, cout, n_res, down=False, up=False):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout) for i in range(n_res)])
        if down:
            self.downsamplers = nn.ModuleList([Downsample2D(cout)])
        if up:
            self.upsamplers = nn.ModuleList([Upsample2D(cout)])

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        for m in getattr(self, "downsamplers", []):
            x = m(x)
        for m in getattr(self, "upsamplers", []):
            x = m(x)
        return x


class MidBlock(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.attentions = nn.ModuleList([Attention(c)])
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c), ResnetBlock2D(c, c)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class Encoder(nn.Module):
    def __init__(self, cin, latent, chans, layers):
        super().__init__()
        self.conv_in = nn.Conv2d(cin, chans[0], 3, padding=1)
        self.down_blocks = nn.ModuleList()
        c = chans[0]
        for i, co in enumerate(chans):
            self.down_blocks.append(Block(c, co, layers, down=i < len(chans) - 1))
            c = co
        self.mid_block = MidBlock(c)
        self.conv_norm_out = nn.GroupNorm(32, c, eps=1e-6)
        self.conv_out = nn.Conv2d(c, 2 * latent, 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for b in self.down_blocks:
            x = b(x)
        x = self.mid_block(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class Decoder(nn.Module):
    def __init__(self, latent, cout, chans, layers):
        super().__init__()
        rev = list(reversed(chans))
        self.conv_in = nn.Conv2d(latent, rev[0], 3, padding=1)
        self.mid_block = MidBlock(rev[0])
        self.up_blocks = nn.ModuleList()
        c = rev[0]
        for i, co in enumerate(rev):
            self.up_blocks.append(Block(c, co, layers + 1, up=i < len(rev) - 1))
            c = co
        self.conv_norm_out = nn.GroupNorm(32, c, eps=1e-6)
        self.conv_out = nn.Conv2d(c, cout, 3, padding=1)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for b in self.up_blocks:
            x = b(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class AutoencoderKLRef(nn.Module):
    def __init__(self, in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512), layers_per_block=2):
        super().__init__()
        self.encoder = Encoder(in_channels, latent_channels, block_out_channels, layers_per_block)
        self.decoder = Decoder(latent_channels, out_channels, block_out_channels, layers_per_block)
        self.quant_conv = nn.Conv2d(2 * latent_channels, 2 * latent_channels, 1)
        self.post_quant_conv = nn.Conv2d(latent_channels, latent_channels, 1)

    def encode_moments(self, x):
        """(mean, logvar) of the posterior, each (B, latent, H/8, W/8); logvar clamped to [-30, 20] like DiagonalGaussianDistribution."""
        mean, logvar = self.quant_conv(self.encoder(x)).chunk(2, dim=1)
        return mean, logvar.clamp(-30.0, 20.0)

    def decode(self, z):
        return self.decoder(self.post_quant_conv(z))


def randomize_(model, seed=0):
    """Random-init weights in the spirit of the rest of the oracle: default torch init, but GroupNorm affines and biases get noise
    so that no term of the arithmetic is trivially 0 / 1."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if "norm" in name and name.endswith("weight"):
                p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
            elif name.endswith("bias"):
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
            else:
                fan_in = p[0].numel()
                p.copy_(torch.randn(p.shape, generator=g) * fan_in ** -0.5)
    return model
