"""`AutoencoderKL` with the interface the reference uses from diffusers (un-vendored, unpinned dependency):

    vae = AutoencoderKL.from_pretrained(path).to(device).to(dtype)            train.py:85,353; inference.py:193-196
    posterior = vae.encode(images).latent_dist; z = posterior.sample()        train.py:149-153
    images = vae.decode(latents / vae.config.scaling_factor).sample           inference.py:136; train.py:88

The parameters live in nn.Conv2d / nn.GroupNorm / nn.Linear containers under diffusers' state-dict names (so its checkpoints load
unchanged); the containers' own forwards are never called.  Compute: bf16 NHWC pixel grids, every convolution a pxa_gemm (3x3
stride 1: implicit GEMM over the zero-padded grid, see include/pixart_hip.h), GroupNorm/SiLU/upsampling/residual/softmax kernels
from csrc/vae.hip.  Forward only: the reference keeps the VAE frozen under torch.no_grad().  There is no CPU / eager fallback -
without libpixart_hip.so and a GPU every call raises.
"""
import json
import os
from types import SimpleNamespace

import torch
import torch.nn as nn

from .. import ops
from ..ops import BF16, F32, Grid


def _c8(n):
    return (n + 7) // 8 * 8


def _img_rows(H, W):
    """Pixel slots per image of a zero-bordered (H+2) x (W+2) grid, rounded up to the GEMM's 256-row tile: every output tile of an
    implicit convolution then belongs to one image (the GroupNorm statistics of its epilogue need that) and M has no partial tile."""
    return ((H + 2) * (W + 2) + 255) // 256 * 256


class DiagonalGaussianDistribution:
    """diffusers.models.autoencoders.vae.DiagonalGaussianDistribution: moments (B, 2*latent, h, w) -> mean / logvar halves."""

    def __init__(self, parameters, deterministic=False):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.deterministic = deterministic
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)
        if deterministic:
            self.var = self.std = torch.zeros_like(self.mean)

    def sample(self, generator=None):
        noise = torch.randn(self.mean.shape, generator=generator, device=self.mean.device, dtype=self.mean.dtype)
        return self.mean + self.std * noise

    def mode(self):
        return self.mean

    def kl(self, other=None):
        if self.deterministic:
            return torch.zeros(1, device=self.mean.device)
        if other is None:
            return 0.5 * torch.sum(self.mean ** 2 + self.var - 1.0 - self.logvar, dim=[1, 2, 3])
        return 0.5 * torch.sum((self.mean - other.mean) ** 2 / other.var + self.var / other.var - 1.0 - self.logvar + other.logvar, dim=[1, 2, 3])


class _Res(nn.Module):
    def __init__(self, cin, cout, groups, eps):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None


class _Attn(nn.Module):
    def __init__(self, c, groups, eps):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, c, eps=eps)
        self.to_q, self.to_k, self.to_v = nn.Linear(c, c), nn.Linear(c, c), nn.Linear(c, c)
        self.to_out = nn.ModuleList([nn.Linear(c, c), nn.Identity()])


class _Resample(nn.Module):
    def __init__(self, c, stride):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=stride, padding=0 if stride == 2 else 1)


class _Block(nn.Module):
    def __init__(self, cin, cout, n_res, groups, eps, down=False, up=False):
        super().__init__()
        self.resnets = nn.ModuleList([_Res(cin if i == 0 else cout, cout, groups, eps) for i in range(n_res)])
        if down:
            self.downsamplers = nn.ModuleList([_Resample(cout, 2)])
        if up:
            self.upsamplers = nn.ModuleList([_Resample(cout, 1)])


class _Mid(nn.Module):
    def __init__(self, c, groups, eps):
        super().__init__()
        self.attentions = nn.ModuleList([_Attn(c, groups, eps)])
        self.resnets = nn.ModuleList([_Res(c, c, groups, eps), _Res(c, c, groups, eps)])


class _Encoder(nn.Module):
    def __init__(self, cin, latent, chans, layers, groups, eps):
        super().__init__()
        self.conv_in = nn.Conv2d(cin, chans[0], 3, padding=1)
        self.down_blocks = nn.ModuleList()
        c = chans[0]
        for i, co in enumerate(chans):
            self.down_blocks.append(_Block(c, co, layers, groups, eps, down=i < len(chans) - 1))
            c = co
        self.mid_block = _Mid(c, groups, eps)
        self.conv_norm_out = nn.GroupNorm(groups, c, eps=eps)
        self.conv_out = nn.Conv2d(c, 2 * latent, 3, padding=1)


class _Decoder(nn.Module):
    def __init__(self, latent, cout, chans, layers, groups, eps):
        super().__init__()
        rev = list(reversed(chans))
        self.conv_in = nn.Conv2d(latent, rev[0], 3, padding=1)
        self.mid_block = _Mid(rev[0], groups, eps)
        self.up_blocks = nn.ModuleList()
        c = rev[0]
        for i, co in enumerate(rev):
            self.up_blocks.append(_Block(c, co, layers + 1, groups, eps, up=i < len(rev) - 1))
            c = co
        self.conv_norm_out = nn.GroupNorm(groups, c, eps=eps)
        self.conv_out = nn.Conv2d(c, cout, 3, padding=1)


# deprecated attention-block key names of older diffusers checkpoints (sd-vae-ft-ema ships these)
_OLD_ATTN_KEYS = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}


class AutoencoderKL(nn.Module):
    config_name = "config.json"

    def __init__(self, in_channels=3, out_channels=3, down_block_types=None, up_block_types=None,
                 block_out_channels=(128, 256, 512, 512), layers_per_block=2, act_fn="silu", latent_channels=4, norm_num_groups=32,
                 sample_size=1024, scaling_factor=0.13025, force_upcast=True, **kwargs):
        super().__init__()
        assert act_fn == "silu", "AutoencoderKL: only the silu activation of the SD / SDXL VAE configs is built"
        down_block_types = down_block_types or ("DownEncoderBlock2D",) * len(block_out_channels)
        up_block_types = up_block_types or ("UpDecoderBlock2D",) * len(block_out_channels)
        assert all(t == "DownEncoderBlock2D" for t in down_block_types) and all(t == "UpDecoderBlock2D" for t in up_block_types)
        assert len(down_block_types) == len(block_out_channels) == len(up_block_types)
        assert all(c % 64 == 0 for c in block_out_channels), "block_out_channels must be multiples of 64 (implicit-GEMM k segments)"
        assert all(c % norm_num_groups == 0 and (c // norm_num_groups) % 4 == 0 for c in block_out_channels), "GroupNorm groups must be >= 4 channels wide"
        self.config = SimpleNamespace(in_channels=in_channels, out_channels=out_channels, down_block_types=tuple(down_block_types),
                                      up_block_types=tuple(up_block_types), block_out_channels=tuple(block_out_channels),
                                      layers_per_block=layers_per_block, act_fn=act_fn, latent_channels=latent_channels,
                                      norm_num_groups=norm_num_groups, sample_size=sample_size, scaling_factor=scaling_factor,
                                      force_upcast=force_upcast)
        eps = 1e-6
        self.encoder = _Encoder(in_channels, latent_channels, block_out_channels, layers_per_block, norm_num_groups, eps)
        self.decoder = _Decoder(latent_channels, out_channels, block_out_channels, layers_per_block, norm_num_groups, eps)
        self.quant_conv = nn.Conv2d(2 * latent_channels, 2 * latent_channels, 1)
        self.post_quant_conv = nn.Conv2d(latent_channels, latent_channels, 1)
        self.requires_grad_(False)
        self._packed, self._packed_key, self._pad_cache = {}, None, {}

    # ------------------------------------------------------------------ checkpoint plumbing (diffusers directory layout)
    @classmethod
    def from_pretrained(cls, path, torch_dtype=None, **kwargs):
        with open(os.path.join(path, cls.config_name)) as f:
            cfg = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
        model = cls(**cfg)
        st = os.path.join(path, "diffusion_pytorch_model.safetensors")
        if os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        else:
            sd = torch.load(os.path.join(path, "diffusion_pytorch_model.bin"), map_location="cpu")
        model.load_state_dict(sd)
        return model.to(torch_dtype) if torch_dtype is not None else model

    def save_pretrained(self, path):
        from safetensors.torch import save_file
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, self.config_name), "w") as f:
            json.dump({"_class_name": "AutoencoderKL", **{k: (list(v) if isinstance(v, tuple) else v) for k, v in vars(self.config).items()}}, f, indent=2)
        save_file({k: v.contiguous() for k, v in self.state_dict().items()}, os.path.join(path, "diffusion_pytorch_model.safetensors"))

    def load_state_dict(self, state_dict, strict=True, **kw):
        sd = {}
        for k, v in state_dict.items():
            parts = k.split(".")
            if "attentions" in parts and parts[-2] in _OLD_ATTN_KEYS:
                k = ".".join(parts[:-2] + [_OLD_ATTN_KEYS[parts[-2]], parts[-1]])
                if v.dim() == 4:
                    v = v[:, :, 0, 0]                          # the deprecated block stored the projections as 1x1 convolutions
            sd[k] = v
        self._packed_key = None
        return super().load_state_dict(sd, strict=strict, **kw)

    @property
    def dtype(self):
        return self.post_quant_conv.weight.dtype

    @property
    def device(self):
        return self.post_quant_conv.weight.device

    # ------------------------------------------------------------------ operand preparation (once per weight version)
    def _prepare(self):
        key = (str(self.device), self.dtype, sum(p._version for p in self.parameters()))
        if key == self._packed_key:
            return
        dev, P = self.device, {}
        for mod in self.modules():
            if isinstance(mod, nn.Conv2d) and mod.kernel_size == (3, 3):
                w = mod.weight.detach().to(dev, F32)
                co, ci = w.shape[:2]
                w2 = torch.zeros(_c8(co), 3, 3, _c8(ci), device=dev)
                w2[:co, :, :, :ci] = w.permute(0, 2, 3, 1)      # [Cout][ky][kx][Cin]: k index = tap * Cin + c (patch-matrix order)
                if mod.stride == (1, 1) and ci % 64 == 0:       # implicit GEMM: tap-interleaved K order [Cin/64][ky][kx][64]
                    w2 = w2.view(_c8(co), 3, 3, ci // 64, 64).permute(0, 3, 1, 2, 4)
                P[id(mod)] = (w2.reshape(_c8(co), -1).to(BF16).contiguous(), self._bias(mod.bias, co, dev), co)
            elif isinstance(mod, (nn.Conv2d, nn.Linear)):
                w = mod.weight.detach().to(dev, F32).flatten(1)
                co, ci = w.shape
                w2 = torch.zeros(_c8(co), _c8(ci), device=dev)
                w2[:co, :ci] = w
                P[id(mod)] = (w2.to(BF16).contiguous(), self._bias(mod.bias, co, dev), co)
            elif isinstance(mod, nn.GroupNorm):
                P[id(mod)] = (mod.weight.detach().to(dev, F32).contiguous(), mod.bias.detach().to(dev, F32).contiguous())
        # Upsample2D convolutions (decoder): conv3x3(nearest 2x upsample(x)) = four 2 x 2 convolutions of x, one per output parity (dy, dx), whose taps are the
        # sums of the 3 x 3 taps that land on the same low-res pixel - 16 of 36 products (ops.gemm(..., up=...)).  Tap sums in fp32, ONE rounding to the operand
        # type; K order [a][b][Cin] (a, b = the 2 x 2 patch rows / columns: low-res pixel (y + dy - 1 + a, x + dx - 1 + b)).
        for blk in self.decoder.up_blocks:
            if hasattr(blk, "upsamplers"):
                cv = blk.upsamplers[0].conv
                w = cv.weight.detach().to(dev, F32)                # (Co, Ci, 3, 3)
                if w.shape[0] % 128 == 0 and w.shape[1] % 64 == 0:
                    phases = []
                    for dy in (0, 1):
                        for dx in (0, 1):
                            wp = torch.zeros(w.shape[0], 2, 2, w.shape[1], device=dev)
                            for ky in range(3):
                                for kx in range(3):
                                    a, b = (dy + ky - 1) // 2 - (dy - 1), (dx + kx - 1) // 2 - (dx - 1)
                                    wp[:, a, b] += w[:, :, ky, kx]
                            phases.append(wp.reshape(w.shape[0], -1).to(BF16).contiguous())
                    P[("up", id(cv))] = (phases, self._bias(cv.bias, w.shape[0], dev))
        co = self.decoder.conv_out                               # few-channel output convolution: direct kernel, weights as [tap][Cout][Cin]
        if co.out_channels <= 4 and co.in_channels % 64 == 0:
            P[("taps", id(co))] = (co.weight.detach().to(dev, F32).permute(2, 3, 0, 1).reshape(9, co.out_channels, co.in_channels).to(BF16).contiguous(),
                                   co.bias.detach().to(dev, F32).contiguous() if co.bias is not None else None)
        for at in (self.encoder.mid_block.attentions[0], self.decoder.mid_block.attentions[0]):   # q, k, v as one GEMM
            ws, bs = zip(*[P[id(m)][:2] for m in (at.to_q, at.to_k, at.to_v)])
            P[("qkv", id(at))] = (torch.cat(ws).contiguous(), torch.cat(bs).contiguous(), sum(w.shape[0] for w in ws))
        self._packed, self._packed_key = P, key

    @staticmethod
    def _bias(b, co, dev):
        out = torch.zeros(_c8(co), device=dev, dtype=F32)
        if b is not None:
            out[:co] = b.detach().to(dev, F32)
        return out

    def clear_cache(self):
        """Drop the cached zero-bordered convolution inputs (one per activation shape)."""
        self._pad_cache.clear()

    # ------------------------------------------------------------------ layers on pixel grids
    def _padded(self, B, H, W, C, dev):
        """Zero-bordered input of an implicit 3x3 convolution: [guard | B x (_img_rows(H, W) >= (H+2) x (W+2)) pixels | guard], guard = W+3
        pixels.  Only the interior is ever written, so the border (and each image's tail) stays zero for the lifetime of the cache entry."""
        key = (B, H, W, C, str(dev))
        buf = self._pad_cache.get(key)
        if buf is None:
            buf = self._pad_cache[key] = torch.zeros((B * _img_rows(H, W) + 2 * (W + 3)) * C, dtype=BF16, device=dev)
        return buf

    def _norm(self, x, gn):
        part = getattr(x, "gn_part", None)
        if part is not None:                                    # the producing convolution's epilogue already summed the output
            mean, rstd = ops.vae_gn_finalize(part, x.B, x.C, gn.num_groups, x.H * x.W, gn.eps)
        else:
            mean, rstd = ops.vae_gn_stats(x, gn.num_groups, gn.eps)
        gamma, beta = self._packed[id(gn)]
        return (mean, rstd, gamma, beta, gn.num_groups)

    def _conv3(self, x, conv, norm=None, silu=False, upsample=1, out_f32=False, residual=None, stats=False):
        """3x3 stride-1 pad-1 convolution of act(norm(x)) (optionally 2x upsampled first); `residual` (a grid) is added to the result -
        inside the GEMM epilogue when it already has the output's padded-grid layout, by the add kernel otherwise.  stats: a GroupNorm
        reads the result next - its per-channel sums are taken in the GEMM epilogue (no separate pass over the output)."""
        w, b, _ = self._packed[id(conv)]
        B, H, W, C, dev = x.B, x.H * upsample, x.W * upsample, x.C, x.buf.device
        assert w.shape[1] == 9 * C
        if C % 64 == 0:                                         # implicit GEMM over the padded pixels
            buf = self._padded(B, H, W, C, dev)
            ip, rp, Co = _img_rows(H, W), W + 2, w.shape[0]
            ops.vae_gn_apply(x, Grid(buf, B, H, W, C, rp, ip, origin=(W + 3) + rp + 1), norm, silu, upsample)
            a = buf.as_strided((B * ip, 9 * C), (C, 1))
            fused = (residual is not None and not out_f32 and residual.C == Co and residual.row_pitch == rp and residual.img_pitch == ip
                     and residual.origin == W + 3 and (residual.B, residual.H, residual.W) == (B, H, W))
            stats = (stats and not out_f32 and (fused or residual is None) and Co % 128 == 0 and B * ip >= 1024 and ip < (1 << 22)
                     and os.environ.get("PXA_VAE_FUSED_GN_STATS", "1") != "0")
            part = torch.zeros(ops.COLSUM_SLOTS, B, Co // 4, 2, dtype=F32, device=dev) if stats else None
            out = ops.gemm(a, w, ops.NT, bias=b, out_dtype=F32 if out_f32 else BF16, k_seg=3 * C, a_seg_stride=rp * C, k_tap=C,
                           act=ops.ACT_ADD_AUX if fused else ops.ACT_NONE, aux=residual.rows() if fused else None,
                           gn_part=part, gn_geom=(ip, rp, H, W) if stats else None)
            if out_f32:
                return out.view(B, ip, -1)[:, :(H + 2) * (W + 2)].view(B, H + 2, W + 2, -1)[:, 1:-1, 1:-1]
            y = Grid(out, B, H, W, Co, rp, ip, origin=W + 3)
            y.gn_part = part
            return y if fused or residual is None else ops.vae_add(y, residual, y)
        assert upsample == 1 and not out_f32 and residual is None
        col = ops.vae_im2col3x3(x, 1, 1, H, W, norm, silu)      # stem convolutions: C = 8 (3 / 4 real channels)
        return Grid(ops.gemm(col, w, ops.NT, bias=b), B, H, W, w.shape[0])

    def _conv3_up2(self, x, conv):
        """Upsample2D: conv3x3(nearest-2x(x)) with GroupNorm statistics of the result, as four low-res 2 x 2 phase convolutions that scatter into the high-res
        padded grid (see _prepare); falls back to the upsampled 3x3 form where the persistent phase kernel does not apply (tiny grids, PXA_VAE_UP_PHASES=0)."""
        up = self._packed.get(("up", id(conv)))
        B, H, W, C, dev = x.B, x.H, x.W, x.C, x.buf.device
        ipL, rpL = _img_rows(H, W), W + 2
        if up is None or B * ipL < 1024 or os.environ.get("PXA_VAE_UP_PHASES", "1") == "0":
            return self._conv3(x, conv, upsample=2, stats=True)
        phases, bias = up
        Co = phases[0].shape[0]
        buf = self._padded(B, H, W, C, dev)                      # low-res zero-bordered copy (no norm / activation in front of an upsampling convolution)
        ops.vae_gn_apply(x, Grid(buf, B, H, W, C, rpL, ipL, origin=(W + 3) + rpL + 1))
        H2, W2 = 2 * H, 2 * W
        ipH, rpH = _img_rows(H2, W2), W2 + 2
        out = torch.empty((B * ipH, Co), dtype=BF16, device=dev)  # only interior rows are written; the others are never read as pixels
        part = torch.zeros(ops.COLSUM_SLOTS, B, Co // 4, 2, dtype=F32, device=dev)
        for i, (dy, dx) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
            off = (W + 3) + (dy - 1) * rpL + dx - 1               # A row m = low-res padded pixel m; its 2 x 2 patch starts at buffer pixel m + off
            a = buf.as_strided((B * ipL, 4 * C), (C, 1), off * C)
            ops.gemm(a, phases[i], ops.NT, bias=bias, out=out, k_seg=2 * C, a_seg_stride=rpL * C, gn_part=part, gn_geom=(ipL, rpL, H, W), up=(rpH, ipH, dy, dx))
        y = Grid(out, B, H2, W2, Co, rpH, ipH, origin=W2 + 3)
        y.gn_part = part
        return y

    def _conv3_s2(self, x, conv):
        """diffusers Downsample2D: F.pad(x, (0, 1, 0, 1)) then Conv2d(3, stride 2, padding 0)."""
        w, b, _ = self._packed[id(conv)]
        Ho, Wo = x.H // 2, x.W // 2
        col = ops.vae_im2col3x3(x, 2, 0, Ho, Wo)
        return Grid(ops.gemm(col, w, ops.NT, bias=b), x.B, Ho, Wo, w.shape[0])

    def _conv1(self, x, lin, key=None, out_f32=False):
        """1x1 convolution / Linear over every pixel slot of the grid."""
        w, b, _ = self._packed[key or id(lin)]
        out = ops.gemm(x.rows(), w, ops.NT, bias=b, out_dtype=F32 if out_f32 else BF16)
        return out if out_f32 else x.like_rows(out, w.shape[0])

    def _resnet(self, x, r, out_stats=True):
        h = self._conv3(x, r.conv1, self._norm(x, r.norm1), silu=True, stats=True)
        sc = x if r.conv_shortcut is None else self._conv1(x, r.conv_shortcut)
        return self._conv3(h, r.conv2, self._norm(h, r.norm2), silu=True, residual=sc, stats=out_stats)

    def _attention(self, x, at):
        B, HW, C, dev = x.B, x.H * x.W, x.C, x.buf.device
        assert HW % 8 == 0, "mid-block attention: H*W must be a multiple of 8"
        t = ops.vae_gn_apply(x, Grid.compact(B, x.H, x.W, C, dev), self._norm(x, at.group_norm))
        qkv = self._conv1(t, None, key=("qkv", id(at))).buf      # (B*HW, 3C)
        o = torch.empty(B * HW, C, dtype=BF16, device=dev)
        # scores of one image at a time (HW x HW fp32).  Round 6: the images alternate over PXA_VAE_ATTN_STREAMS (default 2) HIP streams - the P V product of one
        # image (an NN GEMM of 128 workgroups: half the CUs, 66 us) runs beside the next image's score GEMM and softmax instead of in front of them.  Every
        # buffer is allocated here, on the caller's stream; the side streams start behind the qkv projection and the caller's stream waits for all of them.
        ns = max(1, min(int(os.environ.get("PXA_VAE_ATTN_STREAMS", "2")), B))
        sbuf = [torch.empty(HW, HW, dtype=F32, device=dev) for _ in range(ns)]
        pbuf = [torch.empty(HW, HW, dtype=BF16, device=dev) for _ in range(ns)]
        main = torch.cuda.current_stream(dev)
        side = [main] if ns == 1 else self._side_streams(dev, ns)
        if ns > 1:
            ready = torch.cuda.Event()
            ready.record(main)
            for st in side:
                st.wait_event(ready)
        for i in range(B):
            r, k = slice(i * HW, (i + 1) * HW), i % ns
            with torch.cuda.stream(side[k]):
                ops.gemm(qkv[r, :C], qkv[r, C:2 * C], ops.NT, out_f32=sbuf[k])
                ops.vae_softmax_rows(sbuf[k], C ** -0.5, out=pbuf[k])
                ops.gemm(pbuf[k], qkv[r, 2 * C:], ops.NN, out=o[r])
        if ns > 1:
            for st in side:
                done = torch.cuda.Event()
                done.record(st)
                main.wait_event(done)
        o = self._conv1(Grid(o, B, x.H, x.W, C), at.to_out[0])
        return ops.vae_add(o, x, o)

    def _side_streams(self, dev, n):
        key = (str(dev), n)
        if getattr(self, "_streams", None) is None or self._streams[0] != key:
            self._streams = (key, [torch.cuda.Stream(device=dev) for _ in range(n)])
        return self._streams[1]

    def _mid(self, x, mid):
        return self._resnet(self._attention(self._resnet(x, mid.resnets[0]), mid.attentions[0]), mid.resnets[1])

    def _to_grid(self, t, mul=1.0):
        t = t.detach().to(F32).contiguous()
        B, C, H, W = t.shape
        return ops.vae_nchw_to_grid(t, Grid.compact(B, H, W, _c8(C), t.device), mul)

    # ------------------------------------------------------------------ public API
    @torch.no_grad()
    def encode(self, x, return_dict=True):
        self._prepare()
        enc, n_down = self.encoder, len(self.config.block_out_channels) - 1
        assert x.shape[1] == self.config.in_channels and x.shape[2] % (1 << n_down) == 0 and x.shape[3] % (1 << n_down) == 0
        h = self._conv3(self._to_grid(x), enc.conv_in)
        for blk in enc.down_blocks:
            for i, r in enumerate(blk.resnets):                  # a resampling convolution (no norm) reads the block's last output
                h = self._resnet(h, r, out_stats=not (hasattr(blk, "downsamplers") and i == len(blk.resnets) - 1))
            if hasattr(blk, "downsamplers"):
                h = self._conv3_s2(h, blk.downsamplers[0].conv)
        h = self._mid(h, enc.mid_block)
        h = self._conv3(h, enc.conv_out, self._norm(h, enc.conv_norm_out), silu=True)
        lc2 = 2 * self.config.latent_channels
        m = self._conv1(h, self.quant_conv, out_f32=True)        # fp32 moments over every pixel slot of h's (padded-grid) layout
        assert h.row_pitch == h.W + 2 and h.origin == h.W + 3
        m = m.view(h.B, h.img_pitch, -1)[:, :(h.H + 2) * (h.W + 2)].view(h.B, h.H + 2, h.W + 2, -1)[:, 1:-1, 1:-1, :lc2].permute(0, 3, 1, 2).contiguous().to(x.dtype)
        dist = DiagonalGaussianDistribution(m)
        return SimpleNamespace(latent_dist=dist) if return_dict else (dist,)

    @torch.no_grad()
    def decode(self, z, return_dict=True, generator=None):
        self._prepare()
        dec = self.decoder
        assert z.shape[1] == self.config.latent_channels
        h = self._conv1(self._to_grid(z), self.post_quant_conv)
        h = self._conv3(h, dec.conv_in)
        h = self._mid(h, dec.mid_block)
        for blk in dec.up_blocks:
            for i, r in enumerate(blk.resnets):
                h = self._resnet(h, r, out_stats=not (hasattr(blk, "upsamplers") and i == len(blk.resnets) - 1))
            if hasattr(blk, "upsamplers"):
                h = self._conv3_up2(h, blk.upsamplers[0].conv)
        # (Running the 256 / 512-pixel levels in sub-batches of 2 .. 16 images, so that a layer's tensors stay inside the 256 MB Infinity Cache between the
        # kernels that write and read them, was measured and dropped: 171.8 ms whole batch, 173.3 / 176.4 / 183.7 / 199.2 ms at 16 / 8 / 4 / 2 images -
        # profiles/r6_06_run.txt.)
        taps = self._packed.get(("taps", id(dec.conv_out)))
        if taps is not None and os.environ.get("PXA_VAE_CONV_OUT_GEMM") != "1":      # one pass over h: norm + SiLU + 3x3 conv to the fp32 NCHW image
            img = ops.vae_conv3x3_small_out(h, taps[0], taps[1], self.config.out_channels, self._norm(h, dec.conv_norm_out), silu=True).to(z.dtype)
        else:
            img = self._conv3(h, dec.conv_out, self._norm(h, dec.conv_norm_out), silu=True, out_f32=True)
            img = img[..., : self.config.out_channels].permute(0, 3, 1, 2).contiguous().to(z.dtype)
        return SimpleNamespace(sample=img) if return_dict else (img,)

    def forward(self, sample, sample_posterior=False, return_dict=True, generator=None):
        posterior = self.encode(sample).latent_dist
        z = posterior.sample(generator=generator) if sample_posterior else posterior.mode()
        return self.decode(z, return_dict=return_dict)
