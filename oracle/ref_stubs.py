"""TEST INFRASTRUCTURE ONLY (never imported by the product path).

Stub modules for the four third-party packages the reference imports but this image lacks
(timm 0.6.12, xformers 0.0.19, mmcv 1.7.0, torchvision), so that the *unmodified* reference
python under /root/reference can be imported in the build container to (a) validate
oracle/pixart_oracle.py and (b) generate tests/golden/*.pt  (see oracle/make_golden.py).

Semantics follow the pinned versions (reference requirements.txt:1,3,17):
  * timm.models.vision_transformer.{Mlp, Attention, PatchEmbed}, timm.models.layers.DropPath
  * xformers.ops.memory_efficient_attention (BMHK in, contiguous BMHK out, scale 1/sqrt(K)),
    xformers.ops.fmha.BlockDiagonalMask.from_seqlens  -> exact fp32 softmax attention
  * mmcv.Registry(.register_module/.build), mmcv.utils.logging.logger_initialized,
    mmcv.runner.get_dist_info
/root/reference does not exist on the GPU box: nothing here may be used by -m gpu tests,
smoke() or bench.py.
"""
import math
import sys
import types

import torch
import torch.nn as nn

REFERENCE_ROOT = "/root/reference"


def _module(name):
    m = types.ModuleType(name)
    sys.modules[name] = m
    return m


# ----------------------------------------------------------------------------- timm
class DropPath(nn.Module):
    def __init__(self, drop_prob=0.0, scale_by_keep=True):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1 - self.drop_prob
        shape = (x.shape[0],) + (1,) * (x.ndim - 1)
        mask = x.new_empty(shape).bernoulli_(keep)
        return x * mask / keep


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.drop1 = nn.Dropout(drop)
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop2 = nn.Dropout(drop)

    def forward(self, x):
        return self.drop2(self.fc2(self.drop1(self.act(self.fc1(x)))))


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        assert dim % num_heads == 0
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)

    def forward(self, x):
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv.unbind(0)
        attn = (q @ k.transpose(-2, -1)) * self.scale
        attn = self.attn_drop(attn.softmax(dim=-1))
        x = (attn @ v).transpose(1, 2).reshape(B, N, C)
        return self.proj_drop(self.proj(x))


class PatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, norm_layer=None, flatten=True, bias=True):
        super().__init__()
        img_size = (img_size, img_size) if isinstance(img_size, int) else tuple(img_size)
        patch_size = (patch_size, patch_size) if isinstance(patch_size, int) else tuple(patch_size)
        self.img_size, self.patch_size = img_size, patch_size
        self.grid_size = (img_size[0] // patch_size[0], img_size[1] // patch_size[1])
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.flatten = flatten
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size, bias=bias)
        self.norm = norm_layer(embed_dim) if norm_layer else nn.Identity()

    def forward(self, x):
        x = self.proj(x)
        if self.flatten:
            x = x.flatten(2).transpose(1, 2)
        return self.norm(x)


# ----------------------------------------------------------------------------- xformers
class BlockDiagonalMask:
    def __init__(self, q_seqlen, kv_seqlen):
        self.q_seqlen, self.kv_seqlen = list(q_seqlen), list(kv_seqlen)

    @classmethod
    def from_seqlens(cls, q_seqlen, kv_seqlen=None):
        return cls(q_seqlen, q_seqlen if kv_seqlen is None else kv_seqlen)

    def materialize(self, dtype, device):
        M, N = sum(self.q_seqlen), sum(self.kv_seqlen)
        bias = torch.full((M, N), float("-inf"), dtype=dtype, device=device)
        qs = ks = 0
        for ql, kl in zip(self.q_seqlen, self.kv_seqlen):
            bias[qs:qs + ql, ks:ks + kl] = 0
            qs += ql
            ks += kl
        return bias


def memory_efficient_attention(query, key, value, attn_bias=None, p=0.0, scale=None):
    assert p == 0.0
    B, M, H, K = query.shape
    scale = K ** -0.5 if scale is None else scale
    q = query.permute(0, 2, 1, 3).float()
    k = key.permute(0, 2, 1, 3).float()
    v = value.permute(0, 2, 1, 3).float()
    bias = None
    if attn_bias is not None:
        bias = attn_bias.materialize(torch.float32, q.device) if isinstance(attn_bias, BlockDiagonalMask) else attn_bias.reshape(B, H, M, -1).float()
    # the same arithmetic head group by head group when the (B,H,M,N) score tensor would not fit comfortably in host memory
    # (2K latents: 16 x 16384 x 16384 fp32 = 17 GB per copy); per-head results are independent, so chunking changes nothing
    step = H if B * H * M * k.shape[2] <= (1 << 31) else max(1, (1 << 31) // (B * M * k.shape[2]))
    outs = []
    for h0 in range(0, H, step):
        s = (q[:, h0:h0 + step] @ k[:, h0:h0 + step].transpose(-1, -2)) * scale
        if bias is not None:
            s = s + (bias if bias.dim() == 2 else bias[:, h0:h0 + step])
        outs.append(s.softmax(dim=-1) @ v[:, h0:h0 + step])
    o = outs[0] if len(outs) == 1 else torch.cat(outs, dim=1)
    return o.permute(0, 2, 1, 3).contiguous().to(query.dtype)


# ----------------------------------------------------------------------------- mmcv
class Registry:
    def __init__(self, name):
        self.name = name
        self._module_dict = {}

    def register_module(self, name=None, force=False, module=None):
        def deco(obj):
            self._module_dict[name or obj.__name__] = obj
            return obj
        return deco

    def get(self, key):
        return self._module_dict.get(key)

    def build(self, cfg, default_args=None):
        args = dict(cfg)
        if default_args:
            for k, v in default_args.items():
                args.setdefault(k, v)
        obj = self._module_dict[args.pop("type")]
        return obj(**args)


def install():
    """Insert the stub modules into sys.modules and put the reference on sys.path."""
    sys.dont_write_bytecode = True  # never write __pycache__ into /root/reference
    if "xformers" in sys.modules and getattr(sys.modules["xformers"], "_pxa_stub", False):
        return
    timm = _module("timm")
    tm = _module("timm.models")
    tl = _module("timm.models.layers")
    tv = _module("timm.models.vision_transformer")
    timm.models = tm
    tm.layers, tm.vision_transformer = tl, tv
    tl.DropPath = DropPath
    tv.Mlp, tv.Attention, tv.PatchEmbed = Mlp, Attention, PatchEmbed

    xf = _module("xformers")
    xf._pxa_stub = True
    xo = _module("xformers.ops")
    fm = _module("xformers.ops.fmha")
    xf.ops = xo
    xo.fmha = fm
    xo.memory_efficient_attention = memory_efficient_attention
    fm.BlockDiagonalMask = BlockDiagonalMask

    mm = _module("mmcv")
    mu = _module("mmcv.utils")
    ml = _module("mmcv.utils.logging")
    mr = _module("mmcv.runner")
    mm.Registry = Registry
    mm.utils, mm.runner = mu, mr
    mu.logging = ml
    ml.logger_initialized = {}
    mr.get_dist_info = lambda: (0, 1)

    mm.build_from_cfg = lambda cfg, registry, default_args=None: registry.build(cfg, default_args)

    tvm = _module("torchvision")
    tvt = _module("torchvision.transforms")
    tvm.transforms = tvt
    # data-side imports of the reference (diffusion/data/*): names only, the feature-file path never calls them
    tvd = _module("torchvision.datasets")
    tvf = _module("torchvision.datasets.folder")
    tvm.datasets, tvd.folder = tvd, tvf
    tvf.default_loader = lambda path: (_ for _ in ()).throw(RuntimeError("image loading is outside the stub's scope"))
    tvf.IMG_EXTENSIONS = (".jpg", ".jpeg", ".png", ".webp")
    for n in ("Compose", "Lambda", "Resize", "CenterCrop", "ToTensor", "Normalize", "RandomHorizontalFlip", "RandomCrop"):
        setattr(tvt, n, type(n, (), {"__init__": lambda self, *a, **k: None}))
    tff = _module("torchvision.transforms.functional")
    tvt.functional = tff
    tff.InterpolationMode = type("InterpolationMode", (), {"BICUBIC": "bicubic", "LANCZOS": "lanczos", "BILINEAR": "bilinear"})
    dfs = _module("diffusers")
    dfu = _module("diffusers.utils")
    dft = _module("diffusers.utils.torch_utils")
    dfs.utils, dfu.torch_utils = dfu, dft
    dft.randn_tensor = lambda shape, generator=None, device=None, dtype=None: torch.randn(tuple(shape), generator=generator, device=device, dtype=dtype)
    for n in ("AutoencoderKL", "DPMSolverMultistepScheduler", "Transformer2DModel", "PixArtAlphaPipeline", "PixArtSigmaPipeline",
              "get_cosine_schedule_with_warmup", "get_constant_schedule_with_warmup"):
        setattr(dfs, n, type(n, (), {}))
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def reference_available():
    import os
    return os.path.isdir(REFERENCE_ROOT + "/diffusion/model/nets")
