"""Tensor-level wrappers over the C ABI (pixart_sigma_amd/lib.py).  Each function validates shapes/dtypes, allocates
outputs with torch (device memory + stream plumbing only) and launches exactly one libpixart_hip.so entry point.
No math happens in Python."""
import math

import torch

from . import lib
from .lib import AttnArgs, GemmArgs, GridArg, call, ptr

# BF16 = the process's 16-bit operand dtype: torch.bfloat16, or torch.float16 under PXA_OPERAND_DTYPE=f16 (historical name)
from .lib import OPERAND_DTYPE as BF16  # noqa: E402
F32 = torch.float32
NT, NN, TN = 0, 1, 2
ACT_NONE, ACT_GELU, ACT_GELU_GRAD, ACT_GELU_SAVE_GRAD, ACT_MUL_AUX, ACT_ADD_AUX = 0, 1, 2, 3, 4, 5
_SPLITK_WS = {}


def _chk(t, dtype, name):
    assert t.is_cuda and t.dtype == dtype and t.stride(-1) == 1, f"{name}: need contiguous-last-dim {dtype} CUDA tensor"


def gemm(a, b, layout=NT, bias=None, act=ACT_NONE, aux=None, out=None, out2=None, out_f32=None, accumulate=False,
         split_k=1, out_dtype=BF16, colsum=None, k_seg=0, a_seg_stride=0, k_tap=0, gn_part=None, gn_geom=None, descending=False, up=None):
    """C = op(A) op(B) (see include/pixart_hip.h).  a, b: 2-D bf16 (row stride arbitrary multiple of 8).
    Returns the bf16 output (or fp32 when out_dtype is float32 / out_f32 is given).
    k_seg / a_seg_stride / k_tap: segmented-K A operand (the implicit 3x3 convolution of the VAE kernel set).
    gn_part (PXA_COLSUM_SLOTS, B, N/4, 2) fp32 zeros + gn_geom = (img_rows, row_pitch, H, W): GroupNorm statistics of the output
    accumulated by the epilogue (see pxa_gemm_args.gn_part).
    descending: the persistent NT / NN kernels walk their output tiles from the last token rows to the first (pxa_gemm_args.items_descending) - for a launch
    whose A operand was just written by a kernel that swept the rows upwards; bit-identical results.
    up = (row_pitch, img_rows, dy, dx) of the HIGH-RES padded output grid: this launch is phase (dy, dx) of a 3x3 convolution over a 2x nearest-upsampled
    input, computed on the low-res grid (pxa_gemm_args.up_*); `out` must be given (its rows are high-res padded pixels)."""
    _chk(a, BF16, "A")
    _chk(b, BF16, "B")
    if layout == NT:
        M, K = a.shape
        N, K2 = b.shape
    elif layout == NN:
        M, K = a.shape
        K2, N = b.shape
    else:
        K, M = a.shape
        K2, N = b.shape
    assert K == K2, f"gemm: K mismatch {K} vs {K2}"
    g = GemmArgs()
    g.A, g.B, g.lda, g.ldb = ptr(a), ptr(b), a.stride(0), b.stride(0)
    g.M, g.N, g.K, g.layout = M, N, K, layout
    if bias is not None:
        _chk(bias, F32, "bias")
        assert bias.numel() == N
    g.bias, g.act = ptr(bias), act
    if act in (ACT_GELU_GRAD, ACT_MUL_AUX, ACT_ADD_AUX):
        _chk(aux, BF16, "aux")
        g.aux, g.ldaux = ptr(aux), aux.stride(0)
    want_f32 = out_f32 is not None or out_dtype == F32
    if want_f32:
        if out_f32 is None:
            out_f32 = torch.empty((M, N), dtype=F32, device=a.device)
        _chk(out_f32, F32, "out_f32")
        g.out_f32, g.ld_f32 = ptr(out_f32), out_f32.stride(0)
    else:
        if out is None:
            out = torch.empty((M, N), dtype=BF16, device=a.device)
    if out is not None:
        _chk(out, BF16, "out")
        g.out_bf16, g.ld_out = ptr(out), out.stride(0)
    if out2 is not None:
        _chk(out2, BF16, "out2")
        assert out is not None and out2.stride(0) == out.stride(0)
        g.out2_bf16 = ptr(out2)
    g.accumulate, g.split_k = int(accumulate), split_k
    g.items_descending = int(bool(descending))
    g.k_seg, g.a_seg_stride, g.k_tap = k_seg, a_seg_stride, k_tap
    if gn_part is not None:
        _chk(gn_part, F32, "gn_part")
        assert gn_part.is_contiguous() and gn_part.numel() == COLSUM_SLOTS * (M // gn_geom[0]) * (N // 4) * 2
        g.gn_part = ptr(gn_part)
        g.gn_img_rows, g.gn_row_pitch, g.gn_h, g.gn_w = gn_geom
    if up is not None:
        assert out is not None and gn_part is not None
        g.up_row_pitch, g.up_img_rows, g.up_dy, g.up_dx = up
    if colsum is not None:               # (PXA_COLSUM_SLOTS, stride) partial buffer view: row 0 of the slice to accumulate
        g.colsum, g.colsum_stride = ptr(colsum), colsum.stride(0)
    if accumulate and split_k != 1:       # split-K partial slabs: caller-owned workspace, cached per device (max 16 slabs)
        need = 16 * M * N
        ws = _SPLITK_WS.get(a.device)
        if ws is None or ws.numel() < need:
            ws = _SPLITK_WS[a.device] = torch.empty(need, dtype=F32, device=a.device)
        g.splitk_ws, g.splitk_ws_elems = ptr(ws), ws.numel()
    call("pxa_gemm", g)
    return out_f32 if want_f32 else out


def ln_mod_fwd(x, shift=None, scale=None, mod_stride=0, u=None, gate=None, gate_stride=None, x_out=None, want_xn=True, want_xb=False,
               want_stats=False, rows_per_batch=None, eps=1e-6):
    """x' = x + gate*u; xn = LN(x')*(1+scale)+shift.  x: (R,D) fp32.  shift/scale/gate: fp32 views whose sample b
    starts at data_ptr + b*mod_stride floats.  Returns dict(x, xn, xb, mean, rstd)."""
    _chk(x, F32, "x")
    R, D = x.shape
    rpb = rows_per_batch or R
    xn = torch.empty((R, D), dtype=BF16, device=x.device) if want_xn else None
    xb = torch.empty((R, D), dtype=BF16, device=x.device) if want_xb else None
    mean = torch.empty(R, dtype=F32, device=x.device) if want_stats else None
    rstd = torch.empty(R, dtype=F32, device=x.device) if want_stats else None
    if u is not None and x_out is None:
        x_out = torch.empty_like(x)
    call("pxa_ln_mod_fwd", ptr(x), ptr(u), ptr(gate), mod_stride if gate_stride is None else gate_stride, ptr(shift), ptr(scale), mod_stride, ptr(x_out), ptr(xn), ptr(xb),
         ptr(mean), ptr(rstd), R, D, rpb, eps)
    return {"x": x_out if x_out is not None else x, "xn": xn, "xb": xb, "mean": mean, "rstd": rstd}


def ln_mod_bwd(dy, x, mean, rstd, scale, mod_stride, dx_in, dx_out, dshift, dscale, dmod_stride, rows_per_batch, dx_bf16=None, dbias=None):
    """dbias: optional (COLSUM_SLOTS, >= D) fp32 partials view receiving the column sums of dx_out (the bias gradient of the Linear behind dx_bf16)."""
    R, D = x.shape
    call("pxa_ln_mod_bwd", ptr(dy), ptr(x), ptr(mean), ptr(rstd), ptr(scale), mod_stride, ptr(dx_in), ptr(dx_out), ptr(dx_bf16),
         ptr(dshift), ptr(dscale), dmod_stride, ptr(dbias), dbias.stride(0) if dbias is not None else 0, R, D, rows_per_batch)
    return dx_out


def ln_affine_fwd(x, w, b, eps=1e-5, save=True):
    """nn.LayerNorm(D) (affine) over bf16 rows, IN PLACE on the (possibly column-sliced) 2-D view x; returns (xsave, mean, rstd)."""
    _chk(x, BF16, "x"); _chk(w, F32, "w"); _chk(b, F32, "b")
    R, D = x.shape
    assert x.stride(1) == 1
    xsave = torch.empty((R, D), dtype=BF16, device=x.device) if save else None
    mean, rstd = torch.empty(R, dtype=F32, device=x.device), torch.empty(R, dtype=F32, device=x.device)
    call("pxa_ln_affine_fwd", ptr(x), x.stride(0), ptr(w), ptr(b), ptr(x), x.stride(0), ptr(xsave), ptr(mean), ptr(rstd), R, D, float(eps))
    return xsave, mean, rstd


def ln_affine_bwd(dy, xsave, mean, rstd, w, dw, db):
    """Backward of ln_affine_fwd, IN PLACE on the 2-D view dy (dy <- dx); dw / db (fp32 [D]) are accumulated into."""
    _chk(dy, BF16, "dy"); _chk(xsave, BF16, "xsave")
    R, D = dy.shape
    assert dy.stride(1) == 1 and xsave.is_contiguous()
    call("pxa_ln_affine_bwd", ptr(dy), dy.stride(0), ptr(xsave), ptr(mean), ptr(rstd), ptr(w), ptr(dy), dy.stride(0), ptr(dw), ptr(db), R, D)


def gate_bwd(dx, add=None, u=None, gate=None, mod_stride=0, dx_out=None, du=None, dgate=None, dmod_stride=0, rows_per_batch=None, dbias=None):
    R, D = dx.shape
    call("pxa_gate_bwd", ptr(dx), ptr(add), ptr(u), ptr(gate), mod_stride, ptr(dx_out), ptr(du), ptr(dgate), dmod_stride, ptr(dbias),
         dbias.stride(0) if dbias is not None else 0, R, D, rows_per_batch or R)


COLSUM_SLOTS = 16


def colsum_reduce(part, out):
    """out (n,) += part (COLSUM_SLOTS, >= n).sum(0)"""
    call("pxa_colsum_reduce", ptr(part), part.stride(0), ptr(out), out.numel())


def colsum(dy, out):
    """out[n] += sum_r dy[r][n]  (bf16 in, fp32 accumulate)."""
    R, N = dy.shape
    call("pxa_colsum_bf16", ptr(dy), dy.stride(0), ptr(out), R, N)
    return out


Q_PRESCALE = 72 ** -0.5 * 1.4426950408889634      # scale * log2 e of the denoiser's heads: what a prescaled q carries (pxa_attn_args.q_prescaled)


def _attn_args(q, k, v, o, B, H, Nq, Nk, strides, kv_start=None, kv_len=None, max_kv_len=0, scale=None, head_dim=72, q_prescaled=False):
    a = AttnArgs()
    a.q, a.k, a.v, a.o = ptr(q), ptr(k), ptr(v), ptr(o)
    (a.q_bs, a.q_ts, a.q_hs), (a.k_bs, a.k_ts, a.k_hs), (a.v_bs, a.v_ts, a.v_hs), (a.o_bs, a.o_ts, a.o_hs) = strides
    a.B, a.H, a.Nq, a.Nk, a.head_dim = B, H, Nq, Nk, head_dim
    a.kv_start, a.kv_len, a.max_kv_len = ptr(kv_start), ptr(kv_len), max_kv_len
    a.scale = scale if scale is not None else head_dim ** -0.5
    a.q_prescaled = int(bool(q_prescaled))
    return a


def _check_max_kv_len(kw):
    """The keys-resident kernels size their LDS by max_kv_len and clamp to it: a sample longer than the caller's bound would be truncated silently where
    the streaming kernels honour kv_len.  The lengths are only known here when the caller keeps a host copy (`kv_len_host`)."""
    lens = kw.pop("kv_len_host", None)
    if lens is not None and kw.get("max_kv_len", 0) > 0:
        assert max(lens) <= kw["max_kv_len"], f"max_kv_len {kw['max_kv_len']} < longest sample {max(lens)}"


def attention_fwd(q, k, v, o, lse, B, H, Nq, Nk, strides, **kw):
    """q/k/v/o: bf16 tensors (any view); strides = ((q_bs,q_ts,q_hs),(k..),(v..),(o..)) in elements."""
    _check_max_kv_len(kw)
    a = _attn_args(q, k, v, o, B, H, Nq, Nk, strides, **kw)
    a.lse = ptr(lse)
    call("pxa_attn_fwd", a)
    return o


_bwd_stats = {}


def attn_bwd_stats(B, H, Nq, device):
    """The dK/dV kernel's lse / delta workspace (include/pixart_hip.h: pxa_attn_args.bwd_stats): written by the backward's own pre-pass and consumed
    by its last kernel on ONE stream, so the calls of a (device, stream) pair share the largest buffer asked for so far.  Keyed by the current stream
    (two streams would overwrite each other's rows) and bypassed while a graph is being captured (a buffer first allocated during capture lives in the
    graph's private pool and must not be handed to eager calls): there the caching allocator serves each call."""
    n = lib.load().pxa_attn_bwd_stats_bytes(B, H, Nq)
    if torch.cuda.is_current_stream_capturing():
        return torch.empty(n, dtype=torch.uint8, device=device)
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    buf = _bwd_stats.get(key)
    if buf is None or buf.numel() < n:
        buf = _bwd_stats[key] = torch.empty(n, dtype=torch.uint8, device=device)
    return buf


def attention_bwd(q, k, v, o, d_o, lse, delta, dq, dk, dv, B, H, Nq, Nk, strides, dstrides, colsums=(None, None, None), **kw):
    """colsums: optional fp32 (H*72,) accumulators receiving the column sums of dq / dk / dv (bias gradients)."""
    _check_max_kv_len(kw)
    a = _attn_args(q, k, v, o, B, H, Nq, Nk, strides, **kw)
    stats = attn_bwd_stats(B, H, Nq, q.device) if dk is not None else None     # held until the launch below: under graph capture it is a fresh allocation
    a.bwd_stats = ptr(stats)
    a.dq_colsum, a.dk_colsum, a.dv_colsum = (ptr(t) for t in colsums)
    a.colsum_stride = next((t.stride(0) for t in colsums if t is not None), 0)
    a.lse, a.delta, a.d_o, a.dq, a.dk, a.dv = ptr(lse), ptr(delta), ptr(d_o), ptr(dq), ptr(dk), ptr(dv)
    (a.dq_bs, a.dq_ts, a.dq_hs), (a.dk_bs, a.dk_ts, a.dk_hs), (a.dv_bs, a.dv_ts, a.dv_hs) = dstrides
    call("pxa_attn_bwd", a)            # dq=None skips the dQ kernel, dk=dv=None the dK/dV kernel
    del stats


def patch_embed_fwd(x, w, bias, pos, out=None):
    B, Cc, Hl, Wl = x.shape
    D = w.shape[0]
    if out is None:
        out = torch.empty((B * (Hl // 2) * (Wl // 2), D), dtype=F32, device=x.device)
    call("pxa_patch_embed_fwd", ptr(x), ptr(w), ptr(bias), ptr(pos), ptr(out), B, Cc, Hl, Wl, D)
    return out


def patch_embed_bwd(x, dtok, dw, dbias):
    B, Cc, Hl, Wl = x.shape
    call("pxa_patch_embed_bwd", ptr(x), ptr(dtok), ptr(dw), ptr(dbias), B, Cc, Hl, Wl, dw.shape[0])


def unpatchify_fwd(lin, B, h, w, Co):
    img = torch.empty((B, Co, 2 * h, 2 * w), dtype=F32, device=lin.device)
    call("pxa_unpatchify_fwd", ptr(lin), ptr(img), B, h, w, Co)
    return img


def patchify_bwd(dimg, h, w):
    B, Co = dimg.shape[0], dimg.shape[1]
    dlin = torch.empty((B * h * w, 4 * Co), dtype=BF16, device=dimg.device)
    call("pxa_patchify_bwd", ptr(dimg), ptr(dlin), B, h, w, Co)
    return dlin


def gather_rows_bf16(src, row_idx, L, alt=None, drop=None):
    """src: (B*L, Cw) fp32 rows; row_idx int32 (rows,) -> (rows, Cw) bf16."""
    Cw = src.shape[-1]
    rows = row_idx.numel()
    out = torch.empty((rows, Cw), dtype=BF16, device=src.device)
    call("pxa_gather_rows_bf16", ptr(src), ptr(alt), ptr(row_idx), ptr(drop), ptr(out), rows, L, Cw)
    return out


def kv_compress_fwd(inp, in_bs, in_ts, conv_w, conv_b, ln_w, ln_b, B, H, W, Cc, sr, eps=1e-5):
    out = torch.empty((B, (H // sr) * (W // sr), Cc), dtype=BF16, device=inp.device)
    call("pxa_kv_compress_fwd", ptr(inp), in_bs, in_ts, ptr(conv_w), ptr(conv_b), ptr(ln_w), ptr(ln_b), ptr(out), B, H, W, Cc, sr, eps)
    return out


def kv_compress_bwd(dyc, inp, in_bs, in_ts, conv_w, conv_b, ln_w, din, din_bs, din_ts, d_conv_w, d_conv_b, d_ln_w, d_ln_b, B, H, W, Cc, sr, eps=1e-5):
    call("pxa_kv_compress_bwd", ptr(dyc), ptr(inp), in_bs, in_ts, ptr(conv_w), ptr(conv_b), ptr(ln_w), ptr(din), din_bs, din_ts,
         ptr(d_conv_w), ptr(d_conv_b), ptr(d_ln_w), ptr(d_ln_b), B, H, W, Cc, sr, eps)


def kv_pick(src, dst, full_bs, full_ts, B, H, W, Cc, sr, backward=False):
    call("pxa_kv_pick", int(backward), ptr(src), ptr(dst), full_bs, full_ts, B, H, W, Cc, sr)


def iddpm_loss_fwd(model_out, x0, noise, coef8, tzero):
    B, C, H, W = x0.shape
    mse, vb = torch.empty(B, dtype=F32, device=x0.device), torch.empty(B, dtype=F32, device=x0.device)
    call("pxa_iddpm_loss_fwd", ptr(model_out), ptr(x0), ptr(noise), ptr(coef8), ptr(tzero), B, C, H * W, ptr(mse), ptr(vb))
    return mse, vb


def iddpm_loss_bwd(model_out, x0, noise, coef8, tzero, g_mse, g_vb):
    B, C, H, W = x0.shape
    d = torch.empty_like(model_out)
    call("pxa_iddpm_loss_bwd", ptr(model_out), ptr(x0), ptr(noise), ptr(coef8), ptr(tzero), B, C, H * W, ptr(g_mse), ptr(g_vb), ptr(d))
    return d


def sumsq(x, out):
    call("pxa_sumsq_f32", ptr(x), x.numel(), ptr(out))


def clip_coef(sumsq_t, out2, max_norm, inv_world=1.0):
    call("pxa_clip_coef", ptr(sumsq_t), ptr(out2), float(max_norm), float(inv_world))


def clip_coef_scaled(sumsq_t, out2, max_norm, inv_world, scaler_state, growth_factor, backoff_factor, growth_interval):
    call("pxa_clip_coef_scaled", ptr(sumsq_t), ptr(out2), float(max_norm), float(inv_world), ptr(scaler_state), float(growth_factor),
         float(backoff_factor), int(growth_interval))


def adamw_step_scaled(p, g, m, v, p_bf16, lr, beta1, beta2, eps, weight_decay, gscale, scaler_state):
    call("pxa_adamw_step_scaled", ptr(p), ptr(g), ptr(m), ptr(v), ptr(p_bf16), p.numel(), lr, beta1, beta2, eps, weight_decay, ptr(gscale), ptr(scaler_state))


def adamw_step(p, g, m, v, p_bf16, lr, beta1, beta2, eps, weight_decay, step, gscale=None):
    call("pxa_adamw_step", ptr(p), ptr(g), ptr(m), ptr(v), ptr(p_bf16), p.numel(), lr, beta1, beta2, eps, weight_decay, step, ptr(gscale))


def scale_copy(src, src_stride, nblocks, n_scaled, n_total, scale, out_bf16=None, out_f32=None):
    """nblocks strided blocks of a flat fp32 buffer -> (nblocks, n_total) copies whose first n_scaled elements carry `scale` (one rounding)."""
    out = out_bf16 if out_bf16 is not None else out_f32
    assert out.is_contiguous() and out.shape[0] == nblocks and out[0].numel() == n_total
    call("pxa_scale_copy_f32", ptr(src), src_stride, ptr(out_bf16), ptr(out_f32), n_total, nblocks, n_scaled, n_total, float(scale))


def linear_f32_fwd(x, w, b=None):
    """y = x W^T + b, fp32 (conditioning linears: csrc/condlin.hip)."""
    _chk(x, F32, "x"); _chk(w, F32, "w")
    M, K = x.shape
    N = w.shape[0]
    assert x.is_contiguous() and w.is_contiguous() and w.shape[1] == K
    y = torch.empty((M, N), dtype=F32, device=x.device)
    call("pxa_linear_f32_fwd", ptr(x), ptr(w), ptr(b), ptr(y), M, N, K)
    return y


def linear_f32_bwd(dy, x, w, need_dx=True, need_dw=True, need_db=True):
    _chk(dy, F32, "dy"); _chk(x, F32, "x"); _chk(w, F32, "w")
    M, K = x.shape
    N = w.shape[0]
    assert dy.is_contiguous() and x.is_contiguous() and w.is_contiguous() and dy.shape == (M, N)
    dx = torch.zeros((M, K), dtype=F32, device=x.device) if need_dx else None
    dw = torch.empty((N, K), dtype=F32, device=x.device) if (need_dw or need_db) else None
    db = torch.empty((N,), dtype=F32, device=x.device) if need_db else None
    call("pxa_linear_f32_bwd", ptr(dy), ptr(x), ptr(w), ptr(dx), ptr(dw), ptr(db), M, N, K)
    return dx, (dw if need_dw else None), db


def cast_bf16(x, out=None):
    if out is None:
        out = torch.empty(x.shape, dtype=BF16, device=x.device)
    call("pxa_cast_f32_bf16", ptr(x), ptr(out), x.numel())
    return out


# ------------------------------------------------------------------------------------------------ VAE conv stack
class Grid:
    """A bf16 NHWC pixel grid (pxa_grid): pixel (b, y, x) is the C-vector at buf[(b*img_pitch + y*row_pitch + x + origin) * C]."""

    def __init__(self, buf, B, H, W, C, row_pitch=None, img_pitch=None, origin=0):
        _chk(buf, BF16, "grid buffer")
        self.buf, self.B, self.H, self.W, self.C = buf, B, H, W, C
        self.row_pitch = W if row_pitch is None else row_pitch
        self.img_pitch = H * W if img_pitch is None else img_pitch
        self.origin = origin

    @classmethod
    def compact(cls, B, H, W, C, device):
        return cls(torch.empty(B * H * W, C, dtype=BF16, device=device), B, H, W, C)

    @property
    def is_compact(self):
        return self.origin == 0 and self.row_pitch == self.W and self.img_pitch == self.H * self.W

    def rows(self):
        """(pixels, C) matrix over every pixel slot of the grid (border slots of a padded-grid view included)."""
        return self.buf.view(-1, self.C)[: self.B * self.img_pitch] if not self.is_compact else self.buf.view(-1, self.C)

    def like_rows(self, out, C):
        """The grid of a row-wise op's output: same pixel slots, C channels."""
        return Grid(out, self.B, self.H, self.W, C, self.row_pitch, self.img_pitch, self.origin)

    def arg(self):
        g = GridArg()
        g.ptr, g.B, g.H, g.W, g.C = ptr(self.buf), self.B, self.H, self.W, self.C
        g.row_pitch, g.img_pitch, g.origin = self.row_pitch, self.img_pitch, self.origin
        return g


_GN_WS = {}


def vae_gn_stats(x, groups, eps):
    """GroupNorm statistics of a grid: (mean, rstd), each (B*groups,) fp32."""
    dev = x.buf.device
    ws = _GN_WS.get((dev, x.B * groups))
    if ws is None:
        ws = _GN_WS[(dev, x.B * groups)] = torch.empty(x.B * groups * 2, dtype=torch.float64, device=dev)
    mean = torch.empty(x.B * groups, dtype=F32, device=dev)
    rstd = torch.empty_like(mean)
    call("pxa_vae_gn_stats", x.arg(), groups, eps, ptr(ws), ptr(mean), ptr(rstd))
    return mean, rstd


def vae_gn_finalize(part, B, C, groups, pixels, eps):
    """(mean, rstd) from the per-channel partial sums a convolution epilogue accumulated (gemm(..., gn_part=...))."""
    mean = torch.empty(B * groups, dtype=F32, device=part.device)
    rstd = torch.empty_like(mean)
    call("pxa_vae_gn_finalize", ptr(part), B, C, groups, pixels, eps, ptr(mean), ptr(rstd))
    return mean, rstd


def vae_gn_apply(x, y, norm=None, silu=False, upsample=1):
    """y = act(norm(x)) (optionally 2x nearest upsampled).  norm = (mean, rstd, gamma, beta, groups) or None."""
    mean, rstd, gamma, beta, groups = norm if norm is not None else (None, None, None, None, 1)
    call("pxa_vae_gn_apply", x.arg(), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), groups, int(silu), upsample, y.arg())
    return y


def vae_im2col3x3(x, stride, pad, Ho, Wo, norm=None, silu=False):
    mean, rstd, gamma, beta, groups = norm if norm is not None else (None, None, None, None, 1)
    col = torch.empty(x.B * Ho * Wo, 9 * x.C, dtype=BF16, device=x.buf.device)
    call("pxa_vae_im2col3x3", x.arg(), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), groups, int(silu), stride, pad, Ho, Wo, ptr(col))
    return col


def vae_add(a, b, out):
    call("pxa_vae_add", a.arg(), b.arg(), out.arg())
    return out


def vae_softmax_rows(s, scale, out=None):
    _chk(s, F32, "scores")
    rows, cols = s.shape
    if out is None:
        out = torch.empty(rows, cols, dtype=BF16, device=s.device)
    call("pxa_vae_softmax_rows", ptr(s), s.stride(0), ptr(out), out.stride(0), rows, cols, scale)
    return out


def vae_nchw_to_grid(img, y, mul=1.0):
    _chk(img, F32, "image")
    assert img.is_contiguous() and img.shape[0] == y.B and img.shape[2] == y.H and img.shape[3] == y.W
    call("pxa_vae_nchw_to_grid", ptr(img), img.shape[1], mul, y.arg())
    return y


def vae_conv3x3_small_out(x, w_taps, bias, Cout, norm=None, silu=False):
    """fp32 NCHW image (B, Cout, H, W) = conv3x3(act(norm(x))) for Cout <= 4 (decoder conv_out); w_taps (9, Cout, C) in the operand type."""
    mean, rstd, gamma, beta, groups = norm if norm is not None else (None, None, None, None, 1)
    _chk(w_taps, BF16, "w_taps")
    assert w_taps.is_contiguous() and tuple(w_taps.shape) == (9, Cout, x.C)
    img = torch.empty(x.B, Cout, x.H, x.W, dtype=F32, device=x.buf.device)
    call("pxa_vae_conv3x3_small_out", x.arg(), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), groups, int(silu), ptr(w_taps), ptr(bias), Cout, ptr(img))
    return img


def vae_grid_to_nchw(x, C):
    img = torch.empty(x.B, C, x.H, x.W, dtype=F32, device=x.buf.device)
    call("pxa_vae_grid_to_nchw", x.arg(), C, ptr(img))
    return img
