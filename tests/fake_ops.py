"""TEST INFRASTRUCTURE: a CPU stand-in for pixart_sigma_amd.ops with the same call signatures and output shapes but no arithmetic, so that
engine.Engine.forward / backward - the real kernel SEQUENCING, including the points where gradient buckets are reported complete - can
be driven on CPU processes (tests/test_dp_gloo.py).  Weight-gradient "GEMMs" and bias-gradient passes add a deterministic, rank- and
step-dependent pattern into the flat gradient buffer, so the reduced result is checkable; everything else returns empty tensors."""
import time

import torch

from pixart_sigma_amd import ops as real

BF16, F32 = real.BF16, real.F32
NT, NN, TN = real.NT, real.NN, real.TN
ACT_NONE, ACT_GELU, ACT_GELU_GRAD, ACT_GELU_SAVE_GRAD, ACT_MUL_AUX, ACT_ADD_AUX = (real.ACT_NONE, real.ACT_GELU, real.ACT_GELU_GRAD,
                                                                                   real.ACT_GELU_SAVE_GRAD, real.ACT_MUL_AUX, real.ACT_ADD_AUX)
COLSUM_SLOTS = real.COLSUM_SLOTS
Q_PRESCALE = real.Q_PRESCALE

RANK, STEP, JITTER = 0, 0, 0.0       # set by the test worker
CALLS = []                           # (op, detail) trace


def pattern(t, salt):
    """Deterministic values for a gradient tensor: depends on rank, step, the tensor's size and a per-call-site salt."""
    n = t.numel()
    base = (torch.arange(n, dtype=torch.float32) % 13 - 6.0).view(t.shape)
    return base * (0.25 * (RANK + 1)) * (1.0 + 0.5 * STEP) + float(salt % 5)


def _sleep():
    if JITTER:
        time.sleep(JITTER * ((RANK * 7 + len(CALLS)) % 3))       # ranks finish their buckets at different times


def gemm(a, b, layout=NT, bias=None, act=ACT_NONE, aux=None, out=None, out2=None, out_f32=None, accumulate=False, split_k=1,
         out_dtype=BF16, colsum=None, **kw):
    if layout == NT:
        M, N = a.shape[0], b.shape[0]
    elif layout == NN:
        M, N = a.shape[0], b.shape[1]
    else:
        M, N = a.shape[1], b.shape[1]
    CALLS.append(("gemm", layout, M, N) + (("desc",) if kw.get("descending") else ()))
    if out_f32 is not None:
        assert tuple(out_f32.shape) == (M, N)
        if accumulate:                       # a weight gradient (or the caption-gradient accumulator): lands in the flat buffer
            _sleep()
            out_f32.add_(pattern(out_f32, M + N))
        return out_f32
    if colsum is not None:
        colsum[0].add_(pattern(colsum[0], N))
    if out_dtype == F32:
        return torch.zeros(M, N, dtype=F32)
    return out if out is not None else torch.zeros(M, N, dtype=BF16)


def ln_mod_fwd(x, shift=None, scale=None, mod_stride=0, u=None, gate=None, gate_stride=None, x_out=None, want_xn=True, want_xb=False,
               want_stats=False, rows_per_batch=None, eps=1e-6):
    R, D = x.shape
    z = lambda dt, *s: torch.zeros(*s, dtype=dt)
    return {"x": x_out if x_out is not None else (z(F32, R, D) if u is not None else x), "xn": z(BF16, R, D) if want_xn else None,
            "xb": z(BF16, R, D) if want_xb else None, "mean": z(F32, R) if want_stats else None, "rstd": z(F32, R) if want_stats else None}


def ln_mod_bwd(dy, x, mean, rstd, scale, mod_stride, dx_in, dx_out, dshift, dscale, dmod_stride, rows_per_batch, dx_bf16=None, dbias=None):
    return dx_out


def ln_affine_fwd(x, w, b, eps=1e-5, save=True):
    R, D = x.shape
    return torch.zeros(R, D, dtype=BF16), torch.zeros(R), torch.zeros(R)


def ln_affine_bwd(dy, xsave, mean, rstd, w, dw, db):
    dw.add_(pattern(dw, 3))
    db.add_(pattern(db, 4))


def gate_bwd(dx, add=None, u=None, gate=None, mod_stride=0, dx_out=None, du=None, dgate=None, dmod_stride=0, rows_per_batch=None, dbias=None):
    if dbias is not None:
        dbias[0].add_(pattern(dbias[0], 7))


def colsum_reduce(part, out):
    out.add_(part[:, : out.numel()].sum(0))


def colsum(dy, out):
    out.add_(pattern(out, 11))
    return out


def attention_fwd(q, k, v, o, lse, *a, **kw):
    return o


def attention_bwd(*a, **kw):
    return None


def patch_embed_fwd(x, w, bias, pos, out=None):
    B, Cc, Hl, Wl = x.shape
    return torch.zeros(B * (Hl // 2) * (Wl // 2), w.shape[0], dtype=F32)


def patch_embed_bwd(x, dtok, dw, dbias):
    _sleep()
    dw.add_(pattern(dw, 1))
    dbias.add_(pattern(dbias, 2))


def unpatchify_fwd(lin, B, h, w, Co):
    return torch.zeros(B, Co, 2 * h, 2 * w, dtype=F32)


def patchify_bwd(dimg, h, w):
    return torch.zeros(dimg.shape[0] * h * w, 4 * dimg.shape[1], dtype=BF16)


def gather_rows_bf16(src, row_idx, L, alt=None, drop=None):
    return torch.zeros(row_idx.numel(), src.shape[-1], dtype=BF16)


def kv_compress_fwd(inp, in_bs, in_ts, conv_w, conv_b, ln_w, ln_b, B, H, W, Cc, sr, eps=1e-5):
    return torch.zeros(B, (H // sr) * (W // sr), Cc, dtype=BF16)


def kv_compress_bwd(dyc, inp, in_bs, in_ts, conv_w, conv_b, ln_w, din, din_bs, din_ts, d_conv_w, d_conv_b, d_ln_w, d_ln_b, *a, **kw):
    for i, t in enumerate((d_conv_w, d_conv_b, d_ln_w, d_ln_b)):
        t.add_(pattern(t, 20 + i))


def kv_pick(*a, **kw):
    return None


def scale_copy(src, src_stride, nblocks, n_scaled, n_total, scale, out_bf16=None, out_f32=None):
    CALLS.append(("scale_copy", nblocks, n_total))
    for o in (out_bf16, out_f32):
        if o is not None:
            o.zero_()


def cast_bf16(x, out=None):
    if out is None:
        out = torch.empty(x.shape, dtype=BF16)
    out.copy_(x)
    return out
