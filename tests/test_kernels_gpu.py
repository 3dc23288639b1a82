"""Per-kernel parity of the HIP C-ABI entry points (through pixart_sigma_amd.ops -> ctypes -> libpixart_hip.so)
against plain PyTorch fp32 references of the same op evaluated on the same bf16-rounded inputs.
Tolerances: bf16 outputs carry one bf16 rounding (~1.6e-3 rel-L2 RMS) -> 4e-3; fp32 outputs -> 2e-5 unless stated."""
import math
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from conftest import record_parity, rel_l2  # noqa: E402

F16_BUILD = os.environ.get("PXA_OPERAND_DTYPE", "bf16").lower() in ("f16", "fp16", "float16", "half")
# one rounding of a 16-bit output: bf16 ~1.6e-3 rel-L2 RMS -> 4e-3; IEEE fp16 (3 more mantissa bits) ~2e-4 -> 5e-4
BF16_TOL = 5e-4 if F16_BUILD else 4e-3


def _opd():
    """the 16-bit operand dtype of the loaded library (ops.BF16: _opd(), or torch.float16 under PXA_OPERAND_DTYPE=f16)"""
    from pixart_sigma_amd import ops as o
    return o.BF16


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from pixart_sigma_amd import ops as o
    assert (o.BF16 == torch.float16) == F16_BUILD, "PXA_OPERAND_DTYPE and the loaded library disagree"
    return o


def rnd(*shape, scale=1.0, seed=0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).cuda()


def bf(x):
    return x.to(_opd())


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (300, 1152, 1152), (130, 32, 72), (64, 3456, 256), (1000, 4608, 1152)])
def test_gemm_nt_bias(ops, M, N, K):
    a, w, b = bf(rnd(M, K, seed=1)), bf(rnd(N, K, scale=K ** -0.5, seed=2)), rnd(N, seed=3)
    ref = a.float() @ w.float().t() + b
    out = ops.gemm(a, w, ops.NT, bias=b)
    assert rel_l2(out.float(), ref) < BF16_TOL
    outf = ops.gemm(a, w, ops.NT, bias=b, out_dtype=torch.float32)
    assert rel_l2(outf, ref) < 2e-5


def test_gemm_gelu_dual_output(ops):
    M, N, K = 520, 4608, 1152
    a, w, b = bf(rnd(M, K, seed=1)), bf(rnd(N, K, scale=K ** -0.5, seed=2)), rnd(N, seed=3)
    pre = a.float() @ w.float().t() + b
    out2 = torch.empty(M, N, dtype=_opd(), device="cuda")
    out = ops.gemm(a, w, ops.NT, bias=b, act=ops.ACT_GELU, out2=out2)
    assert rel_l2(out2.float(), pre) < BF16_TOL
    assert rel_l2(out.float(), F.gelu(pre, approximate="tanh")) < BF16_TOL


def test_gemm_nn_gelu_grad(ops):
    M, N, K = 333, 1152, 4608  # dX = dY W ; W stored (K=out_features, N=in_features)
    dy, w = bf(rnd(M, K, seed=1)), bf(rnd(K, N, scale=K ** -0.5, seed=2))
    ref = dy.float() @ w.float()
    out = ops.gemm(dy, w, ops.NN)
    assert rel_l2(out.float(), ref) < BF16_TOL
    pre = bf(rnd(M, N, seed=5))
    x = pre.float().clone().requires_grad_(True)
    F.gelu(x, approximate="tanh").backward(torch.ones_like(x))
    part = torch.zeros(ops.COLSUM_SLOTS, N + 64, device="cuda")    # slotted partials, then folded into the gradient
    out = ops.gemm(dy, w, ops.NN, act=ops.ACT_GELU_GRAD, aux=pre, colsum=part[:, 64:])
    assert rel_l2(out.float(), ref * x.grad) < BF16_TOL
    cs = torch.ones(N, device="cuda")
    ops.colsum_reduce(part[:, 64:], cs)
    assert rel_l2(cs, 1 + (ref * x.grad).sum(0)) < 1e-4          # fused bias-gradient column sums (staged epilogue)
    part.zero_()
    ops.gemm(dy[:, :72], w[:72], ops.NN, colsum=part)                # K=72: register-staged fallback kernel + separate column-sum pass
    assert rel_l2(part.sum(0)[:N], (dy[:, :72].float() @ w[:72].float()).to(_opd()).float().sum(0)) < 1e-4


@pytest.mark.parametrize("M,N,K", [(1100, 1152, 256), (2048, 1280, 128), (1024, 4608, 64), (2300, 1096, 192), (4096, 1152, 96),
                                   (2300, 1096, 640), (1100, 1152, 512)])     # the last two: 20 / 16 k-units - long enough for the aux flavours' L2 touch window (GEMM_AUX_TOUCH)
def test_gemm_persistent_kernel_epilogues(ops, M, N, K):
    """M, N >= 1024: the persistent 256x256 kernel (ragged last tiles in both directions; N % 256 <= 128: half-width remainder items)
    with every epilogue flavour."""
    a, w, b = bf(rnd(M, K, seed=1)), bf(rnd(N, K, scale=K ** -0.5, seed=2)), rnd(N, seed=3)
    pre = a.float() @ w.float().t() + b
    x = pre.clone().requires_grad_(True)
    F.gelu(x, approximate="tanh").backward(torch.ones_like(x))
    assert rel_l2(ops.gemm(a, w, ops.NT, bias=b).float(), pre) < BF16_TOL                      # bias only
    out2 = torch.empty(M, N, dtype=_opd(), device="cuda")
    out = ops.gemm(a, w, ops.NT, bias=b, act=ops.ACT_GELU_SAVE_GRAD, out2=out2)                # training forward of fc1
    assert rel_l2(out.float(), F.gelu(pre, approximate="tanh")) < BF16_TOL
    assert rel_l2(out2.float(), x.grad) < BF16_TOL
    out = ops.gemm(a, w, ops.NT, bias=b, act=ops.ACT_GELU, out2=out2)                          # run-time generic flavour
    assert rel_l2(out.float(), F.gelu(pre, approximate="tanh")) < BF16_TOL and rel_l2(out2.float(), pre) < BF16_TOL
    out = ops.gemm(a, w, ops.NT, bias=b, act=ops.ACT_GELU)                                     # single output (inference forward of fc1)
    assert rel_l2(out.float(), F.gelu(pre, approximate="tanh")) < BF16_TOL
    wt = bf(rnd(K, N, scale=K ** -0.5, seed=4))                                                # NN: dX = dY W
    ref = a.float() @ wt.float()
    aux = bf(rnd(M, N, seed=5))
    part = torch.zeros(ops.COLSUM_SLOTS, N, device="cuda")
    out = ops.gemm(a, wt, ops.NN, act=ops.ACT_MUL_AUX, aux=aux, colsum=part)                   # backward of that path + bias-gradient sums
    assert rel_l2(out.float(), ref * aux.float()) < BF16_TOL
    assert rel_l2(part.sum(0), out.float().sum(0)) < 1e-4       # the column sums are taken over exactly the bf16 values that are stored
    assert rel_l2(part.sum(0), (ref * aux.float()).sum(0)) < 5e-3   # ... so they carry the bf16 rounding of the summands (2^-9 each)
    out = ops.gemm(a, w, ops.NT, bias=b, act=ops.ACT_ADD_AUX, aux=aux)                         # residual add (VAE resnet), run-time flavour
    assert rel_l2(out.float(), pre + aux.float()) < BF16_TOL
    g = bf(pre)
    out = ops.gemm(a, wt, ops.NN, act=ops.ACT_GELU_GRAD, aux=g)
    gx = g.float().clone().requires_grad_(True)
    F.gelu(gx, approximate="tanh").backward(torch.ones_like(gx))
    assert rel_l2(out.float(), ref * gx.grad) < BF16_TOL


@pytest.mark.parametrize("K,split", [(512, 1), (1000, 3), (4096, 8)])
def test_gemm_tn_splitk_accumulate(ops, K, split):
    M, N = 1152, 3456  # dW[M][N] = sum_k A[k][M] B[k][N]
    a, b = bf(rnd(K, M, seed=1)), bf(rnd(K, N, seed=2))
    ref = a.float().t() @ b.float()
    out = torch.zeros(M, N, device="cuda")
    ops.gemm(a, b, ops.TN, out_f32=out, accumulate=True, split_k=split)
    assert rel_l2(out, ref) < 2e-5
    ops.gemm(a, b, ops.TN, out_f32=out, accumulate=True, split_k=split)  # accumulates
    assert rel_l2(out, 2 * ref) < 2e-5


def test_gemm_strided_views(ops):
    """operands that are column slices of wider tensors (k/v halves of kv, q/k/v thirds of dqkv)."""
    M, K = 200, 1152
    big = bf(rnd(M, 3 * K, seed=1))
    w = bf(rnd(1152, K, scale=K ** -0.5, seed=2))
    a = big[:, K:2 * K]
    assert rel_l2(ops.gemm(a, w, ops.NT).float(), a.float() @ w.float().t()) < BF16_TOL


# ------------------------------------------------------------------------------------------------ adaLN rows
def _mod(B, D, seed):
    return rnd(B, 6, D, scale=0.3, seed=seed)


def test_ln_mod_fwd_variants(ops):
    B, N, D = 3, 50, 1152
    R = B * N
    x, u, mod = rnd(R, D, seed=1), bf(rnd(R, D, seed=2)), _mod(B, D, 3)
    shift, scale, gate = mod[:, 0], mod[:, 1], mod[:, 2]
    rep = lambda t: t.repeat_interleave(N, 0)
    # plain LN + modulate
    r = ops.ln_mod_fwd(x, shift, scale, 6 * D, rows_per_batch=N, want_stats=True)
    ref = F.layer_norm(x, (D,), eps=1e-6) * (1 + rep(scale)) + rep(shift)
    assert rel_l2(r["xn"].float(), ref) < BF16_TOL
    assert rel_l2(r["mean"], x.mean(1)) < 1e-4 and rel_l2(r["rstd"], (x.var(1, unbiased=False) + 1e-6).rsqrt()) < 1e-5
    # gated residual + LN, bf16 copy
    r = ops.ln_mod_fwd(x, shift, scale, 6 * D, u=u, gate=gate, rows_per_batch=N, want_xb=True)
    xr = x + rep(gate) * u.float()
    assert rel_l2(r["x"], xr) < 1e-6
    assert rel_l2(r["xb"].float(), xr) < BF16_TOL
    assert rel_l2(r["xn"].float(), F.layer_norm(xr, (D,), eps=1e-6) * (1 + rep(scale)) + rep(shift)) < BF16_TOL
    # ungated residual, no LN, in place
    x2 = x.clone()
    r = ops.ln_mod_fwd(x2, u=u, x_out=x2, want_xn=False, want_xb=True, rows_per_batch=N)
    assert rel_l2(x2, x + u.float()) < 1e-6 and r["xn"] is None


@pytest.mark.parametrize("B,N", [(2, 40), (2, 256), (3, 128)])   # 40: 16-row chunks straddle samples (direct atomics); others: LDS block combine
def test_ln_mod_bwd(ops, B, N):
    D = 1152
    R = B * N
    x, mod, dy, dxin = rnd(R, D, seed=1), _mod(B, D, 2), bf(rnd(R, D, seed=3)), rnd(R, D, seed=4)
    shift, scale = mod[:, 0], mod[:, 1]
    st = ops.ln_mod_fwd(x, shift, scale, 6 * D, rows_per_batch=N, want_stats=True)
    xr = x.clone().requires_grad_(True)
    sh, sc = shift.clone().requires_grad_(True), scale.clone().requires_grad_(True)
    y = F.layer_norm(xr, (D,), eps=1e-6) * (1 + sc.repeat_interleave(N, 0)) + sh.repeat_interleave(N, 0)
    y.backward(dy.float())
    dmod = torch.zeros(B, 6, D, device="cuda")
    dx = torch.empty_like(x)
    ops.ln_mod_bwd(dy, x, st["mean"], st["rstd"], scale, 6 * D, dxin, dx, dmod[:, 0], dmod[:, 1], 6 * D, N)
    assert rel_l2(dx, xr.grad + dxin) < 1e-5
    assert rel_l2(dmod[:, 0], sh.grad) < 1e-5 and rel_l2(dmod[:, 1], sc.grad) < 1e-5
    assert dmod[:, 2:].abs().max() == 0
    # round 5: the bias gradient of the Linear behind dx (column sums of dx) into slotted partials, with the 16-bit copy of dx
    part = torch.zeros(ops.COLSUM_SLOTS, D + 128, device="cuda")
    dmod2, dx2 = torch.zeros(B, 6, D, device="cuda"), torch.empty_like(x)
    dxb = torch.empty(R, D, dtype=_opd(), device="cuda")
    ops.ln_mod_bwd(dy, x, st["mean"], st["rstd"], scale, 6 * D, dxin, dx2, dmod2[:, 0], dmod2[:, 1], 6 * D, N, dx_bf16=dxb, dbias=part[:, 64:64 + D])
    assert torch.equal(dx2, dx) and rel_l2(dmod2, dmod) < 1e-6 and rel_l2(dxb.float(), dx) < BF16_TOL
    assert rel_l2(part.sum(0)[64:64 + D], dx.sum(0)) < 1e-5 and part[:, :64].abs().max() == 0 and part[:, 64 + D:].abs().max() == 0


def test_ln_affine_fwd_bwd(ops):
    """q / k LayerNorm of qk_norm=True: affine nn.LayerNorm(1152, eps 1e-5) in place on a column block of a wider bf16 buffer."""
    R, D = 300, 1152
    buf = bf(rnd(R, 3 * D, seed=1))
    x = buf[:, D:2 * D].float().clone()
    w, b = 1 + 0.1 * rnd(D, seed=2), 0.1 * rnd(D, seed=3)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y = F.layer_norm(xr, (D,), wr, br, eps=1e-5)
    dy = bf(rnd(R, D, seed=4))
    y.backward(dy.float())
    keep = buf.clone()
    xs, mean, rstd = ops.ln_affine_fwd(buf[:, D:2 * D], w, b)
    assert rel_l2(buf[:, D:2 * D].float(), y.detach()) < BF16_TOL and torch.equal(xs, keep[:, D:2 * D])
    assert torch.equal(buf[:, :D], keep[:, :D]) and torch.equal(buf[:, 2 * D:], keep[:, 2 * D:])     # neighbours untouched
    dbuf = torch.zeros(R, 3 * D, dtype=_opd(), device="cuda")
    dbuf[:, D:2 * D] = dy
    dw, db = torch.ones(D, device="cuda"), torch.ones(D, device="cuda")
    ops.ln_affine_bwd(dbuf[:, D:2 * D], xs, mean, rstd, w, dw, db)
    assert rel_l2(dbuf[:, D:2 * D].float(), xr.grad) < BF16_TOL
    assert rel_l2(dw - 1, wr.grad) < 1e-4 and rel_l2(db - 1, br.grad) < 1e-4


@pytest.mark.parametrize("B,N", [(2, 40), (2, 256), (3, 128)])
def test_gate_bwd_and_colsum(ops, B, N):
    D = 1152
    R = B * N
    dx, add, u, mod = rnd(R, D, seed=1), bf(rnd(R, D, seed=2)), bf(rnd(R, D, seed=3)), _mod(B, D, 4)
    gate = mod[:, 2]
    g = dx + add.float()
    dgate = torch.zeros(B, 6, D, device="cuda")
    dxo, du = torch.empty_like(dx), torch.empty(R, D, dtype=_opd(), device="cuda")
    dbias = torch.zeros(ops.COLSUM_SLOTS, D, device="cuda")
    ops.gate_bwd(dx, add=add, u=u, gate=gate, mod_stride=6 * D, dx_out=dxo, du=du, dgate=dgate[:, 2], dmod_stride=6 * D, rows_per_batch=N, dbias=dbias)
    assert rel_l2(dxo, g) < 1e-6
    assert rel_l2(dbias.sum(0), (g * gate.repeat_interleave(N, 0)).sum(0)) < 1e-5
    assert rel_l2(du.float(), g * gate.repeat_interleave(N, 0)) < BF16_TOL
    assert rel_l2(dgate[:, 2], (g * u.float()).view(B, N, D).sum(1)) < 1e-5
    du2 = torch.empty(R, D, dtype=_opd(), device="cuda")
    ops.gate_bwd(dx, du=du2, rows_per_batch=N)  # plain cast
    assert rel_l2(du2.float(), dx) < BF16_TOL
    dy = bf(rnd(777, 3456, seed=5))
    out = torch.ones(3456, device="cuda")
    ops.colsum(dy, out)
    assert rel_l2(out, 1 + dy.float().sum(0)) < 1e-5


# ------------------------------------------------------------------------------------------------ attention
def _attn_ref(q, k, v):
    """q (B,Nq,H,72), k/v (B,Nk,H,72) fp32 -> o, with autograd."""
    s = torch.einsum("bqhd,bkhd->bhqk", q, k) * 72 ** -0.5
    return torch.einsum("bhqk,bkhd->bqhd", s.softmax(-1), v)


@pytest.mark.parametrize("B,H,Nq,Nk", [(2, 16, 256, 256), (1, 4, 200, 200), (2, 3, 130, 77), (1, 2, 64, 1024), (1, 2, 300, 77), (2, 2, 520, 200)])
def test_attention_fwd_bwd_dense(ops, B, H, Nq, Nk):
    C = H * 72
    q, k, v = (bf(rnd(B, n, C, seed=s)) for n, s in ((Nq, 1), (Nk, 2), (Nk, 3)))
    do = bf(rnd(B, Nq, C, seed=4))
    o = torch.empty(B, Nq, C, dtype=_opd(), device="cuda")
    lse = torch.empty(B, H, Nq, device="cuda")
    st = ((Nq * C, C, 72), (Nk * C, C, 72), (Nk * C, C, 72), (Nq * C, C, 72))
    ops.attention_fwd(q, k, v, o, lse, B, H, Nq, Nk, st)
    qr, kr, vr = (t.float().view(B, -1, H, 72).requires_grad_(True) for t in (q, k, v))
    oref = _attn_ref(qr, kr, vr)
    assert rel_l2(o.float().view(B, Nq, H, 72), oref) < BF16_TOL
    sref = torch.einsum("bqhd,bkhd->bhqk", qr, kr) * 72 ** -0.5
    assert rel_l2(lse, torch.logsumexp(sref, -1) / math.log(2)) < 1e-4
    oref.backward(do.float().view(B, Nq, H, 72))
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    delta = torch.empty(B, H, Nq, device="cuda")
    part = torch.zeros(ops.COLSUM_SLOTS, 3 * C, device="cuda")
    ops.attention_bwd(q, k, v, o, do, lse, delta, dq, dk, dv, B, H, Nq, Nk, st, (st[0], st[1], st[2]),
                      colsums=(part[:, :C], part[:, C:2 * C], part[:, 2 * C:]))
    for i, ref in enumerate((qr.grad, kr.grad, vr.grad)):           # fused bias-gradient column sums (dK sums are ~0 by the softmax Jacobian)
        got, want = part.sum(0)[i * C:(i + 1) * C], ref.sum((0, 1)).reshape(-1)
        assert (got - want).norm() < 5e-3 * ref.norm(), i
    assert rel_l2(dv.float().view_as(vr), vr.grad) < 2 * BF16_TOL
    assert rel_l2(dq.float().view_as(qr), qr.grad) < 2 * BF16_TOL
    assert rel_l2(dk.float().view_as(kr), kr.grad) < 2 * BF16_TOL


def test_attention_qkv_packed_layout(ops):
    """q/k/v read from, and dq/dk/dv written into, the (B,N,3,H,72) layout of the qkv GEMM (PixArt_blocks.py:130-131)."""
    B, H, N = 2, 16, 192
    C = H * 72
    qkv = bf(rnd(B, N, 3 * C, seed=1))
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    o = torch.empty(B, N, C, dtype=_opd(), device="cuda")
    lse = torch.empty(B, H, N, device="cuda")
    s3 = (N * 3 * C, 3 * C, 72)
    st = (s3, s3, s3, (N * C, C, 72))
    ops.attention_fwd(q, k, v, o, lse, B, H, N, N, st)
    qr, kr, vr = (t.float().reshape(B, N, H, 72).requires_grad_(True) for t in (q, k, v))
    oref = _attn_ref(qr, kr, vr)
    assert rel_l2(o.float().view(B, N, H, 72), oref) < BF16_TOL
    do = bf(rnd(B, N, C, seed=2))
    oref.backward(do.float().view(B, N, H, 72))
    dqkv = torch.zeros_like(qkv)
    delta = torch.empty(B, H, N, device="cuda")
    ops.attention_bwd(q, k, v, o, do, lse, delta, dqkv[..., :C], dqkv[..., C:2 * C], dqkv[..., 2 * C:], B, H, N, N, st, (s3, s3, s3))
    ref = torch.cat([qr.grad.reshape(B, N, C), kr.grad.reshape(B, N, C), vr.grad.reshape(B, N, C)], -1)
    assert rel_l2(dqkv.float(), ref) < 2 * BF16_TOL


def test_attention_varlen_cross(ops):
    """BlockDiagonalMask.from_seqlens([N]*B, y_lens): sample b attends only to its own packed text rows."""
    B, H, N, lens = 3, 16, 160, [300, 7, 64]
    C = H * 72
    tot = sum(lens)
    q = bf(rnd(B, N, C, seed=1))
    kv = bf(rnd(tot, 2 * C, seed=2))
    do = bf(rnd(B, N, C, seed=3))
    starts = [0, lens[0], lens[0] + lens[1]]
    kv_start = torch.tensor(starts, dtype=torch.int32, device="cuda")
    kv_len = torch.tensor(lens, dtype=torch.int32, device="cuda")
    o = torch.empty(B, N, C, dtype=_opd(), device="cuda")
    lse = torch.empty(B, H, N, device="cuda")
    st = ((N * C, C, 72), (0, 2 * C, 72), (0, 2 * C, 72), (N * C, C, 72))
    ops.attention_fwd(q, kv[:, :C], kv[:, C:], o, lse, B, H, N, max(lens), st, kv_start=kv_start, kv_len=kv_len, max_kv_len=max(lens))
    qr = q.float().view(B, N, H, 72).requires_grad_(True)
    kvr = kv.float().view(tot, 2, H, 72).requires_grad_(True)
    outs = [_attn_ref(qr[b:b + 1], kvr[s:s + n, 0][None], kvr[s:s + n, 1][None]) for b, (s, n) in enumerate(zip(starts, lens))]
    oref = torch.cat(outs, 0)
    assert rel_l2(o.float().view(B, N, H, 72), oref) < BF16_TOL
    oref.backward(do.float().view(B, N, H, 72))
    dq, dkv = torch.empty_like(q), torch.zeros_like(kv)
    delta = torch.empty(B, H, N, device="cuda")
    ops.attention_bwd(q, kv[:, :C], kv[:, C:], o, do, lse, delta, dq, dkv[:, :C], dkv[:, C:], B, H, N, max(lens), st,
                      ((N * C, C, 72), (0, 2 * C, 72), (0, 2 * C, 72)), kv_start=kv_start, kv_len=kv_len, max_kv_len=max(lens))
    assert rel_l2(dq.float().view_as(qr), qr.grad) < 2 * BF16_TOL
    assert rel_l2(dkv.float().view(tot, 2, H, 72), kvr.grad) < 2 * BF16_TOL


@pytest.mark.parametrize("N,lens", [(4133, [300, 7, 64, 129]), (1024, [320, 1, 65]), (600, [20, 300])])
def test_attention_keys_resident(ops, N, lens):
    """Cross-attention with every key of a sample resident in LDS (attn_fwd_kvres_kernel, attn_bwd_dq_kvres_kernel: max_kv_len <= 320 and N_q >= 512): ragged text
    lengths incl. 1 key, exact tile multiples and the 320-key maximum; query counts that are no multiple of the 64-query trip, the 512-query round or
    the 4,096-query workgroup; O and the log-sum-exp the backward reads, against per-sample fp32 attention.  A spike key in the last tile forces the
    online-softmax rescale inside the resident loop."""
    B, H = len(lens), 16
    C = H * 72
    tot = sum(lens)
    q = bf(rnd(B, N, C, seed=1))
    kv = bf(rnd(tot, 2 * C, seed=2))
    starts = [sum(lens[:i]) for i in range(B)]
    kv[starts[0] + lens[0] - 1, :72] = q[0, 5, :72] * 6            # head 0 of sample 0: its last key dominates query 5
    kv_start = torch.tensor(starts, dtype=torch.int32, device="cuda")
    kv_len = torch.tensor(lens, dtype=torch.int32, device="cuda")
    o = torch.full((B, N, C), float("nan"), dtype=ops.BF16, device="cuda")
    lse = torch.full((B, H, N), float("nan"), device="cuda")
    st = ((N * C, C, 72), (0, 2 * C, 72), (0, 2 * C, 72), (N * C, C, 72))
    ops.attention_fwd(q, kv[:, :C], kv[:, C:], o, lse, B, H, N, max(lens), st, kv_start=kv_start, kv_len=kv_len, max_kv_len=max(lens))
    qr = q.float().view(B, N, H, 72)
    kvr = kv.float().view(tot, 2, H, 72)
    for b, (s0, n) in enumerate(zip(starts, lens)):
        oref = _attn_ref(qr[b:b + 1], kvr[s0:s0 + n, 0][None], kvr[s0:s0 + n, 1][None])
        assert rel_l2(o[b].float().view(1, N, H, 72), oref) < BF16_TOL, (b, n)
        sc = torch.einsum("nhd,khd->hnk", qr[b], kvr[s0:s0 + n, 0]) * 72 ** -0.5
        lref = torch.logsumexp(sc, dim=-1) * 1.4426950408889634       # the kernels keep log2-domain statistics
        assert (lse[b] - lref).abs().max() < 2e-2, (b, n)
    assert torch.isfinite(o.float()).all() and torch.isfinite(lse).all()
    # backward on the same shapes: dQ by attn_bwd_dq_kvres_kernel, which also stands in for the delta pre-pass (delta array + the statistics rows the
    # dK/dV kernel reads), then the streaming dK/dV kernel
    do = bf(rnd(B, N, C, seed=3))
    qg = q.float().view(B, N, H, 72).requires_grad_(True)
    kvg = kv.float().view(tot, 2, H, 72).requires_grad_(True)
    oref = torch.cat([_attn_ref(qg[b:b + 1], kvg[s0:s0 + n, 0][None], kvg[s0:s0 + n, 1][None]) for b, (s0, n) in enumerate(zip(starts, lens))], 0)
    oref.backward(do.float().view(B, N, H, 72))
    dq, dkv = torch.full_like(q, float("nan")), torch.zeros_like(kv)
    delta = torch.full((B, H, N), float("nan"), device="cuda")
    ops.attention_bwd(q, kv[:, :C], kv[:, C:], o, do, lse, delta, dq, dkv[:, :C], dkv[:, C:], B, H, N, max(lens), st,
                      ((N * C, C, 72), (0, 2 * C, 72), (0, 2 * C, 72)), kv_start=kv_start, kv_len=kv_len, max_kv_len=max(lens))
    dref = (do.float() * o.float()).view(B, N, H, 72).sum(-1).permute(0, 2, 1)
    assert rel_l2(delta, dref) < 1e-5
    assert rel_l2(dq.float().view_as(qg), qg.grad) < 2 * BF16_TOL
    assert rel_l2(dkv.float().view(tot, 2, H, 72), kvg.grad) < 2 * BF16_TOL


def test_attention_online_softmax_rescale(ops):
    """A key far above the running max arriving in a late tile forces the rescale branch (guide rule 26)."""
    B, H, N = 1, 1, 256
    q, k, v = bf(rnd(B, N, 72, seed=1)), bf(rnd(B, N, 72, seed=2)), bf(rnd(B, N, 72, seed=3))
    k[0, 200] = q[0, 5] * 6  # spike for query 5 in the 4th kv tile
    o = torch.empty_like(q)
    lse = torch.empty(B, H, N, device="cuda")
    st = ((N * 72, 72, 72),) * 4
    ops.attention_fwd(q, k, v, o, lse, B, H, N, N, st)
    oref = _attn_ref(q.float().view(B, N, 1, 72), k.float().view(B, N, 1, 72), v.float().view(B, N, 1, 72))
    assert rel_l2(o.float().view(B, N, 1, 72), oref) < BF16_TOL
    assert (o.float().view(B, N, 72)[0, 5] - oref[0, 5, 0]).abs().max() < 0.03


# ------------------------------------------------------------------------------------------------ token boundary
def test_patch_embed_fwd_bwd(ops):
    B, Hl, Wl, D = 2, 16, 24, 1152
    x, w, b = rnd(B, 4, Hl, Wl, seed=1), rnd(D, 4, 2, 2, scale=0.2, seed=2), rnd(D, seed=3)
    N = (Hl // 2) * (Wl // 2)
    pos = rnd(N, D, seed=4)
    out = ops.patch_embed_fwd(x, w, b, pos)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.conv2d(x, wr, br, stride=2).flatten(2).transpose(1, 2) + pos
    assert rel_l2(out.view(B, N, D), ref) < 1e-6
    dtok = rnd(B * N, D, seed=5)
    ref.backward(dtok.view(B, N, D))
    dw, db = torch.zeros_like(w), torch.zeros_like(b)
    ops.patch_embed_bwd(x, dtok, dw, db)
    assert rel_l2(dw, wr.grad) < 1e-5 and rel_l2(db, br.grad) < 1e-5


def test_unpatchify_and_inverse(ops):
    B, h, w, Co = 2, 8, 12, 8
    lin = rnd(B * h * w, 4 * Co, seed=1)
    img = ops.unpatchify_fwd(lin, B, h, w, Co)
    ref = torch.einsum("nhwpqc->nchpwq", lin.view(B, h, w, 2, 2, Co)).reshape(B, Co, 2 * h, 2 * w)
    assert torch.equal(img, ref)
    back = ops.patchify_bwd(img, h, w)
    assert torch.equal(back, lin.to(_opd()))


def test_gather_rows(ops):
    B, L, Cw = 3, 20, 4096
    y, alt = rnd(B * L, Cw, seed=1), rnd(L, Cw, seed=2)
    mask = torch.zeros(B, L, dtype=torch.bool)
    mask[0, :20] = mask[1, :3] = mask[2, 5:9] = True
    idx = mask.flatten().nonzero().flatten().to(torch.int32).cuda()
    drop = torch.tensor([0, 1, 0], dtype=torch.int32, device="cuda")
    out = ops.gather_rows_bf16(y, idx, L, alt=alt, drop=drop)
    ref = torch.where(drop.bool()[:, None, None], alt[None], y.view(B, L, Cw)).reshape(B * L, Cw)[idx.long()]
    assert torch.equal(out, ref.to(_opd()))


def test_kv_compress_fwd(ops):
    B, H, W, C, sr = 2, 8, 12, 1152, 2
    qkv = bf(rnd(B, H * W, 3 * C, seed=1))
    cw, cb = 0.25 + rnd(C, 1, sr, sr, scale=0.05, seed=2), rnd(C, scale=0.05, seed=3)
    lw, lb = 1 + rnd(C, scale=0.05, seed=4), rnd(C, scale=0.05, seed=5)
    k = qkv[..., C:2 * C]
    out = ops.kv_compress_fwd(k, H * W * 3 * C, 3 * C, cw, cb, lw, lb, B, H, W, C, sr)
    t = k.float().reshape(B, H, W, C).permute(0, 3, 1, 2)
    t = F.conv2d(t, cw, cb, stride=sr, groups=C).reshape(B, C, -1).permute(0, 2, 1)
    ref = F.layer_norm(t, (C,), lw, lb, eps=1e-5)
    assert rel_l2(out.float(), ref) < BF16_TOL


def test_kv_compress_bwd_and_pick(ops):
    B, H, W, C, sr = 2, 8, 12, 1152, 2
    N, Nk = H * W, (H // sr) * (W // sr)
    qkv = bf(rnd(B * N, 3 * C, seed=1))
    cw, cb = 0.25 + rnd(C, 1, sr, sr, scale=0.05, seed=2), rnd(C, scale=0.05, seed=3)
    lw, lb = 1 + rnd(C, scale=0.05, seed=4), rnd(C, scale=0.05, seed=5)
    dyc = bf(rnd(B, Nk, C, seed=6))
    k = qkv[:, C:2 * C]
    kr = k.float().clone().requires_grad_(True)
    cwr, cbr, lwr, lbr = (t.clone().requires_grad_(True) for t in (cw, cb, lw, lb))
    t = F.conv2d(kr.reshape(B, H, W, C).permute(0, 3, 1, 2), cwr, cbr, stride=sr, groups=C).reshape(B, C, -1).permute(0, 2, 1)
    F.layer_norm(t, (C,), lwr, lbr, eps=1e-5).backward(dyc.float())
    dqkv = torch.zeros_like(qkv)
    gcw, gcb, glw, glb = (torch.zeros_like(t) for t in (cw, cb, lw, lb))
    ops.kv_compress_bwd(dyc, k, N * 3 * C, 3 * C, cw, cb, lw, dqkv[:, C:2 * C], N * 3 * C, 3 * C, gcw, gcb, glw, glb, B, H, W, C, sr)
    assert rel_l2(dqkv[:, C:2 * C].float(), kr.grad) < BF16_TOL
    for got, ref, nm in ((gcw, cwr.grad, "conv_w"), (gcb, cbr.grad, "conv_b"), (glw, lwr.grad, "ln_w"), (glb, lbr.grad, "ln_b")):
        assert rel_l2(got, ref) < 1e-4, nm
    assert dqkv[:, :C].abs().max() == 0 and dqkv[:, 2 * C:].abs().max() == 0
    # token pick ('uniform' / 'ave') forward + scatter backward
    kc = torch.empty(B, Nk, C, dtype=_opd(), device="cuda")
    ops.kv_pick(k, kc, N * 3 * C, 3 * C, B, H, W, C, sr)
    ref = k.reshape(B, H, W, C)[:, ::sr, ::sr].reshape(B, Nk, C)
    assert torch.equal(kc, ref)
    back = torch.zeros_like(qkv)
    ops.kv_pick(kc, back[:, C:2 * C], N * 3 * C, 3 * C, B, H, W, C, sr, backward=True)
    refb = torch.zeros(B, H, W, C, dtype=_opd(), device="cuda")
    refb[:, ::sr, ::sr] = ref.reshape(B, H // sr, W // sr, C)
    assert torch.equal(back[:, C:2 * C].reshape(B, H, W, C), refb)


# ------------------------------------------------------------------------------------------------ optimizer
def test_adamw_matches_torch(ops):
    n = 1 << 16
    p0, g1, g2 = rnd(n, seed=1), rnd(n, scale=0.1, seed=2), rnd(n, scale=0.1, seed=3)
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([pr], lr=2e-5, weight_decay=3e-2, eps=1e-10)
    p, m, v = p0.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    pb = torch.empty(n, dtype=_opd(), device="cuda")
    for step, g in enumerate((g1, g2), 1):
        pr.grad = g.clone()
        opt.step()
        ops.adamw_step(p, g, m, v, pb, 2e-5, 0.9, 0.999, 1e-10, 3e-2, step)
    assert rel_l2(p, pr.detach()) < 1e-6
    assert torch.equal(pb, p.to(_opd()))


def test_grad_norm_and_clip(ops):
    g = rnd(100003, seed=1)
    s = torch.zeros(1, device="cuda")
    ops.sumsq(g[:100000], s)
    assert abs(s.item() - g[:100000].double().pow(2).sum().item()) / s.item() < 1e-5
    out = torch.empty(2, device="cuda")
    ops.clip_coef(s, out, 0.01, 0.5)
    norm = math.sqrt(s.item()) * 0.5
    assert abs(out[1].item() - norm) / norm < 1e-5
    assert abs(out[0].item() - min(1.0, 0.01 / (norm + 1e-6)) * 0.5) < 1e-7


# ------------------------------------------------------------------------------------------------ headline operand shapes (BASELINE configs 2-4)
# The shapes bench.py and tools/bench_infer.py time: M = B*N = 65,536 token rows into K / N in {1152, 3456, 4608}; dW with K = 65,536 and
# the library's own split-K choice; attention at N = 4096 (8192 workgroups per launch at batch 16) and N_q = 16384 against N_kv = 4096.
# References: fp32 torch on the same bf16-rounded operands, computed on the GPU (per head for attention).  Each test prints its rel-L2.
M_TOK = 65536


def _gpu_rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return torch.randn(*shape, generator=g, device="cuda") * scale


@pytest.mark.parametrize("N,K,flavour", [(3456, 1152, "bias"), (1152, 1152, "bias"), (4608, 1152, "gelu_save_grad"), (1152, 4608, "bias"),
                                         (1152, 4608, "add_aux")])
def test_gemm_nt_headline_shapes(ops, N, K, flavour):
    """qkv / proj / fc1 (+GELU, GELU' saved) / fc2 forward GEMMs at M = 65,536: persistent 256x256 kernel, 4.5 / 13.5 / 18 tile columns,
    half-width remainder items with K = 1152 and K = 4608."""
    a, w, b = bf(_gpu_rnd(M_TOK, K, seed=1)), bf(_gpu_rnd(N, K, scale=K ** -0.5, seed=2)), _gpu_rnd(N, seed=3)
    pre = a.float() @ w.float().t() + b
    if flavour == "bias":
        e = rel_l2(ops.gemm(a, w, ops.NT, bias=b).float(), pre)
    elif flavour == "add_aux":
        aux = bf(_gpu_rnd(M_TOK, N, seed=5))
        e = rel_l2(ops.gemm(a, w, ops.NT, bias=b, act=ops.ACT_ADD_AUX, aux=aux).float(), pre + aux.float())
    else:
        out2 = torch.empty(M_TOK, N, dtype=_opd(), device="cuda")
        out = ops.gemm(a, w, ops.NT, bias=b, act=ops.ACT_GELU_SAVE_GRAD, out2=out2)
        x = pre.clone().requires_grad_(True)
        F.gelu(x, approximate="tanh").backward(torch.ones_like(x))
        e = max(rel_l2(out.float(), F.gelu(pre, approximate="tanh")), rel_l2(out2.float(), x.grad))
    print(f"\n[NT {M_TOK}x{N}x{K} {flavour}] rel-L2 {e:.2e} (bound {BF16_TOL:.0e})")
    record_parity(f"gemm NT {M_TOK}x{N}x{K} {flavour} vs fp32", e, BF16_TOL)
    assert e < BF16_TOL


@pytest.mark.parametrize("M,N,K,bias,strided", [(2048, 256, 256, True, False), (2048, 384, 256, True, False), (4096, 1152, 1152, True, False), (8192, 3456, 1152, True, True),
                                                (65536, 1152, 1152, True, False), (16384, 4608, 1152, False, False), (16384, 1152, 4608, True, False),
                                                (2304, 1280, 384, True, True), (65536, 256, 384, False, False)])
def test_gemm_nt4_one_wave_per_simd(ops, monkeypatch, M, N, K, bias, strided):
    """gemm_nt4_kernel (csrc/gemm_nt4.hip: one wave per SIMD, 128 x 128 per wave, LDS-DMA line pairs) forced on for every call it can take, against fp32 and
    BIT FOR BIT against the eight-wave ping-pong kernel (same k order, same MFMA shape, same epilogue arithmetic): full and half-width items, 8 .. 144
    k-units, fewer / more items than one round of the CUs, operands and outputs that are column slices of wider tensors, with and without bias (the
    no-bias instance once re-used the registers of an in-flight dummy load), and run-to-run reproducibility of the persistent item stream."""
    a = bf(_gpu_rnd(M, K + (64 if strided else 0), seed=1))[:, :K]
    w, b = bf(_gpu_rnd(N, K, scale=K ** -0.5, seed=2)), (_gpu_rnd(N, seed=3) if bias else None)
    outs = []
    for mode in ("1", "1", "0"):
        monkeypatch.setenv("PXA_GEMM_NT4", mode)
        o = torch.full((M, N + (128 if strided else 0)), float("nan"), dtype=_opd(), device="cuda")[:, :N]
        outs.append(ops.gemm(a, w, ops.NT, bias=b, out=o))
    ref = a.float() @ w.float().t() + (b if bias else 0)
    e4, e8 = rel_l2(outs[0].float(), ref), rel_l2(outs[2].float(), ref)
    print(f"\n[NT4 {M}x{N}x{K} bias {bias} strided {strided}] rel-L2 {e4:.2e} (ping-pong kernel {e8:.2e}), bit-identical {torch.equal(outs[0], outs[2])}")
    record_parity(f"gemm_nt4 {M}x{N}x{K} vs fp32", e4, BF16_TOL)
    assert e4 < BF16_TOL and torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("layout", ["NT", "NN"])
def test_gemm_item_schedulers_agree(ops, layout):
    """The persistent token GEMMs hand their items out by static split (default on one GPU) or from per-XCD cursors (pxa_gemm_set_dynamic_items: what dp.py turns
    on beside the bucket all-reduces): same items, same arithmetic - bit-identical outputs, full and half-width items, more items than CUs."""
    from pixart_sigma_amd import lib
    M, N, K = 16384, 1152, 1152
    a, b = bf(_gpu_rnd(M, K, seed=1)), _gpu_rnd(N, seed=3)
    w = bf(_gpu_rnd(N, K, scale=K ** -0.5, seed=2)) if layout == "NT" else bf(_gpu_rnd(K, N, scale=K ** -0.5, seed=2))
    L = lib.load()
    prev = L.pxa_gemm_set_dynamic_items(0)
    try:
        o_static = ops.gemm(a, w, getattr(ops, layout), bias=b).clone()
        o_static_desc = ops.gemm(a, w, getattr(ops, layout), bias=b, descending=True).clone()     # round 5: every XCD walks its range from the end
        assert L.pxa_gemm_set_dynamic_items(1) == 0
        o_dyn = ops.gemm(a, w, getattr(ops, layout), bias=b).clone()
        o_dyn_desc = ops.gemm(a, w, getattr(ops, layout), bias=b, descending=True).clone()
        assert L.pxa_gemm_set_dynamic_items(0) == 1
    finally:
        L.pxa_gemm_set_dynamic_items(prev)
    ref = a.float() @ (w.float().t() if layout == "NT" else w.float()) + b
    assert rel_l2(o_static.float(), ref) < BF16_TOL and torch.equal(o_static, o_dyn)
    assert torch.equal(o_static, o_static_desc) and torch.equal(o_static, o_dyn_desc)
    # ragged shape (a remainder column, a partial last row tile), descending, with an epilogue flavour that reads aux
    M2, N2, K2 = 2300, 1096 if layout == "NN" else 1152, 640
    a2 = bf(_gpu_rnd(M2, K2, seed=4))
    w2 = bf(_gpu_rnd(N2, K2, scale=K2 ** -0.5, seed=5)) if layout == "NT" else bf(_gpu_rnd(K2, N2, scale=K2 ** -0.5, seed=5))
    kw = dict(bias=_gpu_rnd(N2, seed=6)) if layout == "NT" else dict(act=ops.ACT_MUL_AUX, aux=bf(_gpu_rnd(M2, N2, seed=7)))
    assert torch.equal(ops.gemm(a2, w2, getattr(ops, layout), **kw), ops.gemm(a2, w2, getattr(ops, layout), descending=True, **kw))


@pytest.mark.parametrize("K,N,flavour", [(3456, 1152, "plain"), (1152, 1152, "plain"), (4608, 1152, "plain"), (1152, 4608, "mul_aux_colsum")])
def test_gemm_nn_headline_shapes(ops, K, N, flavour):
    """dX = dY W at M = 65,536: qkv / proj / fc1 input gradients and the fc2 input gradient times the saved GELU' with the fused fc1
    bias-gradient column sums (K = reduction over the layer's output features)."""
    dy, w = bf(_gpu_rnd(M_TOK, K, seed=1)), bf(_gpu_rnd(K, N, scale=K ** -0.5, seed=2))
    ref = dy.float() @ w.float()
    if flavour == "plain":
        e = rel_l2(ops.gemm(dy, w, ops.NN).float(), ref)
    else:
        aux = bf(_gpu_rnd(M_TOK, N, seed=5))
        part = torch.zeros(ops.COLSUM_SLOTS, N, device="cuda")
        out = ops.gemm(dy, w, ops.NN, act=ops.ACT_MUL_AUX, aux=aux, colsum=part)
        e = rel_l2(out.float(), ref * aux.float())
        e_cs = rel_l2(part.sum(0), out.float().sum(0))
        print(f"\n[NN colsum] rel-L2 vs column sums of the stored values {e_cs:.2e}")
        assert e_cs < 1e-4
    print(f"\n[NN {M_TOK}x{N}x{K} {flavour}] rel-L2 {e:.2e} (bound {BF16_TOL:.0e})")
    record_parity(f"gemm NN {M_TOK}x{N}x{K} {flavour} vs fp32", e, BF16_TOL)
    assert e < BF16_TOL


@pytest.mark.parametrize("M,N", [(3456, 1152), (1152, 1152), (4608, 1152), (1152, 4608), (2304, 1152)])
def test_gemm_tn_weight_gradient_k65536(ops, M, N):
    """dW[M][N] += sum over K = 65,536 tokens of dY[k][M] X[k][N], split_k = 0 (the library chooses tile shape and split; partial slabs +
    one reduce launch), accumulated twice into the fp32 gradient.  Reference in fp64 on a 256-row band (fp32 torch beside it)."""
    a, b = bf(_gpu_rnd(M_TOK, M, seed=1)), bf(_gpu_rnd(M_TOK, N, seed=2))
    out = torch.zeros(M, N, device="cuda")
    ops.gemm(a, b, ops.TN, out_f32=out, accumulate=True, split_k=0)
    rows = slice(M // 2 - 128, M // 2 + 128)
    ref64 = a[:, rows].double().t() @ b.double()
    e64 = rel_l2(out[rows], ref64)
    e32 = rel_l2(out, a.float().t() @ b.float())
    ops.gemm(a, b, ops.TN, out_f32=out, accumulate=True, split_k=0)
    e2 = rel_l2(out[rows], 2 * ref64)
    print(f"\n[TN dW {M}x{N} K={M_TOK}] rel-L2 vs fp64 {e64:.2e}, vs torch fp32 {e32:.2e}, after 2nd accumulate {e2:.2e} (bound 2e-5)")
    assert e64 < 2e-5 and e2 < 2e-5 and e32 < 5e-5


def _attn_ref_heads(q, k, v, do):
    """fp32 softmax attention + its gradients, one (batch, head) at a time.  q (B,Nq,H,72), k/v (B,Nk,H,72), do like q."""
    B, Nq, H, Dh = q.shape
    o, dq, dk, dv = torch.empty_like(q, dtype=torch.float32), torch.empty_like(q, dtype=torch.float32), torch.empty_like(k, dtype=torch.float32), torch.empty_like(v, dtype=torch.float32)
    for b in range(B):
        for h in range(H):
            qq, kk, vv = (t[b, :, h].float().clone().requires_grad_(True) for t in (q, k, v))
            p = torch.softmax((qq @ kk.t()) * Dh ** -0.5, dim=-1)
            oo = p @ vv
            oo.backward(do[b, :, h].float())
            o[b, :, h], dq[b, :, h], dk[b, :, h], dv[b, :, h] = oo.detach(), qq.grad, kk.grad, vv.grad
    return o, dq, dk, dv


@pytest.mark.parametrize("B,H,Nq,Nk", [(2, 16, 4096, 4096), (1, 16, 16384, 4096)])
def test_attention_headline_shapes(ops, B, H, Nq, Nk):
    """Self-attention of the 1024px training step (N = 4096: 64 key tiles per query tile, 32 query... per key block in the backward
    kernels) and the 2K KV-compressed layers (N_q = 16384 against N_kv = 4096), forward + backward, vs fp32 attention per head."""
    C = H * 72
    q, k, v, do = bf(_gpu_rnd(B, Nq, C, seed=1)), bf(_gpu_rnd(B, Nk, C, seed=2)), bf(_gpu_rnd(B, Nk, C, seed=3)), bf(_gpu_rnd(B, Nq, C, seed=4))
    o = torch.empty(B, Nq, C, dtype=_opd(), device="cuda")
    lse = torch.empty(B, H, Nq, device="cuda")
    sq, sk = (Nq * C, C, 72), (Nk * C, C, 72)
    ops.attention_fwd(q, k, v, o, lse, B, H, Nq, Nk, (sq, sk, sk, sq))
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    delta = torch.empty(B, H, Nq, device="cuda")
    ops.attention_bwd(q, k, v, o, do, lse, delta, dq, dk, dv, B, H, Nq, Nk, (sq, sk, sk, sq), (sq, sk, sk))
    ro, rdq, rdk, rdv = _attn_ref_heads(q.view(B, Nq, H, 72), k.view(B, Nk, H, 72), v.view(B, Nk, H, 72), do.view(B, Nq, H, 72))
    errs = {n: rel_l2(t.float().view_as(r), r) for n, t, r in (("o", o, ro), ("dq", dq, rdq), ("dk", dk, rdk), ("dv", dv, rdv))}
    print(f"\n[attention B{B} H{H} Nq{Nq} Nk{Nk}] rel-L2 " + " ".join(f"{n} {e:.2e}" for n, e in errs.items()) + f" (bounds {BF16_TOL:.0e} / {2 * BF16_TOL:.0e})")
    for n, e in errs.items():
        record_parity(f"attention B{B} H{H} Nq{Nq} Nk{Nk}: {n} vs fp32 attention", e, BF16_TOL if n == "o" else 2 * BF16_TOL)
    assert errs["o"] < BF16_TOL
    assert max(errs["dq"], errs["dk"], errs["dv"]) < 2 * BF16_TOL


@pytest.mark.parametrize("mode,dqm", [("0", "0"), ("1", "1"), ("2", "1"), ("2", "0"), ("3", "1"), ("4", "1"), ("4", "4"), ("2", "4"), ("5", "4")])
@pytest.mark.parametrize("B,H,Nq,Nk,lens", [(2, 3, 130, 77, None), (1, 2, 64, 1024, None), (2, 2, 520, 200, None), (1, 4, 96, 96, None),
                                             (3, 16, 160, 300, [300, 7, 64]), (2, 16, 1024, 1024, None), (1, 2, 200, 40, None),
                                             (1, 2, 128, 256, None), (2, 3, 192, 512, None), (1, 2, 1024, 256, None), (1, 16, 2048, 1024, None), (2, 2, 960, 960, None), (1, 3, 320, 576, None)])
def test_attention_dkv_kernel_modes(ops, monkeypatch, mode, dqm, B, H, Nq, Nk, lens):
    """The three dK/dV kernels of csrc/attn.hip (PXA_ATTN_DKV: 0 = round-2 kernel, 1 = lse / delta through the matrix products + three-stage ring,
    2 = + hand-placed software pipeline with asm LDS reads, 3 = 512-thread workgroups whose two waves per SIMD alternate matrix and softmax phases
    in lock-step, 4 = round 4: one wave per SIMD with 64 keys per wave and asm-owned accumulator registers - dense keys in whole 256-key blocks and
    whole 64-query tiles, the launcher falls back to 2 elsewhere; 5 = 4 with the second products on 16-row MFMA tiles - the default where it applies) and the three dQ kernels (PXA_ATTN_DQ: 0 = round-2 kernel, 1 = hand-placed pipeline; its ragged
    last key tile runs the masked compiler-scheduled path; 4 = round 4: one wave per SIMD with 64 queries per wave - dense keys in whole 64-key tiles, the
    launcher falls back to 1 elsewhere) against fp32 attention per head: ragged query tiles (Nq % 64 != 0: sentinel stats rows),
    one / many key blocks, fewer than 64 keys, partial key waves, packed varlen text keys with inactive waves, and the bias-gradient column sums."""
    monkeypatch.setenv("PXA_ATTN_DKV", mode)
    monkeypatch.setenv("PXA_ATTN_DQ", dqm)
    C = H * 72
    q, do = bf(_gpu_rnd(B, Nq, C, seed=1)), bf(_gpu_rnd(B, Nq, C, seed=4))
    o = torch.empty(B, Nq, C, dtype=_opd(), device="cuda")
    lse = torch.empty(B, H, Nq, device="cuda")
    delta = torch.empty(B, H, Nq, device="cuda")
    dq = torch.empty_like(q)
    part = torch.zeros(ops.COLSUM_SLOTS, 2 * C, device="cuda")
    if lens is None:
        k, v = bf(_gpu_rnd(B, Nk, C, seed=2)), bf(_gpu_rnd(B, Nk, C, seed=3))
        sq, sk = (Nq * C, C, 72), (Nk * C, C, 72)
        ops.attention_fwd(q, k, v, o, lse, B, H, Nq, Nk, (sq, sk, sk, sq))
        dk, dv = torch.empty_like(k), torch.empty_like(v)
        ops.attention_bwd(q, k, v, o, do, lse, delta, dq, dk, dv, B, H, Nq, Nk, (sq, sk, sk, sq), (sq, sk, sk), colsums=(None, part[:, :C], part[:, C:]))
        _, rdq, rdk, rdv = _attn_ref_heads(q.view(B, Nq, H, 72), k.view(B, Nk, H, 72), v.view(B, Nk, H, 72), do.view(B, Nq, H, 72))
        rdk, rdv = rdk.reshape(B, Nk, C), rdv.reshape(B, Nk, C)
    else:
        tot, starts = sum(lens), [sum(lens[:i]) for i in range(len(lens))]
        kv = bf(_gpu_rnd(tot, 2 * C, seed=2))
        ks, kl = torch.tensor(starts, dtype=torch.int32, device="cuda"), torch.tensor(lens, dtype=torch.int32, device="cuda")
        st = ((Nq * C, C, 72), (0, 2 * C, 72), (0, 2 * C, 72), (Nq * C, C, 72))
        ops.attention_fwd(q, kv[:, :C], kv[:, C:], o, lse, B, H, Nq, max(lens), st, kv_start=ks, kv_len=kl, max_kv_len=max(lens))
        dkv = torch.zeros_like(kv)
        dk, dv = dkv[:, :C], dkv[:, C:]
        ops.attention_bwd(q, kv[:, :C], kv[:, C:], o, do, lse, delta, dq, dk, dv, B, H, Nq, max(lens), st, (st[0], st[1], st[2]),
                          colsums=(None, part[:, :C], part[:, C:]), kv_start=ks, kv_len=kl, max_kv_len=max(lens))
        rdq, rdk, rdv = torch.empty(B, Nq, H, 72, device="cuda"), torch.empty(tot, C, device="cuda"), torch.empty(tot, C, device="cuda")
        for b, (s0, n) in enumerate(zip(starts, lens)):
            _, a, bk, bv = _attn_ref_heads(q[b:b + 1].view(1, Nq, H, 72), kv[s0:s0 + n, :C].reshape(1, n, H, 72), kv[s0:s0 + n, C:].reshape(1, n, H, 72),
                                           do[b:b + 1].view(1, Nq, H, 72))
            rdq[b], rdk[s0:s0 + n], rdv[s0:s0 + n] = a[0], bk.reshape(n, C), bv.reshape(n, C)
    errs = {"dq": rel_l2(dq.float().view_as(rdq), rdq), "dk": rel_l2(dk.float(), rdk), "dv": rel_l2(dv.float(), rdv)}
    print(f"\n[dK/dV kernel mode {mode}, dQ kernel {dqm}, B{B} H{H} Nq{Nq} Nk{Nk}] " + " ".join(f"{n} {e:.2e}" for n, e in errs.items()))
    assert max(errs.values()) < 2 * BF16_TOL, errs
    for i, ref in enumerate((rdk, rdv)):                        # fused bias-gradient column sums
        got, want = part.sum(0)[i * C:(i + 1) * C], ref.reshape(-1, C).sum(0)
        assert (got - want).norm() < 5e-3 * ref.norm() + 1e-6, (i, (got - want).norm().item(), ref.norm().item())


@pytest.mark.parametrize("B,H,Nq,Nk,scale,drift", [(2, 3, 256, 64, 1.0, 0.0), (2, 3, 256, 128, 1.0, 0.0), (1, 2, 300, 192, 1.0, 0.0), (2, 16, 1024, 1024, 1.0, 0.0),
                                                     (1, 2, 512, 1024, 6.0, 3.0), (1, 2, 512, 4096, 3.0, 0.0), (1, 16, 4096, 1024, 1.0, 0.0), (1, 2, 256, 512, 0.02, 0.0)])
def test_attention_fwd4_one_wave_per_simd(ops, monkeypatch, B, H, Nq, Nk, scale, drift):
    """attn_fwd4_kernel (round 4: one wave per SIMD, 4-slot K / V rings, deferred maximum in a slow path) forced on wherever it applies, against fp32
    attention and against attn_fwd2_kernel: 1 .. 64 key tiles (ring prologue, clamped re-fetches past the last tile), ragged query blocks, tiny scores,
    and scores whose level drifts upward along the keys so that the deferred maximum moves on most tiles (the rescale path: O, l, the pending S').
    The fp16 build folds scale and maximum into the first product (one more rounding of the query operand: bounds 1.3 x the two-wave kernel's error)."""
    C = H * 72
    q, k, v = bf(_gpu_rnd(B, Nq, C, seed=1) * scale), _gpu_rnd(B, Nk, C, seed=2) * scale, bf(_gpu_rnd(B, Nk, C, seed=3))
    if drift:
        k = k + torch.linspace(0, drift, Nk, device="cuda")[None, :, None] * q.float().mean(1, keepdim=True).sign()
    k = bf(k)
    st = ((Nq * C, C, 72), (Nk * C, C, 72), (Nk * C, C, 72), (Nq * C, C, 72))
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("PXA_ATTN_FWD4", mode)
        o = torch.full((B, Nq, C), float("nan"), dtype=_opd(), device="cuda")
        lse = torch.full((B, H, Nq), float("nan"), device="cuda")
        ops.attention_fwd(q, k, v, o, lse, B, H, Nq, Nk, st)
        res[mode] = (o, lse)
    qf, kf, vf = (t.float().view(B, -1, H, 72).transpose(1, 2) for t in (q, k, v))
    sc = (qf @ kf.transpose(-1, -2)) * 72 ** -0.5
    ro, rl = (sc.softmax(-1) @ vf).transpose(1, 2).reshape(B, Nq, C), torch.logsumexp(sc, -1) / math.log(2)
    e4, e2 = rel_l2(res["1"][0].float(), ro), rel_l2(res["0"][0].float(), ro)
    l4, l2 = (res["1"][1] - rl).abs().max().item(), (res["0"][1] - rl).abs().max().item()
    print(f"\n[fwd4 B{B} H{H} Nq{Nq} Nk{Nk} x{scale} drift {drift}] o rel-L2 {e4:.2e} (two-wave kernel {e2:.2e}), lse max abs {l4:.1e} ({l2:.1e})")
    record_parity(f"attn_fwd4 B{B} H{H} Nq{Nq} Nk{Nk} x{scale}: o vs fp32 attention", e4)
    assert torch.isfinite(res["1"][0].float()).all() and torch.isfinite(res["1"][1]).all()
    if F16_BUILD and scale > 1:       # folded scale: the second rounding of the query operand grows with the score level (|c S| up to ~400 here)
        assert e4 < 2e-3 and l4 < 0.1
    else:
        assert e4 < max(BF16_TOL, 1.3 * e2) and l4 < max(2e-3, 1.5 * l2, 2e-5 * rl.abs().max().item())     # lse itself reaches several hundred in the scaled cases


def test_attention_full_grid_b16(ops):
    """The benchmark's own launch geometry (B16 H16 N4096: 8,192 workgroups per backward kernel, 4,096 in the forward): every workgroup of the XCD-aware
    block order (csrc/attn.hip block_coords) writes its rows.  Checked on a strided subset of (batch, head) pairs against fp32 attention per head (the
    full set is 256 heads x 4096^2) and, for ALL pairs, through a checksum against the round-2 dK/dV kernel."""
    B, H, N = 16, 16, 4096
    C = H * 72
    qkv = bf(_gpu_rnd(B, N, 3 * C, seed=1))
    do = bf(_gpu_rnd(B, N, C, seed=2))
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    o = torch.full((B, N, C), float("nan"), dtype=_opd(), device="cuda")
    lse, delta = torch.empty(B, H, N, device="cuda"), torch.empty(B, H, N, device="cuda")
    s3 = (N * 3 * C, 3 * C, 72)
    st = (s3, s3, s3, (N * C, C, 72))
    ops.attention_fwd(q, k, v, o, lse, B, H, N, N, st)
    dqkv = torch.full_like(qkv, float("nan"))
    ops.attention_bwd(q, k, v, o, do, lse, delta, dqkv[..., :C], dqkv[..., C:2 * C], dqkv[..., 2 * C:], B, H, N, N, st, (s3, s3, s3))
    assert torch.isfinite(o.float()).all() and torch.isfinite(dqkv.float()).all()      # no workgroup of the grid was skipped
    for b, h in ((0, 0), (3, 7), (8, 15), (15, 4)):
        sl = slice(h * 72, (h + 1) * 72)
        ro, rdq, rdk, rdv = _attn_ref_heads(q[b:b + 1, :, sl].reshape(1, N, 1, 72), k[b:b + 1, :, sl].reshape(1, N, 1, 72), v[b:b + 1, :, sl].reshape(1, N, 1, 72),
                                            do[b:b + 1, :, sl].reshape(1, N, 1, 72))
        assert rel_l2(o[b, :, sl].float(), ro[0, :, 0]) < BF16_TOL
        for i, r in enumerate((rdq, rdk, rdv)):
            assert rel_l2(dqkv[b, :, i * C + h * 72:i * C + (h + 1) * 72].float(), r[0, :, 0]) < 2 * BF16_TOL, (b, h, i)
    os.environ["PXA_ATTN_DKV"] = "0"
    try:
        ref = torch.empty_like(qkv)
        ops.attention_bwd(q, k, v, o, do, lse, delta, ref[..., :C], ref[..., C:2 * C], ref[..., 2 * C:], B, H, N, N, st, (s3, s3, s3))
    finally:
        del os.environ["PXA_ATTN_DKV"]
    per_head = (dqkv.float() - ref.float()).view(B, N, 3, H, 72).pow(2).sum((1, 4)).sqrt() / ref.float().view(B, N, 3, H, 72).pow(2).sum((1, 4)).sqrt()
    assert per_head.max() < BF16_TOL, per_head.max().item()        # two kernels, same products: they differ by operand rounding of lse / delta only


# ------------------------------------------------------------------------------------------------ fused diffusion loss
def test_fused_iddpm_loss_matches_torch_expressions(ops, monkeypatch):
    """csrc/loss.hip (one launch per direction) against the elementwise torch statement of GaussianDiffusion.training_losses in
    pixart_sigma_amd/diffusion/iddpm.py (itself pinned to the reference by the CPU golden tests): loss terms and d loss / d model_output,
    including a t = 0 sample whose x0 covers the three branches of the discretized-Gaussian log-likelihood."""
    from pixart_sigma_amd import IDDPM
    diff = IDDPM(str(1000), learn_sigma=True, pred_sigma=True, snr=False)
    B, C, H, W = 4, 4, 16, 24
    g = torch.Generator().manual_seed(3)
    x0 = torch.randn(B, C, H, W, generator=g).cuda()
    x0[0] = (torch.rand(C, H, W, generator=g) * 2.4 - 1.2).cuda()           # t = 0 sample: values below -0.999, inside, above 0.999
    noise = torch.randn(B, C, H, W, generator=g).cuda()
    t = torch.tensor([0, 1, 500, 999]).cuda()
    fixed = (torch.randn(B, 2 * C, H, W, generator=g) * 0.7).cuda()
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("PXA_FUSED_LOSS", mode)
        out = fixed.clone().requires_grad_(True)
        terms = diff.training_losses(lambda x, timestep, **kw: out, x0, t, noise=noise)
        w = torch.tensor([0.3, 1.0, 2.0, 0.5]).cuda()
        (terms["loss"] * w).sum().backward()
        res[mode] = (terms["mse"].detach(), terms["vb"].detach(), out.grad.clone())
    for name, a, b in zip(("mse", "vb", "d_out"), res["1"], res["0"]):
        e = rel_l2(a, b)
        print(f"fused loss {name}: rel-L2 vs torch expressions {e:.2e}")
        assert e < 2e-5, name
    assert (res["1"][2][:, C:].abs().sum() > 0) and (res["1"][2][:, :C].abs().sum() > 0)
    assert float(res["1"][1][0]) != float(res["0"][1][1])                  # the t = 0 (NLL) sample differs from the KL samples


# ------------------------------------------------------------------------------------------------ round 5: softmax scale folded into q
@pytest.mark.parametrize("logit_std", [1.0, 8.0, 24.0])
@pytest.mark.parametrize("B,H,Nq,Nk,dkv,fwd4", [(2, 16, 1024, 1024, "4", "1"), (1, 2, 1024, 256, "5", "1"), (2, 3, 130, 77, "2", "0"), (1, 2, 200, 40, "0", "0"),
                                              (1, 4, 2048, 1024, "4", "0")])
def test_attention_q_prescaled(ops, monkeypatch, B, H, Nq, Nk, dkv, fwd4, logit_std):
    """pxa_attn_args.q_prescaled: q carries scale * log2 e (what engine.py's copy of the qkv weight produces in ONE rounding).  Forward, lse and all three
    gradients against fp32 attention on the queries the operand stands for (q~ / (scale log2 e)), through the one-wave kernels (dkv 4 / 5, whose PRE
    instances drop the multiply in front of exp2; the folded forward) and the two-wave / ragged ones - at N(0,1) scores and at the score levels of a
    trained model (|scale q k| of 8 and 24 standard deviations: ADVICE r04 on the folded forward, whose second rounding of q this mode removes), with
    the SAME bounds at every level."""
    monkeypatch.setenv("PXA_ATTN_DKV", dkv)
    monkeypatch.setenv("PXA_ATTN_FWD4", fwd4)
    C = H * 72
    cpre = ops.Q_PRESCALE
    # queries whose scores have the requested spread: q k^T / sqrt(72) ~ logit_std for unit-variance k
    qraw = rnd(B, Nq, C, seed=1) * logit_std
    qt = bf(qraw * cpre)                                      # the operand: one rounding of the scaled query
    k, v, do = bf(rnd(B, Nk, C, seed=2)), bf(rnd(B, Nk, C, seed=3)), bf(rnd(B, Nq, C, seed=4))
    o = torch.empty(B, Nq, C, dtype=_opd(), device="cuda")
    lse = torch.empty(B, H, Nq, device="cuda")
    sq, sk = (Nq * C, C, 72), (Nk * C, C, 72)
    ops.attention_fwd(qt, k, v, o, lse, B, H, Nq, Nk, (sq, sk, sk, sq), q_prescaled=True)
    dq, dk, dv = torch.empty_like(qt), torch.empty_like(k), torch.empty_like(v)
    delta = torch.empty(B, H, Nq, device="cuda")
    ops.attention_bwd(qt, k, v, o, do, lse, delta, dq, dk, dv, B, H, Nq, Nk, (sq, sk, sk, sq), (sq, sk, sk), q_prescaled=True)
    qeff = (qt.float() / cpre).view(B, Nq, H, 72)             # the unscaled queries the operand stands for (exact in fp32 up to 1 ulp)
    ro, rdq, rdk, rdv = _attn_ref_heads(qeff, k.view(B, Nk, H, 72), v.view(B, Nk, H, 72), do.view(B, Nq, H, 72))
    sref = torch.einsum("bqhd,bkhd->bhqk", qeff, k.float().view(B, Nk, H, 72)) * 72 ** -0.5
    dl = (lse - torch.logsumexp(sref, -1) / math.log(2)).abs().max().item()
    errs = {n: rel_l2(t.float().view_as(r), r) for n, t, r in (("o", o, ro), ("dq", dq, rdq), ("dk", dk, rdk), ("dv", dv, rdv))}
    print(f"\n[q prescaled B{B} H{H} Nq{Nq} Nk{Nk} dkv{dkv} fwd4={fwd4} logit std {logit_std}] " + " ".join(f"{n} {e:.2e}" for n, e in errs.items()) + f" |dlse| {dl:.1e}")
    for n, e in errs.items():
        record_parity(f"attention q_prescaled B{B} H{H} Nq{Nq} Nk{Nk} dkv{dkv} logit_std {logit_std}: {n}", e, BF16_TOL if n == "o" else 2 * BF16_TOL)
    assert errs["o"] < BF16_TOL and dl < (2e-2 if F16_BUILD else 1.5e-1) * max(1.0, logit_std / 8)
    assert max(errs["dq"], errs["dk"], errs["dv"]) < 2 * BF16_TOL


def test_scale_copy_blocks(ops):
    """pxa_scale_copy_f32: strided blocks of a flat fp32 buffer -> operand-type / fp32 copies whose leading part carries the factor (one rounding)."""
    nb, n_total, n_scaled, stride = 3, 3456 * 16, 1152 * 16, 3456 * 16 + 256
    src = rnd(nb * stride + 64, seed=9)
    out = torch.empty(nb, n_total, dtype=_opd(), device="cuda")
    outf = torch.empty(nb, n_total, device="cuda")
    ops.scale_copy(src, stride, nb, n_scaled, n_total, ops.Q_PRESCALE, out_bf16=out)
    ops.scale_copy(src, stride, nb, n_scaled, n_total, ops.Q_PRESCALE, out_f32=outf)
    for b in range(nb):
        ref = src[b * stride:b * stride + n_total].clone()
        ref[:n_scaled] *= torch.tensor(ops.Q_PRESCALE, dtype=torch.float32, device="cuda")
        assert torch.equal(outf[b], ref)
        assert torch.equal(out[b], ref.to(_opd()))


@pytest.mark.parametrize("M,K,N", [(16, 256, 1152), (16, 1152, 6912), (64, 1152, 1152), (3, 1152, 1152), (33, 256, 1152)])
def test_cond_linear_f32(ops, M, K, N):
    """pxa_linear_f32_fwd / _bwd (csrc/condlin.hip: the t_embedder / t_block / size-embedder linears, fp32 end to end) against torch's fp32 linear and
    its autograd, through the module the model uses (_CondLinear keeps nn.Linear's names and init)."""
    from pixart_sigma_amd.model.nets.PixArtMS import _CondLinear
    torch.manual_seed(0)
    lin = _CondLinear(K, N).cuda()
    x = rnd(M, K, seed=1).requires_grad_(True)
    g = rnd(M, N, seed=2)
    y = lin(x)
    y.backward(g)
    got = (y.detach(), x.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone())
    x2 = x.detach().double().requires_grad_(True)
    w2, b2 = lin.weight.detach().double().requires_grad_(True), lin.bias.detach().double().requires_grad_(True)
    y2 = F.linear(x2, w2, b2)
    y2.backward(g.double())
    for name, a, b in zip(("y", "dx", "dw", "db"), got, (y2.detach(), x2.grad, w2.grad, b2.grad)):
        e = rel_l2(a.double(), b)
        assert e < 2e-6, (name, e)
    # a second backward accumulates into .grad like any nn.Linear
    lin(x).backward(g)
    assert rel_l2(lin.weight.grad.double(), 2 * w2.grad) < 2e-6


def test_gemm_counted_waits_match_full_waits(tmp_path):
    """the persistent GEMM's epilogue waits are COUNTED (`s_waitcnt vmcnt(N)`, N mirroring the issue order of the prefetch / aux / bias / store
    instructions by hand); a count that over-estimates would let registers be read before they land.  The debug build -DGEMM_WAIT_ALL=1 turns every counted wait
    into vmcnt(0): every epilogue flavour must give the same BITS under both libraries (column-sum / GroupNorm partials, added with atomics, to 1e-5)."""
    import shutil
    import subprocess
    import sys
    from conftest import ROOT
    from pixart_sigma_amd import lib as L_
    if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc on this box: the debug variant cannot be built")
    env = dict(os.environ, VARIANT_OPERAND=L_.OPERAND, PXA_OPERAND_DTYPE=L_.OPERAND)
    env.pop("PXA_LIB_PATH", None)
    name = "waitall_" + L_.OPERAND
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "build_variant.py"), name, "csrc/gemm.hip", "-DGEMM_WAIT_ALL=1"], capture_output=True, text=True,
                       env=env, cwd=ROOT, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    variant = os.path.join(ROOT, "pixart_sigma_amd", "variants", f"lib_{name}.so")
    outs = {}
    for tag, e in (("product", env), ("waitall", dict(env, PXA_LIB_PATH=variant))):
        out = str(tmp_path / f"{tag}.pt")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gemm_flavour_dump.py"), out], capture_output=True, text=True, env=e, cwd=ROOT, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        outs[tag] = torch.load(out, weights_only=False)
    assert set(outs["product"]) == set(outs["waitall"]) and len(outs["product"]) >= 12
    for k, a in outs["product"].items():
        b = outs["waitall"][k]
        assert torch.isfinite(a.float()).all(), k
        if k.endswith("_sum"):
            assert rel_l2(a, b) < 1e-5, k
        else:
            assert torch.equal(a, b), (k, rel_l2(a, b))
