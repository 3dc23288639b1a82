"""attn_bwd_dkv4_kernel (one wave per SIMD, 64 keys per wave) against attn_bwd_dkv2_kernel<1> and fp32 attention gradients, and the two timed at the
headline shape (dK/dV kernel alone: dq = NULL, pre-pass skipped through PXA_ATTN_BWD_NO_PREPASS after one full call filled the workspace).
Usage (GPU box): python tools/kbench_dkv4.py [check|time|all]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from box_sampler import Sampler
from pixart_sigma_amd import ops

dev, OPD = "cuda", ops.BF16
NEW = os.environ.get("KB_DKV_MODE", "4")      # the dK/dV kernel under test (4: one wave per SIMD; 5: + 16-row second products)


def rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def bwd(q, k, v, o, do, lse, B, H, Nq, Nk, mode, dq=True):
    os.environ["PXA_ATTN_DKV"] = mode
    C = H * 72
    dqt = torch.full_like(q, float("nan")) if dq else None
    dk, dv = torch.full_like(k, float("nan")), torch.full_like(v, float("nan"))
    delta = torch.empty(B, H, Nq, device=dev)
    sq, sk = (Nq * C, C, 72), (Nk * C, C, 72)
    ops.attention_bwd(q, k, v, o, do, lse, delta, dqt, dk, dv, B, H, Nq, Nk, (sq, sk, sk, sq), (sq, sk, sk))
    torch.cuda.synchronize()
    return dk, dv


def check():
    bad = 0
    g = torch.Generator(device=dev).manual_seed(0)
    tol = 1e-3 if OPD == torch.float16 else 8e-3
    for B, H, Nq, Nk, sc in [(1, 2, 128, 256, 1.0), (2, 3, 192, 512, 1.0), (1, 2, 1024, 256, 1.0), (2, 16, 1024, 1024, 1.0), (1, 4, 4096, 1024, 2.0), (1, 16, 256, 4096, 1.0), (2, 2, 960, 960, 1.0), (1, 3, 320, 576, 1.0)]:
        C = H * 72
        q = (torch.randn(B, Nq, C, device=dev, generator=g) * sc).to(OPD)
        k = (torch.randn(B, Nk, C, device=dev, generator=g) * sc).to(OPD)
        v, do = (torch.randn(B, n, C, device=dev, generator=g).to(OPD) for n in (Nk, Nq))
        o = torch.empty(B, Nq, C, dtype=OPD, device=dev)
        lse = torch.empty(B, H, Nq, device=dev)
        sq, sk = (Nq * C, C, 72), (Nk * C, C, 72)
        ops.attention_fwd(q, k, v, o, lse, B, H, Nq, Nk, (sq, sk, sk, sq))
        dk4, dv4 = bwd(q, k, v, o, do, lse, B, H, Nq, Nk, NEW)
        dk2, dv2 = bwd(q, k, v, o, do, lse, B, H, Nq, Nk, "2")
        qf, kf, vf = (t.float().view(B, -1, H, 72).transpose(1, 2).requires_grad_(True) for t in (q, k, v))
        s = (qf @ kf.transpose(-1, -2)) * 72 ** -0.5
        (s.softmax(-1) @ vf).backward(do.float().view(B, Nq, H, 72).transpose(1, 2))
        rk, rv = kf.grad.transpose(1, 2).reshape(B, Nk, C), vf.grad.transpose(1, 2).reshape(B, Nk, C)
        e = dict(dk4=rel(dk4.float(), rk), dv4=rel(dv4.float(), rv), dk2=rel(dk2.float(), rk), dv2=rel(dv2.float(), rv), dk42=rel(dk4.float(), dk2.float()), dv42=rel(dv4.float(), dv2.float()))
        ok = e["dk4"] < max(tol, 1.2 * e["dk2"]) and e["dv4"] < max(tol, 1.2 * e["dv2"]) and torch.isfinite(dk4.float()).all().item() and torch.isfinite(dv4.float()).all().item()
        bad += not ok
        print(f"B{B} H{H} Nq{Nq} Nk{Nk} x{sc}: " + " ".join(f"{n} {x:.2e}" for n, x in e.items()) + ("  ok" if ok else "  FAIL"), flush=True)
    # full grid
    B, H, N = 16, 16, 4096
    C = H * 72
    qkv = torch.randn(B, N, 3 * C, device=dev, generator=g).to(OPD)
    do = torch.randn(B, N, C, device=dev, generator=g).to(OPD)
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    o = torch.empty(B, N, C, dtype=OPD, device=dev)
    lse, delta = torch.empty(B, H, N, device=dev), torch.empty(B, H, N, device=dev)
    s3 = (N * 3 * C, 3 * C, 72)
    st = (s3, s3, s3, (N * C, C, 72))
    ops.attention_fwd(q, k, v, o, lse, B, H, N, N, st)
    outs = {}
    for mode in (NEW, NEW, "2"):
        os.environ["PXA_ATTN_DKV"] = mode
        d = torch.full_like(qkv, float("nan"))
        ops.attention_bwd(q, k, v, o, do, lse, delta, d[..., :C], d[..., C:2 * C], d[..., 2 * C:], B, H, N, N, st, (s3, s3, s3))
        torch.cuda.synchronize()
        outs.setdefault(mode, []).append(d)
    a, b2 = outs[NEW][0].float(), outs["2"][0].float()
    per_head = (a - b2).view(B, N, 3, H, 72).pow(2).sum((1, 4)).sqrt() / b2.view(B, N, 3, H, 72).pow(2).sum((1, 4)).sqrt()
    rep = torch.equal(outs[NEW][0], outs[NEW][1])
    ok = per_head.max().item() < tol and rep and torch.isfinite(a).all().item()
    bad += not ok
    print(f"full grid B16: worst (head, tensor) vs the two-wave kernel {per_head.max().item():.2e}, bit-reproducible {rep}  {'ok' if ok else 'FAIL'}", flush=True)
    return bad


def check_dq():
    """attn_bwd_dq4_kernel (PXA_ATTN_DQ=4) against attn_bwd_dq2_kernel (=1) and fp32 attention gradients"""
    bad = 0
    g = torch.Generator(device=dev).manual_seed(1)
    tol = 1e-3 if OPD == torch.float16 else 8e-3
    for B, H, Nq, Nk, sc in [(1, 2, 256, 128, 1.0), (2, 3, 300, 192, 1.0), (1, 2, 1024, 256, 1.0), (2, 16, 1024, 1024, 1.0), (1, 4, 512, 4096, 2.0), (1, 16, 4096, 320, 1.0)]:
        C = H * 72
        q = (torch.randn(B, Nq, C, device=dev, generator=g) * sc).to(OPD)
        k = (torch.randn(B, Nk, C, device=dev, generator=g) * sc).to(OPD)
        v, do = (torch.randn(B, n, C, device=dev, generator=g).to(OPD) for n in (Nk, Nq))
        o = torch.empty(B, Nq, C, dtype=OPD, device=dev)
        lse, delta = torch.empty(B, H, Nq, device=dev), torch.empty(B, H, Nq, device=dev)
        sq, sk = (Nq * C, C, 72), (Nk * C, C, 72)
        ops.attention_fwd(q, k, v, o, lse, B, H, Nq, Nk, (sq, sk, sk, sq))
        res = {}
        for mode in ("4", "1"):
            os.environ["PXA_ATTN_DQ"] = mode
            dq = torch.full_like(q, float("nan"))
            ops.attention_bwd(q, k, v, o, do, lse, delta, dq, None, None, B, H, Nq, Nk, (sq, sk, sk, sq), (sq, sk, sk))
            torch.cuda.synchronize()
            res[mode] = dq
        del os.environ["PXA_ATTN_DQ"]
        qf, kf, vf = (t.float().view(B, -1, H, 72).transpose(1, 2).requires_grad_(True) for t in (q, k, v))
        s = (qf @ kf.transpose(-1, -2)) * 72 ** -0.5
        (s.softmax(-1) @ vf).backward(do.float().view(B, Nq, H, 72).transpose(1, 2))
        rq = qf.grad.transpose(1, 2).reshape(B, Nq, C)
        e4, e1, e41 = rel(res["4"].float(), rq), rel(res["1"].float(), rq), rel(res["4"].float(), res["1"].float())
        ok = e4 < max(tol, 1.2 * e1) and torch.isfinite(res["4"].float()).all().item()
        bad += not ok
        print(f"dQ B{B} H{H} Nq{Nq} Nk{Nk} x{sc}: dq4 {e4:.2e} dq2 {e1:.2e} dq4 vs dq2 {e41:.2e}" + ("  ok" if ok else "  FAIL"), flush=True)
    return bad


def time_dq():
    B, H, N, D = 16, 16, 4096, 1152
    R = B * N
    qkv = torch.randn(R, 3 * D, device=dev).to(OPD)
    a = torch.empty(R, D, dtype=OPD, device=dev)
    lse, delta = torch.empty(B, H, N, device=dev), torch.empty(B, H, N, device=dev)
    s3 = (N * 3 * D, 3 * D, 72)
    st = (s3, s3, s3, (N * D, D, 72))
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    ops.attention_fwd(q, k, v, a, lse, B, H, N, N, st)
    da, dqkv = torch.randn(R, D, device=dev).to(OPD), torch.empty_like(qkv)
    ops.attention_bwd(q, k, v, a, da, lse, delta, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:], B, H, N, N, st, (s3, s3, s3))
    fl = 6.0 * B * H * N * N * 72
    os.environ["PXA_ATTN_BWD_NO_PREPASS"] = "1"
    outs = {}
    for mode in ("1", "4", "1", "4"):
        os.environ["PXA_ATTN_DQ"] = mode
        fn = lambda: ops.attention_bwd(q, k, v, a, da, lse, delta, dqkv[:, :D], None, None, B, H, N, N, st, (s3, s3, s3))
        for _ in range(10):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with Sampler() as box:
            e0.record()
            for _ in range(60):
                fn()
            e1.record()
            e1.synchronize()
        t = e0.elapsed_time(e1) / 60 * 1e-3
        outs[mode] = dqkv[:, :D].clone()
        print(f"dQ kernel alone B16 H16 N4096 PXA_ATTN_DQ={mode} lib={os.environ.get('PXA_LIB_PATH', 'default')}: {t * 1e3:7.3f} ms {fl / t / 1e12:7.1f} TF/s  {box.summary()}", flush=True)
    print(f"full grid B16: dq4 vs dq2 rel-L2 {rel(outs['4'].float(), outs['1'].float()):.2e}, finite {torch.isfinite(outs['4'].float()).all().item()}", flush=True)
    del os.environ["PXA_ATTN_BWD_NO_PREPASS"], os.environ["PXA_ATTN_DQ"]


def timeit():
    B, H, N, D = 16, 16, 4096, 1152
    R = B * N
    qkv = torch.randn(R, 3 * D, device=dev).to(OPD)
    a = torch.empty(R, D, dtype=OPD, device=dev)
    lse, delta = torch.empty(B, H, N, device=dev), torch.empty(B, H, N, device=dev)
    s3 = (N * 3 * D, 3 * D, 72)
    st = (s3, s3, s3, (N * D, D, 72))
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    ops.attention_fwd(q, k, v, a, lse, B, H, N, N, st)
    da, dqkv = torch.randn(R, D, device=dev).to(OPD), torch.empty_like(qkv)
    ops.attention_bwd(q, k, v, a, da, lse, delta, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:], B, H, N, N, st, (s3, s3, s3))   # fills the statistics workspace
    fl = 8.0 * B * H * N * N * 72
    os.environ["PXA_ATTN_BWD_NO_PREPASS"] = "1"
    for mode in ("2", NEW, "2", NEW):
        os.environ["PXA_ATTN_DKV"] = mode
        fn = lambda: ops.attention_bwd(q, k, v, a, da, lse, delta, None, dqkv[:, D:2 * D], dqkv[:, 2 * D:], B, H, N, N, st, (s3, s3, s3))
        for _ in range(10):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with Sampler() as box:
            e0.record()
            for _ in range(60):
                fn()
            e1.record()
            e1.synchronize()
        t = e0.elapsed_time(e1) / 60 * 1e-3
        print(f"dK/dV kernel alone B16 H16 N4096 PXA_ATTN_DKV={mode} lib={os.environ.get('PXA_LIB_PATH', 'default')}: {t * 1e3:7.3f} ms {fl / t / 1e12:7.1f} TF/s  {box.summary()}", flush=True)
    del os.environ["PXA_ATTN_BWD_NO_PREPASS"]


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    rc = 0
    if what in ("check", "all"):
        rc = check()
    if what in ("time", "all"):
        timeit()
    if what in ("dq",):
        rc = check_dq()
        time_dq()
    sys.exit(1 if rc else 0)
