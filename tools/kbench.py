"""Kernel micro-benchmarks at the headline shapes (1024px, batch 16): TF/s of each GEMM layout and attention kernel.
Usage (GPU box): python tools/kbench.py [attn|gemm|all]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pixart_sigma_amd import ops

D, H, DFF = 1152, 16, 4608
B, N, L = 16, 4096, 300
R = B * N
dev = "cuda"


def timed(fn, iters=50, warm=5):      # short runs under-read by up to 20 % (clock ramp): keep >= 50 launches
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


LIBREF = bool(int(os.environ.get("KBENCH_LIBREF", "0")))


def rb(*s):
    return torch.randn(*s, device=dev).to(ops.BF16)


def gemm():
    x = rb(R, D)
    for name, n_out, k in (("qkv", 3 * D, D), ("proj", D, D), ("fc1", DFF, D), ("fc2", D, DFF)):
        a = rb(R, k)
        w = (torch.randn(n_out, k, device=dev) * k ** -0.5).to(ops.BF16)
        bias = torch.zeros(n_out, device=dev)
        out = torch.empty(R, n_out, dtype=ops.BF16, device=dev)
        t = timed(lambda: ops.gemm(a, w, ops.NT, bias=bias, out=out))
        fl = 2.0 * R * n_out * k
        print(f"gemm NT {name:5s} M={R} N={n_out} K={k}: {t*1e3:7.3f} ms {fl/t/1e12:7.1f} TF/s")
        dy = rb(R, n_out)
        if LIBREF:          # the vendor library (hipBLASLt through torch) on the same shapes: a yardstick for the report, never the product path
            tl = [timed(lambda: torch.matmul(a, w.t(), out=out)), timed(lambda: torch.matmul(dy, w, out=torch.empty(R, k, dtype=ops.BF16, device=dev))),
                  timed(lambda: torch.matmul(dy.t(), a))]
            print("   hipBLASLt NT/NN/TN: " + " ".join(f"{fl/x/1e12:7.1f}" for x in tl) + " TF/s")
        dx = torch.empty(R, k, dtype=ops.BF16, device=dev)
        t = timed(lambda: ops.gemm(dy, w, ops.NN, out=dx))
        print(f"gemm NN {name:5s} (dX)                      : {t*1e3:7.3f} ms {fl/t/1e12:7.1f} TF/s")
        dw = torch.zeros(n_out, k, device=dev)
        for sk in (0, 2, 4):
            t = timed(lambda: ops.gemm(dy, a, ops.TN, out_f32=dw, accumulate=True, split_k=sk))
            print(f"gemm TN {name:5s} (dW) split_k={sk}            : {t*1e3:7.3f} ms {fl/t/1e12:7.1f} TF/s")


def attn():
    qkv32 = torch.randn(R, 3 * D, device=dev)
    noscale = os.environ.get("PXA_KBENCH_NO_PRESCALE") == "1"     # A/B: the kernel instances that multiply by scale * log2 e themselves (round 4's operands)
    if not noscale:
        qkv32[:, :D] *= ops.Q_PRESCALE               # the step's operands: q carries scale * log2 e (pxa_attn_args.q_prescaled)
    qkv = qkv32.to(ops.BF16)
    del qkv32
    pre = {} if noscale else dict(q_prescaled=True)
    a = torch.empty(R, D, dtype=ops.BF16, device=dev)
    lse = torch.empty(B, H, N, device=dev)
    s3 = (N * 3 * D, 3 * D, 72)
    st = (s3, s3, s3, (N * D, D, 72))
    t = timed(lambda: ops.attention_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], a, lse, B, H, N, N, st, **pre))
    fl = 4.0 * B * N * N * D
    print(f"attn fwd self : {t*1e3:7.3f} ms {fl/t/1e12:7.1f} TF/s")
    da, dqkv, delta = rb(R, D), torch.empty_like(qkv), torch.empty(B, H, N, device=dev)
    t = timed(lambda: ops.attention_bwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], a, da, lse, delta, dqkv[:, :D], dqkv[:, D:2 * D],
                                        dqkv[:, 2 * D:], B, H, N, N, st, (s3, s3, s3), **pre))
    print(f"attn bwd self : {t*1e3:7.3f} ms {2.5*fl/t/1e12:7.1f} TF/s (algorithmic 2.5x fwd)")
    if LIBREF:
        import torch.nn.functional as F
        q4, k4, v4 = (qkv[:, i * D:(i + 1) * D].reshape(B, N, H, 72).transpose(1, 2).contiguous().requires_grad_() for i in range(3))
        try:
            t = timed(lambda: F.scaled_dot_product_attention(q4, k4, v4))
            print(f"   torch SDPA fwd: {t*1e3:7.3f} ms {fl/t/1e12:7.1f} TF/s")
            o = F.scaled_dot_product_attention(q4, k4, v4)
            g = torch.randn_like(o)
            t = timed(lambda: torch.autograd.grad(o, (q4, k4, v4), g, retain_graph=True))
            print(f"   torch SDPA bwd: {t*1e3:7.3f} ms {2.5*fl/t/1e12:7.1f} TF/s")
        except Exception as e:
            print("   torch SDPA unavailable:", repr(e)[:200])
    lens = [L] * B
    q = rb(R, D)
    kv = rb(sum(lens), 2 * D)
    ks = torch.tensor([i * L for i in range(B)], dtype=torch.int32, device=dev)
    kl = torch.tensor(lens, dtype=torch.int32, device=dev)
    sc = ((N * D, D, 72), (0, 2 * D, 72), (0, 2 * D, 72), (N * D, D, 72))
    t = timed(lambda: ops.attention_fwd(q, kv[:, :D], kv[:, D:], a, lse, B, H, N, L, sc, kv_start=ks, kv_len=kl, max_kv_len=L))
    flc = 4.0 * B * N * L * D
    print(f"attn fwd cross: {t*1e3:7.3f} ms {flc/t/1e12:7.1f} TF/s")
    dq, dkv = torch.empty_like(q), torch.empty_like(kv)
    t = timed(lambda: ops.attention_bwd(q, kv[:, :D], kv[:, D:], a, da, lse, delta, dq, dkv[:, :D], dkv[:, D:], B, H, N, L, sc,
                                        ((N * D, D, 72), (0, 2 * D, 72), (0, 2 * D, 72)), kv_start=ks, kv_len=kl, max_kv_len=L))
    print(f"attn bwd cross: {t*1e3:7.3f} ms {2.5*flc/t/1e12:7.1f} TF/s")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    print("lib:", os.environ.get("PXA_LIB_PATH", "default"))
    if what in ("attn", "all"):
        attn()
    if what in ("gemm", "all"):
        gemm()
