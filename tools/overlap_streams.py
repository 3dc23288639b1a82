"""Can an HBM-bound row kernel run beside a persistent GEMM on another stream (one process, two HIP streams)?  The GEMM holds one 512-thread workgroup and
all 160 KiB of LDS per CU; ln_mod_fwd uses no LDS and 36 VGPRs, so its workgroups can co-reside.  Prints GEMM alone, row kernels alone, both at once.
Usage: python tools/overlap_streams.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pixart_sigma_amd import ops
B, N, D = 16, 4096, 1152
R = B * N
dev = "cuda"
a = torch.randn(R, D, device=dev).to(ops.BF16)
w = (torch.randn(4608, D, device=dev) * D ** -0.5).to(ops.BF16)
out = torch.empty(R, 4608, dtype=ops.BF16, device=dev)
dy = torch.randn(R, 4608, device=dev).to(ops.BF16)
dw = torch.zeros(4608, D, device=dev)
x = torch.randn(R, D, device=dev)
mod = torch.randn(B, 6, D, device=dev) * 0.3
u = torch.randn(R, D, device=dev).to(ops.BF16)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def gemm_nt():
    ops.gemm(a, w, ops.NT, out=out)


def gemm_tn():
    ops.gemm(dy, a, ops.TN, out_f32=dw, accumulate=True, split_k=0)


def rows(n=3):
    for _ in range(n):
        ops.ln_mod_fwd(x, mod[:, 0], mod[:, 1], 6 * D, u=u, gate=mod[:, 2], gate_stride=6 * D, rows_per_batch=N, want_stats=True)


def wall(fa, fb, iters=10):
    for _ in range(2):
        if fa: 
            with torch.cuda.stream(s1): fa()
        if fb:
            with torch.cuda.stream(s2): fb()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        if fa:
            with torch.cuda.stream(s1): fa()
        if fb:
            with torch.cuda.stream(s2): fb()
        s1.synchronize(); s2.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


for name, g in (("NT fc1", gemm_nt), ("TN fc1 dW", gemm_tn)):
    ta, tb, tab = wall(g, None), wall(None, rows), wall(g, rows)
    print(f"{name}: GEMM alone {ta:7.1f} us   3 x ln_mod_fwd alone {tb:7.1f} us   both streams {tab:7.1f} us   (sum {ta+tb:7.1f}, max {max(ta,tb):7.1f})")
