"""fc1-shaped NT GEMM with the three forward epilogues (bias only / GELU one output / GELU + GELU' two outputs) at the inference and training row counts.
Usage: python tools/kbench_gelu.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pixart_sigma_amd import ops
from tools.kbench import timed
N, K = 4608, 1152
for M in (16384, 65536):
    a = torch.randn(M, K, device="cuda").to(ops.BF16)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(ops.BF16)
    b = torch.randn(N, device="cuda")
    out, out2 = torch.empty(M, N, dtype=ops.BF16, device="cuda"), torch.empty(M, N, dtype=ops.BF16, device="cuda")
    for name, kw in (("bias", {}), ("gelu (one output)", dict(act=ops.ACT_GELU)), ("gelu + gelu' (two outputs)", dict(act=ops.ACT_GELU_SAVE_GRAD, out2=out2))):
        t = timed(lambda: ops.gemm(a, w, ops.NT, bias=b, out=out, **kw), iters=20, warm=2)
        print(f"M={M:6d} {name:28s}: {t*1e6:7.1f} us  {2.0*M*N*K/t/1e12:7.1f} TF/s")
