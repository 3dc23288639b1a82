"""kv_linear-shaped GEMMs of the cross-attention branch (M = 16 x 300 text rows): forward NT, dX NN with fp32 accumulate, dW TN - time per split / tile choice.
Usage: python tools/kbench_kvlin.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pixart_sigma_amd import ops
from tools.kbench import timed
M, D = 4800, 1152
dev = "cuda"
dy = torch.randn(M, 2 * D, device=dev).to(ops.BF16)
w = (torch.randn(2 * D, D, device=dev) * D ** -0.5).to(ops.BF16)
ref = dy.float() @ w.float()
for sk in (1, 0, 2):
    out = torch.zeros(M, D, device=dev)
    ops.gemm(dy, w, ops.NN, out_f32=out, accumulate=True, split_k=sk)
    err = ((out - ref).norm() / ref.norm()).item()
    t = timed(lambda: ops.gemm(dy, w, ops.NN, out_f32=out, accumulate=True, split_k=sk), iters=20, warm=2)
    print(f"NN dX fp32-accumulate M={M} N={D} K={2*D} split_k={sk}: {t*1e6:6.1f} us  {2.0*M*D*2*D/t/1e12:6.1f} TF/s  rel err {err:.1e}")
