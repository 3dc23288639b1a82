"""Split-K sweep of the weight-gradient (TN) GEMMs at the headline shapes: python tools/kbench_tn.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pixart_sigma_amd import ops
from tools.kbench import timed, rb
R = 65536
for name, n_out, k in (("qkv", 3456, 1152), ("proj", 1152, 1152), ("fc1", 4608, 1152), ("fc2", 1152, 4608)):
    dy, x = rb(R, n_out), rb(R, k)
    dw = torch.zeros(n_out, k, device="cuda")
    fl = 2.0 * R * n_out * k
    res = []
    for sk in (0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 11, 12, 14, 16):
        t = timed(lambda: ops.gemm(dy, x, ops.TN, out_f32=dw, accumulate=True, split_k=sk), iters=30, warm=3)
        res.append((sk, t))
    tiles = ((n_out + 255) // 256) * ((k + 255) // 256)
    print(f"TN {name:4s} M={n_out} N={k} tiles256={tiles}: " + "  ".join(f"sk{sk}:{t*1e3:.3f}ms/{fl/t/1e12:.0f}" for sk, t in res))
