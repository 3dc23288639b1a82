"""Self-attention backward at the headline shape (B16 H16 N4096 d72), kernels timed separately: dK/dV kernel alone (dq = NULL), dQ kernel
alone (dk = dv = NULL), and the forward.  Used for A/B builds: PXA_LIB_PATH=pixart_sigma_amd/variants/lib_<name>.so python tools/kbench_attn_bwd.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pixart_sigma_amd import ops

D, H = 1152, 16
B, N = int(os.environ.get("KB_B", 16)), int(os.environ.get("KB_N", 4096))
R = B * N


def timed(fn, iters=40, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


qkv = (torch.randn(R, 3 * D, device="cuda") * float(os.environ.get("KB_SCALE", 1.0))).to(ops.BF16)
a = torch.empty(R, D, dtype=ops.BF16, device="cuda")
lse = torch.empty(B, H, N, device="cuda")
s3 = (N * 3 * D, 3 * D, 72)
st = (s3, s3, s3, (N * D, D, 72))
q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
tf = timed(lambda: ops.attention_fwd(q, k, v, a, lse, B, H, N, N, st))
da, dqkv, delta = (torch.randn(R, D, device="cuda") * float(os.environ.get("KB_SCALE", 1.0))).to(ops.BF16), torch.empty_like(qkv), torch.empty(B, H, N, device="cuda")
dq, dk, dv = dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:]
t_dkv = timed(lambda: ops.attention_bwd(q, k, v, a, da, lse, delta, None, dk, dv, B, H, N, N, st, (s3, s3, s3)))
t_dq = timed(lambda: ops.attention_bwd(q, k, v, a, da, lse, delta, dq, None, None, B, H, N, N, st, (s3, s3, s3)))
t_all = timed(lambda: ops.attention_bwd(q, k, v, a, da, lse, delta, dq, dk, dv, B, H, N, N, st, (s3, s3, s3)))
unit = 2.0 * B * H * N * N * 72 / 1e9      # one N^2 d product, GFLOP
print(f"{os.path.basename(os.environ.get('PXA_LIB_PATH', 'default')):24s} fwd {tf:6.3f} ms ({2 * unit / tf:6.0f} TF/s)  dkv+delta {t_dkv:6.3f} ms ({4 * unit / t_dkv:6.0f})  "
      f"dq+delta {t_dq:6.3f} ms ({3 * unit / t_dq:6.0f})  bwd {t_all:6.3f} ms ({5 * unit / t_all:6.0f} TF/s algorithmic)")
