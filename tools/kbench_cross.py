"""Cross-attention kernels (N_q = 4096 image tokens against L text tokens per sample, B16 H16 d72) as a function of L: how much of their time is per-key-tile work
and how much is per-workgroup overhead (query loads, pads, first DMA, stores).  Usage: python tools/kbench_cross.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pixart_sigma_amd import ops
from tools.kbench import timed
B, H, N, D = 16, 16, 4096, 1152
R = B * N
dev = "cuda"
rb = lambda *s: torch.randn(*s, device=dev).to(ops.BF16)
q, a, da = rb(R, D), torch.empty(R, D, dtype=ops.BF16, device=dev), rb(R, D)
lse, delta = torch.empty(B, H, N, device=dev), torch.empty(B, H, N, device=dev)
dq = torch.empty_like(q)
for L in (64, 128, 192, 256, 300, 320):
    kv = rb(B * L, 2 * D)
    dkv = torch.empty_like(kv)
    ks = torch.tensor([i * L for i in range(B)], dtype=torch.int32, device=dev)
    kl = torch.tensor([L] * B, dtype=torch.int32, device=dev)
    sc = ((N * D, D, 72), (0, 2 * D, 72), (0, 2 * D, 72), (N * D, D, 72))
    tf = timed(lambda: ops.attention_fwd(q, kv[:, :D], kv[:, D:], a, lse, B, H, N, L, sc, kv_start=ks, kv_len=kl, max_kv_len=L))
    tb = timed(lambda: ops.attention_bwd(q, kv[:, :D], kv[:, D:], a, da, lse, delta, dq, dkv[:, :D], dkv[:, D:], B, H, N, L, sc,
                                         ((N * D, D, 72), (0, 2 * D, 72), (0, 2 * D, 72)), kv_start=ks, kv_len=kl, max_kv_len=L))
    print(f"L={L:4d}: fwd {tf*1e6:7.1f} us   bwd (delta + dQ + dK/dV) {tb*1e6:7.1f} us")
