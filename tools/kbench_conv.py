"""One implicit 3x3 convolution (zero-padded NHWC grid + segmented-K pxa_gemm) at a VAE layer shape: time and TFLOP/s.
Usage (GPU box): python tools/kbench_conv.py C Cout H W B [iters]      (KBENCH_CONV_PLAIN=1: segment-major K order instead of the
tap-interleaved one the VAE uses - same FLOPs, same bytes, different re-read distance)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pixart_sigma_amd import ops  # noqa: E402

C, Co, H, W, B = (int(v) for v in sys.argv[1:6])
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 20
ip, rp = (H + 2) * (W + 2), W + 2
buf = torch.zeros((B * ip + 2 * (W + 3)) * C, dtype=ops.BF16, device="cuda")
buf.view(-1, C)[W + 3: W + 3 + B * ip].normal_()
w = (torch.randn(Co, 9 * C, device="cuda") * (9 * C) ** -0.5).to(ops.BF16)
bias = torch.zeros(Co, device="cuda")
a = buf.as_strided((B * ip, 9 * C), (C, 1))
out = torch.empty(B * ip, Co, dtype=ops.BF16, device="cuda")
run = lambda: ops.gemm(a, w, ops.NT, bias=bias, out=out, k_seg=3 * C, a_seg_stride=rp * C, k_tap=0 if os.environ.get("KBENCH_CONV_PLAIN") else C)
for _ in range(3):
    run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    run()
e1.record()
e1.synchronize()
ms = e0.elapsed_time(e1) / iters
fl = 2.0 * B * H * W * 9 * C * Co
print(f"conv3x3 C={C} Cout={Co} {H}x{W} B={B}: {ms:.3f} ms {fl / ms / 1e9:.1f} TFLOP/s (unpadded FLOPs); input {B * ip * C * 2 / 1e6:.0f} MB, output {B * ip * Co * 2 / 1e6:.0f} MB, weights {Co * 9 * C * 2 / 1e6:.2f} MB")
