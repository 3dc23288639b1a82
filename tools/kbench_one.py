"""One GEMM shape, few launches (PMC passes): python tools/kbench_one.py M N K [NT|NN] [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pixart_sigma_amd import ops
from tools.kbench import timed
M, N, K = (int(x) for x in sys.argv[1:4])
lay = sys.argv[4] if len(sys.argv) > 4 else "NT"
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 3
a = torch.randn(M, K, device="cuda").to(ops.BF16)
b = (torch.randn(*((N, K) if lay == "NT" else (K, N)), device="cuda") * K ** -0.5).to(ops.BF16)
if os.environ.get("KBENCH_DATA") == "u":   # +-[0.5, 2) like probe/dma_bench.hip (operand statistics change MFMA power, hence clocks)
    a = ((torch.rand_like(a.float()) * 1.5 + 0.5) * torch.sign(torch.randn_like(a.float()))).to(ops.BF16)
    b = ((torch.rand_like(b.float()) * 1.5 + 0.5) * torch.sign(torch.randn_like(b.float()))).to(ops.BF16)
elif os.environ.get("KBENCH_DATA") == "0":
    a.zero_(); b.zero_()
out = torch.empty(M, N, dtype=ops.BF16, device="cuda")
if os.environ.get("KBENCH_LIBREF"):      # vendor library yardstick (hipBLASLt through torch)
    t = timed(lambda: torch.matmul(a, b.t() if lay == "NT" else b, out=out), iters=iters, warm=1)
else:
    t = timed(lambda: ops.gemm(a, b, ops.NT if lay == "NT" else ops.NN, out=out), iters=iters, warm=1)
print(f"{lay} {M}x{N}x{K}: {t*1e3:.3f} ms {2.0*M*N*K/t/1e12:.1f} TF/s")
