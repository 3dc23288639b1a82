"""Square-GEMM sanity benchmark (compare with the CDNA guide's 4096^3 / 8192^3 reference numbers)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pixart_sigma_amd import ops
from tools.kbench import timed
for n in (4096, 8192):
    a = torch.randn(n, n, device="cuda").to(torch.bfloat16)
    b = (torch.randn(n, n, device="cuda") * n ** -0.5).to(torch.bfloat16)
    out = torch.empty(n, n, dtype=torch.bfloat16, device="cuda")
    for lay, nm in ((ops.NT, "NT"), (ops.NN, "NN")):
        t = timed(lambda: ops.gemm(a, b, lay, out=out), iters=10)
        print(f"{nm} {n}^3: {t*1e3:.3f} ms {2.0*n**3/t/1e12:.1f} TF/s")
    outf = torch.zeros(n, n, device="cuda")
    for tile in ("128", "256"):
        os.environ["PXA_GEMM_TILE"] = tile
    t = timed(lambda: ops.gemm(a, b, ops.TN, out_f32=outf, accumulate=False, split_k=1), iters=10)
    print(f"TN {n}^3 fp32 store: {t*1e3:.3f} ms {2.0*n**3/t/1e12:.1f} TF/s")
    t = timed(lambda: ops.gemm(a, b, ops.TN, out_f32=outf, accumulate=True, split_k=1), iters=10)
    print(f"TN {n}^3 fp32 atomic: {t*1e3:.3f} ms {2.0*n**3/t/1e12:.1f} TF/s")
