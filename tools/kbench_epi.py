"""The two heavy-epilogue token GEMMs of the MLP beside their plain twins, at the training row count, with ROTATING operand / output
sets so that no launch finds its operands in L2 / MALL (the step's cache state: profiles/r4_32).  fp16 or bf16 by PXA_OPERAND_DTYPE; ablation builds by
PXA_LIB_PATH (tools/build_variant.py ... -DGEMM_ABL=...).
Usage: python tools/kbench_epi.py [sets=4] [iters=24]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pixart_sigma_amd import ops
from tools.kbench import timed
M, D, DFF = 65536, 1152, 4608
SETS = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ITERS = int(sys.argv[2]) if len(sys.argv) > 2 else 24
dev = "cuda"
rb = lambda *s: torch.randn(*s, device=dev).to(ops.BF16)
w1 = (torch.randn(DFF, D, device=dev) * D ** -0.5).to(ops.BF16)          # fc1.weight (4608, 1152)
w2 = (torch.randn(D, DFF, device=dev) * DFF ** -0.5).to(ops.BF16)        # fc2.weight (1152, 4608)
b1 = torch.randn(DFF, device=dev)
xs = [rb(M, D) for _ in range(SETS)]                                      # xn2 / du
hs = [torch.empty(M, DFF, dtype=ops.BF16, device=dev) for _ in range(SETS)]
gs = [torch.rand(M, DFF, device=dev).to(ops.BF16) for _ in range(SETS)]   # saved GELU'
part = torch.zeros(ops.COLSUM_SLOTS, DFF, device=dev)
fl = 2.0 * M * DFF * D
i = [0]


def rot(fn):
    def f():
        i[0] = (i[0] + 1) % SETS
        fn(i[0])
    return f


print("lib:", os.environ.get("PXA_LIB_PATH", "default"), "operand:", os.environ.get("PXA_OPERAND_DTYPE", "bf16"), "sets:", SETS)
for name, fn in (
        ("fc1 NT bias (plain twin)", lambda k: ops.gemm(xs[k], w1, ops.NT, bias=b1, out=hs[k])),
        ("fc1 NT bias+GELU, one output", lambda k: ops.gemm(xs[k], w1, ops.NT, bias=b1, act=ops.ACT_GELU, out=hs[k])),
        ("fc1 NT bias+GELU+GELU' (two outputs)", lambda k: ops.gemm(xs[k], w1, ops.NT, bias=b1, act=ops.ACT_GELU_SAVE_GRAD, out=hs[k], out2=gs[(k + 1) % SETS])),
        ("fc2 dX NN (plain twin)", lambda k: ops.gemm(xs[k], w2, ops.NN, out=hs[k])),
        ("fc2 dX NN x aux + colsum", lambda k: ops.gemm(xs[k], w2, ops.NN, act=ops.ACT_MUL_AUX, aux=gs[k], colsum=part, out=hs[k]))):
    t = timed(rot(fn), iters=ITERS, warm=4)
    print(f"{name:40s}: {t*1e6:7.1f} us  {fl/t/1e12:7.1f} TF/s")
