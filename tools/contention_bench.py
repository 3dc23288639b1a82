"""What happens to the persistent GEMM when another kernel holds some CUs (the situation of an RCCL all-reduce overlapping the
backward)?  A long-running 'hog' (an attention-forward launch with 32 workgroups looping over 2M keys) runs on a side stream while
a train of fc1-shaped GEMMs is timed on the main stream.  Run twice: PXA_GEMM_DYNAMIC=1 (dynamic per-XCD item cursors: what the data-parallel runtime switches on) and default (static split)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pixart_sigma_amd import ops
dev = "cuda"
R, D, DFF = 65536, 1152, 4608
x = torch.randn(R, D, device=dev).to(torch.bfloat16)
w = (torch.randn(DFF, D, device=dev) * D ** -0.5).to(torch.bfloat16)
out = torch.empty(R, DFF, dtype=torch.bfloat16, device=dev)
def train(n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        ops.gemm(x, w, ops.NT, out=out)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n
for _ in range(3): train(10)
print(f"alone            : {train(40):.3f} ms per GEMM")
# hog: B=1, H=1, Nq = 32 x 128 queries -> 32 workgroups, each streaming Nk keys
Nq, Nk = int(os.environ.get("HOG_BLOCKS", "32")) * 128, 1 << 21
q = torch.randn(Nq, 72, device=dev).to(torch.bfloat16)
kv = torch.randn(Nk, 144, device=dev).to(torch.bfloat16)
o = torch.empty(Nq, 72, dtype=torch.bfloat16, device=dev)
lse = torch.empty(1, 1, Nq, device=dev)
side = torch.cuda.Stream()
def hog():
    with torch.cuda.stream(side):
        ops.attention_fwd(q, kv[:, :72], kv[:, 72:], o, lse, 1, 1, Nq, Nk, ((0, 72, 72), (0, 144, 72), (0, 144, 72), (0, 72, 72)))
hog(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
with torch.cuda.stream(side):
    e0.record(side); 
hog()
with torch.cuda.stream(side):
    e1.record(side)
torch.cuda.synchronize()
print(f"hog alone        : {e0.elapsed_time(e1):.1f} ms ({Nq // 128} workgroups)")
hog()
t = train(40)
torch.cuda.synchronize()
print(f"beside the hog   : {t:.3f} ms per GEMM  ({'dynamic' if os.environ.get('PXA_GEMM_DYNAMIC') and not os.environ.get('PXA_GEMM_STATIC') else 'static'} item assignment)")
