"""HBM-bound row kernels at the headline shape (R = 65,536 rows x 1152): time and effective TB/s.  Usage: python tools/kbench_elem.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pixart_sigma_amd import ops
from tools.kbench import timed
B, N, D = 16, 4096, 1152
R = B * N
x, dxin = torch.randn(R, D, device="cuda"), torch.randn(R, D, device="cuda")
dy = torch.randn(R, D, device="cuda").to(ops.BF16)
mod = torch.randn(B, 6, D, device="cuda") * 0.3
st = ops.ln_mod_fwd(x, mod[:, 0], mod[:, 1], 6 * D, rows_per_batch=N, want_stats=True)
dmod = torch.zeros(B, 6, D, device="cuda")
dx = torch.empty_like(x)
dxb = torch.empty(R, D, dtype=ops.BF16, device="cuda")
t = timed(lambda: ops.ln_mod_bwd(dy, x, st["mean"], st["rstd"], mod[:, 1], 6 * D, dxin, dx, dmod[:, 0], dmod[:, 1], 6 * D, N))
print(f"ln_mod_bwd (dx_in)          : {t*1e6:7.1f} us  {(2+4+4+4)*R*D/t/1e12:5.2f} TB/s")
t = timed(lambda: ops.ln_mod_bwd(dy, x, st["mean"], st["rstd"], mod[:, 1], 6 * D, dxin, dx, dmod[:, 0], dmod[:, 1], 6 * D, N, dx_bf16=dxb))
print(f"ln_mod_bwd (dx_in, dx_bf16) : {t*1e6:7.1f} us  {(2+4+4+4+2)*R*D/t/1e12:5.2f} TB/s")
u = torch.randn(R, D, device="cuda").to(ops.BF16)
du = torch.empty(R, D, dtype=ops.BF16, device="cuda")
part = torch.zeros(ops.COLSUM_SLOTS, D, device="cuda")
t = timed(lambda: ops.gate_bwd(dx, u=u, gate=mod[:, 5], mod_stride=6 * D, du=du, dgate=dmod[:, 5], dmod_stride=6 * D, rows_per_batch=N, dbias=part))
print(f"gate_bwd (mlp form)         : {t*1e6:7.1f} us  {(4+2+2)*R*D/t/1e12:5.2f} TB/s")
t = timed(lambda: ops.gate_bwd(dx, add=dy, u=u, gate=mod[:, 2], mod_stride=6 * D, dx_out=dx, du=du, dgate=dmod[:, 2], dmod_stride=6 * D, rows_per_batch=N, dbias=part))
print(f"gate_bwd (attn form)        : {t*1e6:7.1f} us  {(4+2+2+4+2)*R*D/t/1e12:5.2f} TB/s")
out = torch.zeros(D, device="cuda")
t = timed(lambda: ops.colsum(du, out))
print(f"colsum (R x 1152 bf16)      : {t*1e6:7.1f} us  {2*R*D/t/1e12:5.2f} TB/s")
q3 = torch.randn(R, 3 * D, device="cuda").to(ops.BF16)
out3 = torch.zeros(3 * D, device="cuda")
t = timed(lambda: ops.colsum(q3, out3))
print(f"colsum (R x 3456 bf16)      : {t*1e6:7.1f} us  {2*R*3*D/t/1e12:5.2f} TB/s")
t = timed(lambda: ops.ln_mod_fwd(x, mod[:, 0], mod[:, 1], 6 * D, u=u, gate=mod[:, 2], gate_stride=6 * D, rows_per_batch=N, want_stats=True))
print(f"ln_mod_fwd (residual + LN)  : {t*1e6:7.1f} us  {(4+2+4+2)*R*D/t/1e12:5.2f} TB/s")
