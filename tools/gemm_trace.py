"""Timeline of one workgroup of the persistent GEMM (diagnostics build -DGEMM_TRACE=1): s_memtime at 6 points of its first 12 items.
PXA_LIB_PATH=pixart_sigma_amd/variants/lib_gtrace.so python tools/gemm_trace.py M N K [NT|NN]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pixart_sigma_amd import lib, ops
M, N, K = (int(x) for x in sys.argv[1:4])
lay = sys.argv[4] if len(sys.argv) > 4 else "NT"
a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
b = (torch.randn(*((N, K) if lay == "NT" else (K, N)), device="cuda") * K ** -0.5).to(torch.bfloat16)
out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
bias = torch.zeros(N, device="cuda")
for _ in range(5):
    ops.gemm(a, b, ops.NT if lay == "NT" else ops.NN, bias=bias, out=out)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 96)()
L = lib.load()
L.pxa_gemm_trace.argtypes = [ctypes.c_void_p]
assert L.pxa_gemm_trace(buf) == 0
rows = [[buf[i * 8 + k] for k in range(6)] for i in range(12)]
print("s_memtime ticks are 100 MHz (10 ns).  per item: wait+barrier | main loop | tail barrier | prefetch issue | epilogue | (next) total")
for i in range(11):
    r, nx = rows[i], rows[i + 1]
    if not r[0] or not nx[0]:
        break
    d = [r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3], r[5] - r[4], nx[0] - r[5], nx[0] - r[0]]
    print(f"item {i}: " + " | ".join(f"{x * 0.01:6.2f} us" for x in d))
