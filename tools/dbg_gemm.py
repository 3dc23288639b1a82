import sys, os, torch
sys.path.insert(0, os.getcwd())
from pixart_sigma_amd import ops
torch.manual_seed(0)
def rb(*s): return torch.randn(*s, device="cuda").to(torch.bfloat16)
def report(name, out, ref, bm=128, bn=64):
    e = (out.float() - ref).abs()
    M, N = ref.shape
    print(name, "rel", ((out.float()-ref).norm()/ref.norm()).item())
    bad = []
    for i in range(0, M, bm):
        for j in range(0, N, bn):
            v = e[i:i+bm, j:j+bn].max().item() / (ref[i:i+bm, j:j+bn].abs().max().item() + 1e-9)
            if not (v < 0.05): bad.append((i, j, round(v, 3)))
    print("   bad blocks (row, col, err):", bad[:24], "... total", len(bad))
for (M, N, K) in [(1024, 1024, 128), (1024, 1152, 128), (1280, 1152, 128)]:
    a, w = rb(M, K), rb(N, K)
    report(f"NT {M}x{N}x{K}", ops.gemm(a, w, ops.NT), a.float() @ w.float().t())
    wt = rb(K, N)
    report(f"NN {M}x{N}x{K}", ops.gemm(a, wt, ops.NN), a.float() @ wt.float())
    at, bt = rb(K, M), rb(K, N)
    o = torch.zeros(M, N, device="cuda")
    ops.gemm(at, bt, ops.TN, out_f32=o, accumulate=True, split_k=1)
    report(f"TN {M}x{N}x{K}", o, at.float().t() @ bt.float())
