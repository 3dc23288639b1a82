"""gemm_nt4_kernel (one wave per SIMD, 128 x 128 per wave: csrc/gemm_nt4.hip) against the eight-wave ping-pong kernel of gemm.hip and an fp32 matmul of
the same 16-bit operands; both timed at the training step's NT shapes, the vendor library beside them.
Usage (GPU box): python tools/kbench_nt4.py [check|time|all]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from box_sampler import Sampler
from pixart_sigma_amd import ops

dev, OPD = "cuda", ops.BF16


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm()).item()


def run(mode, a, b, bias, out=None):
    os.environ["PXA_GEMM_NT4"] = mode
    return ops.gemm(a, b, ops.NT, bias=bias, out=out)


def check():
    bad = 0
    g = torch.Generator(device=dev).manual_seed(0)
    tol = 5e-4 if OPD == torch.float16 else 4e-3
    # (M, N, K, bias, strided): full and half-width items, few / many k-units, item counts below / above one round, operands that are column slices
    cases = [(2048, 256, 256, True, False), (2048, 384, 256, True, False), (4096, 1152, 1152, True, False), (8192, 3456, 1152, True, True),
             (65536, 1152, 1152, True, False), (16384, 4608, 1152, False, False), (16384, 1152, 4608, True, False), (2304, 1280, 384, True, True),
             (65536, 256, 128 * 3, True, False)]
    for M, N, K, hb, strided in cases:
        if strided:
            abig = torch.randn(M, K + 64, device=dev, generator=g).to(OPD)
            a = abig[:, 32:32 + K] if False else abig[:, :K]          # row stride K + 64 (16-byte aligned rows)
            obig = torch.full((M, N + 128), float("nan"), dtype=OPD, device=dev)
            outs = [obig[:, :N], torch.full((M, N + 128), float("nan"), dtype=OPD, device=dev)[:, :N]]
        else:
            a = torch.randn(M, K, device=dev, generator=g).to(OPD)
            outs = [torch.full((M, N), float("nan"), dtype=OPD, device=dev) for _ in range(2)]
        b = (torch.randn(N, K, device=dev, generator=g) * K ** -0.5).to(OPD)
        bias = torch.randn(N, device=dev, generator=g) if hb else None
        o4 = run("1", a, b, bias, outs[0])
        o4b = run("1", a, b, bias, torch.empty_like(outs[0]) if not strided else None)
        o8 = run("0", a, b, bias, outs[1])
        ref = a.float() @ b.float().t() + (bias if hb else 0)
        e4, e8, d = rel(o4.float(), ref), rel(o8.float(), ref), rel(o4.float(), o8.float())
        ok = e4 < max(tol, 1.2 * e8) and torch.isfinite(o4.float()).all().item() and torch.equal(o4, o4b)
        bad += not ok
        print(f"M{M} N{N} K{K} bias {hb} strided {strided}: nt4 vs fp32 {e4:.2e} (ping-pong kernel {e8:.2e}), nt4 vs ping-pong {d:.2e}, bit-identical to it {torch.equal(o4, o8)}  {'ok' if ok else 'FAIL'}", flush=True)
    return bad


def timeit():
    shapes = [("qkv", 65536, 3456, 1152), ("proj", 65536, 1152, 1152), ("fc1", 65536, 4608, 1152), ("fc2", 65536, 1152, 4608)]
    if os.environ.get("KB_NT4_SHAPES"):
        shapes = [s for s in shapes if s[0] in os.environ["KB_NT4_SHAPES"].split(",")]
    for name, M, N, K in shapes:
        ROT = int(os.environ.get("KB_NT4_ROTATE", "1"))           # > 1: cycle through ROT operand / output sets, so that no launch finds its own data in L2 / MALL
        sets = [(torch.randn(M, K, device=dev).to(OPD), (torch.randn(N, K, device=dev) * K ** -0.5).to(OPD), torch.empty(M, N, dtype=OPD, device=dev)) for _ in range(ROT)]
        a, b, out = sets[0]
        bias = torch.randn(N, device=dev)
        fl = 2.0 * M * N * K
        it = [0]
        for mode in os.environ.get("KB_NT4_MODES", "0,1,0,1,lib").split(","):
            def nxt():
                it[0] += 1
                return sets[it[0] % ROT]
            if mode == "lib":
                bb = bias.to(OPD)
                def fn():
                    a_, b_, o_ = nxt()
                    torch.addmm(bb, a_, b_.t(), out=o_)
            else:
                os.environ["PXA_GEMM_NT4"] = mode
                def fn():
                    a_, b_, o_ = nxt()
                    ops.gemm(a_, b_, ops.NT, bias=bias, out=o_)
            for _ in range(5):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with Sampler() as box:
                e0.record()
                for _ in range(50):
                    fn()
                e1.record()
                e1.synchronize()
            t = e0.elapsed_time(e1) / 50 * 1e-3
            print(f"[{os.path.basename(os.environ.get('PXA_LIB_PATH', 'default'))} rot {ROT}] NT {name} M{M} N{N} K{K} {'vendor library (addmm)' if mode == 'lib' else 'PXA_GEMM_NT4=' + mode}: {t * 1e3:7.3f} ms {fl / t / 1e12:7.1f} TF/s  {box.summary()}", flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    rc = 0
    if what in ("check", "all"):
        rc = check()
    if what in ("time", "all"):
        timeit()
    sys.exit(1 if rc else 0)
