"""Kernel time per family from a step profile (tools/profile_round.sh -> profiles/*_step_kernel_stats.csv): the table of DESIGN.md section 5.
Usage: python tools/family_times.py profiles/r4final_step_kernel_stats.csv [more.csv ...]"""
import csv
import sys

FAMILIES = [
    ("attention dK/dV (self + cross)", lambda k: "attn_bwd_dkv" in k),
    ("attention dQ (self + cross)", lambda k: "attn_bwd_dq" in k),
    ("attention forward (self + cross)", lambda k: "attn_fwd" in k),
    ("GEMM NT", lambda k: ("gemm_pers_kernel<0" in k or "gemm_glds_kernel<0" in k or "gemm_nt4" in k or "gemm_kernel<0" in k)),
    ("GEMM NN", lambda k: ("gemm_pers_kernel<1" in k or "gemm_glds_kernel<1" in k or "gemm_kernel<1" in k)),
    ("GEMM TN", lambda k: ("gemm_pers_kernel<2" in k or "gemm_glds_kernel<2" in k or "gemm_kernel<2" in k)),
    ("row passes + column sums + split-K", lambda k: any(t in k for t in ("ln_mod", "gate_bwd", "colsum", "splitk_reduce"))),
]


def main(path):
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["total_ms_per_step"]) for r in rows)
    used = 0.0
    print(f"{path}: {tot:.1f} ms of kernel time per step")
    for name, pred in FAMILIES:
        t = sum(float(r["total_ms_per_step"]) for r in rows if pred(r["kernel"]))
        used += t
        print(f"  {name:38s} {t:7.1f} ms  {100 * t / tot:5.1f} %")
    print(f"  {'everything else':38s} {tot - used:7.1f} ms  {100 * (tot - used) / tot:5.1f} %")


if __name__ == "__main__":
    for p in sys.argv[1:]:
        main(p)
