"""Outputs of every epilogue flavour of the persistent GEMM on fixed seeded inputs, saved to a file: two libraries (PXA_LIB_PATH) must produce the same bits
(tests/test_kernels_gpu.py::test_gemm_counted_waits_match_full_waits: the product library against its -DGEMM_WAIT_ALL=1 build).
Shapes are the token GEMMs' (M = 16,384 rows: every workgroup hands over between items; widths 1152 / 3456 / 4608: full, half and - for the implicit
convolution with 128 output channels - paired items).   usage: python tools/gemm_flavour_dump.py OUT.pt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pixart_sigma_amd import ops  # noqa: E402


def main(out_path):
    g = torch.Generator().manual_seed(0)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).cuda()      # noqa: E731
    M, D, F = 16384, 1152, 4608
    x, h = r(M, D).to(ops.BF16), r(M, F).to(ops.BF16)
    w1, w2, wq = r(F, D, sc=D ** -0.5).to(ops.BF16), r(D, F, sc=F ** -0.5).to(ops.BF16), r(3 * D, D, sc=D ** -0.5).to(ops.BF16)
    b1, b2, bq = r(F, sc=0.1), r(D, sc=0.1), r(3 * D, sc=0.1)
    res = {}
    res["nt_qkv_bias"] = ops.gemm(x, wq, ops.NT, bias=bq, descending=True)                                  # EPI 0, full tiles + half items
    res["nt_fc2_bias"] = ops.gemm(h, w2, ops.NT, bias=b2)                                                   # EPI 0, K = 4608
    pre = torch.empty(M, F, dtype=ops.BF16, device="cuda")
    res["nt_fc1_gelu_dual"] = ops.gemm(x, w1, ops.NT, bias=b1, act=ops.ACT_GELU_SAVE_GRAD, out2=pre)        # EPI 1: two outputs
    res["nt_fc1_gelu_dual_2"] = pre
    res["nt_fc1_gelu"] = ops.gemm(x, w1, ops.NT, bias=b1, act=ops.ACT_GELU)                                 # EPI 7
    cs = torch.zeros(ops.COLSUM_SLOTS, F, device="cuda")
    du = r(M, D).to(ops.BF16)
    res["nn_fc2_dx_mul_aux_colsum"] = ops.gemm(du, w2, ops.NN, act=ops.ACT_MUL_AUX, aux=pre, colsum=cs)     # EPI 2: aux loads with counted waits + column sums
    res["nn_fc2_dx_colsum_sum"] = cs.sum(0)       # (atomic order is not fixed: compared to a tolerance by the caller)
    res["nn_fc1_dx"] = ops.gemm(h, w1, ops.NN, descending=True)                                             # EPI 0 NN, half items
    # implicit 3x3 convolutions of the VAE: 128 -> 128 (paired items) and 256 -> 256 (full tiles) with residual + GroupNorm partial sums
    for C, Co in ((128, 128), (256, 256)):
        B, H, W = 2, 62, 62
        ip, rp = ((H + 2) * (W + 2) + 255) // 256 * 256, W + 2
        buf = torch.zeros((B * ip + 2 * (W + 3)) * C, dtype=ops.BF16, device="cuda")
        buf.view(-1, C)[W + 3: W + 3 + B * ip].copy_(r(B * ip, C).to(ops.BF16))
        wc = r(Co, 9 * C, sc=(9 * C) ** -0.5).to(ops.BF16)
        resid = r(B * ip, Co).to(ops.BF16)
        part = torch.zeros(ops.COLSUM_SLOTS, B, Co // 4, 2, device="cuda")
        a = buf.as_strided((B * ip, 9 * C), (C, 1))
        res[f"conv_{C}_{Co}_add_aux_gn"] = ops.gemm(a, wc, ops.NT, bias=r(Co, sc=0.1), k_seg=3 * C, a_seg_stride=rp * C, k_tap=C, act=ops.ACT_ADD_AUX, aux=resid,
                                                    gn_part=part, gn_geom=(ip, rp, H, W))
        res[f"conv_{C}_{Co}_gn_part_sum"] = part.sum(0)
        res[f"conv_{C}_{Co}_plain"] = ops.gemm(a, wc, ops.NT, k_seg=3 * C, a_seg_stride=rp * C, k_tap=C)
    torch.cuda.synchronize()
    torch.save({k: v.cpu() for k, v in res.items()}, out_path)
    print("saved", len(res), "tensors to", out_path)


if __name__ == "__main__":
    main(sys.argv[1])
