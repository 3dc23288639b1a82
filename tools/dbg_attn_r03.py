"""Round-3 debug (GPU box): (1) every (batch, head) of the B16 H16 N4096 backward against fp32 attention, per dK/dV kernel mode; (2) the depth-28 1024px
training golden under each mode: worst gradient tensors and where the error enters."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pixart_sigma_amd import ops, lib

what = sys.argv[1] if len(sys.argv) > 1 else "grid"


def rel(a, b):
    return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()


if what == "grid":
    B, H, N = 16, 16, 4096
    C = H * 72
    g = torch.Generator(device="cuda").manual_seed(1)
    qkv = torch.randn(B, N, 3 * C, device="cuda", generator=g).to(ops.BF16)
    do = torch.randn(B, N, C, device="cuda", generator=g).to(ops.BF16)
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    o = torch.empty(B, N, C, dtype=ops.BF16, device="cuda")
    lse, delta = torch.empty(B, H, N, device="cuda"), torch.empty(B, H, N, device="cuda")
    s3 = (N * 3 * C, 3 * C, 72)
    st = (s3, s3, s3, (N * C, C, 72))
    ops.attention_fwd(q, k, v, o, lse, B, H, N, N, st)
    outs = {}
    for mode in ("0", "2", "3", "3", "2"):
        os.environ["PXA_ATTN_DKV"] = mode
        os.environ["PXA_ATTN_DQ"] = "0" if mode == "0" else "1"
        d = torch.full_like(qkv, float("nan"))
        ops.attention_bwd(q, k, v, o, do, lse, delta, d[..., :C], d[..., C:2 * C], d[..., 2 * C:], B, H, N, N, st, (s3, s3, s3))
        torch.cuda.synchronize()
        outs.setdefault(mode, []).append(d)
    print("mode 3 run-to-run identical:", torch.equal(outs["3"][0], outs["3"][1]), "mode 2:", torch.equal(outs["2"][0], outs["2"][1]), "| dQ kernel 0 vs 1 max abs diff:",
          (outs["0"][0][..., :C].float() - outs["2"][0][..., :C].float()).abs().max().item())
    err = {m: torch.zeros(B, H, 3) for m in outs}
    for b in range(B):
        for h in range(H):
            sl = slice(h * 72, (h + 1) * 72)
            qq, kk, vv = (t[b, :, sl].float().clone().requires_grad_(True) for t in (q, k, v))
            p = torch.softmax((qq @ kk.t()) * 72 ** -0.5, dim=-1)
            (p @ vv).backward(do[b, :, sl].float())
            for m, ds in outs.items():
                for i, r in enumerate((qq.grad, kk.grad, vv.grad)):
                    err[m][b, h, i] = rel(ds[0][b, :, i * C + h * 72:i * C + (h + 1) * 72].float(), r)
    for m, e in err.items():
        print(f"mode {m}: max rel-L2 over all 256 heads dq {e[..., 0].max():.3e} dk {e[..., 1].max():.3e} dv {e[..., 2].max():.3e}; worst dk at (b, h) = {divmod(int(e[..., 1].argmax()), H)},"
              f" heads with dk err > 5e-3: {int((e[..., 1] > 5e-3).sum())}, dv: {int((e[..., 2] > 5e-3).sum())}")
else:
    from oracle import pixart_oracle as po
    from oracle.weights import make_inputs, make_state_dict
    from pixart_sigma_amd import IDDPM, build_model
    F16 = lib.OPERAND == "f16"
    g = torch.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "train_xl2_1024_b1.pt"), weights_only=False)
    cfg = po.OracleCfg(**g["cfg"])
    sd = make_state_dict(cfg, seed=g["weights_seed"])
    inp = make_inputs(seed=g["inputs_seed"], **g["inputs"])
    m = build_model("PixArtMS", depth=cfg.depth, hidden_size=1152, num_heads=16, input_size=cfg.input_size, pe_interpolation=cfg.pe_interpolation,
                    model_max_length=cfg.model_max_length, class_dropout_prob=0.0)
    m.load_state_dict(sd)
    m = m.cuda().train()
    diff = IDDPM(str(1000), learn_sigma=True, pred_sigma=True, snr=False)
    kw = dict(y=inp["y"].cuda(), mask=inp["mask"][:, None, None, :].cuda(), data_info=None)
    for mode in sys.argv[2:] or ["0", "2"]:
        os.environ["PXA_ATTN_DKV"] = mode
        if m._store is not None:
            m._store.grad.zero_()
        scale = 65536.0 if F16 else 1.0
        terms = diff.training_losses(m, inp["x"].cuda(), g["t"].cuda(), model_kwargs=kw, noise=inp["noise"].cuda())
        (terms["loss"].mean() * scale).backward()
        m._store.grad.div_(scale)
        rows = []
        for k_, p_ in m.named_parameters():
            ref = g["grads"][k_]
            gr = p_.grad.detach().float().cpu()
            if ref["norm"] < 1e-9:
                continue
            e = rel(gr, ref["full"]) if "full" in ref else rel(gr.flatten()[:: ref["stride"]], ref["sample"])
            rows.append((e, gr.norm().item() / ref["norm"], ref["norm"], k_))
        rows.sort(reverse=True)
        print(f"== dkv mode {mode} ({lib.OPERAND}): loss {terms['loss'].tolist()} ref {g['loss'].tolist()}")
        for r in rows[:6]:
            print("   err %.2e  norm ratio %.3e  ref norm %.3e  %s" % r)
        byblk = {}
        for e, _, _, k_ in rows:
            b_ = k_.split(".")[1] if k_.startswith("blocks.") else k_.split(".")[0]
            byblk[b_] = max(byblk.get(b_, 0), e)
        print("   worst err per block:", " ".join(f"{b_}:{e:.1e}" for b_, e in sorted(byblk.items(), key=lambda x: (not x[0].isdigit(), int(x[0]) if x[0].isdigit() else 0, x[0]))))
