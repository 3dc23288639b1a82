"""Which fp16 rounding puts the cross-attention query gradient at 1.09e-3?  Test infrastructure (imports oracle/): the fp32 oracle replays
the depth-2 training golden on the CPU with ONE tensor of the cross-attention branch rounded to fp16 at a time - in the backward exactly where the HIP path (and
the reference's own fp16 autocast path) rounds it - and prints the rel-L2 error this alone causes in d(cross_attn.q_linear.weight / .bias) against the
all-fp32 run.  delta = rowsum(dO * O) from the ROUNDED O is the hypothesis DESIGN section 2 named; the candidates beside it: dO (the upstream gradient as
stored), q / k / v as stored, P and dS as fed to the second products, dq as stored, x1b (the 16-bit copy of the residual stream that q_linear's dW multiplies).
Usage: python tools/rounding_budget_cross_q.py [golden=train_d2] [loss_scale=65536]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import pixart_oracle as po
from oracle.weights import make_inputs, make_state_dict

gname = sys.argv[1] if len(sys.argv) > 1 else "train_d2"
LOSS_SCALE = float(sys.argv[2]) if len(sys.argv) > 2 else 65536.0
g = torch.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", gname + ".pt"), weights_only=False)
r16 = lambda t: t.to(torch.float16).to(torch.float32)
ROUND = set()          # names of the tensors rounded in this run


class CrossAttn(torch.autograd.Function):
    """softmax(q k^T / sqrt(d)) v for one sample, (H, M, d) x (H, L, d), with the backward written out as the kernels compute it."""
    @staticmethod
    def forward(ctx, q, k, v):
        if "q" in ROUND: q = r16(q)
        if "k" in ROUND: k = r16(k)
        if "v" in ROUND: v = r16(v)
        s = (q @ k.transpose(-1, -2)) * q.shape[-1] ** -0.5
        p = torch.softmax(s, dim=-1)
        o = (r16(p) if "P" in ROUND else p) @ v
        ctx.save_for_backward(q, k, v, p, o)
        return r16(o) if "O_out" in ROUND else o

    @staticmethod
    def backward(ctx, do):
        q, k, v, p, o = ctx.saved_tensors
        if "dO" in ROUND: do = r16(do)
        delta = (do * (r16(o) if "O_in_delta" in ROUND else o)).sum(-1, keepdim=True)
        dp = do @ v.transpose(-1, -2)
        ds = p * (dp - delta)
        pv = r16(p) if "P" in ROUND else p
        dsv = r16(ds) if "dS" in ROUND else ds
        sc = q.shape[-1] ** -0.5
        dq = (dsv @ k) * sc
        dk = (dsv.transpose(-1, -2) @ q) * sc
        dv = pv.transpose(-1, -2) @ do
        if "dq" in ROUND: dq = r16(dq)
        return dq, dk, dv


def cross_attention(sd, pfx, x, y_packed, y_lens, cfg, rp=False):
    B, N, C = x.shape
    nh, dh = cfg.num_heads, C // cfg.num_heads
    xin = x + (r16(x) - x).detach() if "x1b" in ROUND else x          # the 16-bit residual copy q_linear reads (straight-through for the gradient)
    q = po._linear(xin, sd, pfx + ".q_linear", rp).view(B, N, nh, dh)
    kv = po._linear(y_packed, sd, pfx + ".kv_linear", rp).view(-1, 2, nh, dh)
    outs, s = [], 0
    for b in range(B):
        L = int(y_lens[b])
        o = CrossAttn.apply(q[b].permute(1, 0, 2), kv[s:s + L, 0].permute(1, 0, 2), kv[s:s + L, 1].permute(1, 0, 2))
        outs.append(o.permute(1, 0, 2)[None])
        s += L
    o = torch.cat(outs, 0).reshape(B, N, C)
    return po._linear(o, sd, pfx + ".proj", rp)


po.cross_attention = cross_attention


def grads():
    cfg = po.OracleCfg(**g["cfg"])
    sd = make_state_dict(cfg, seed=g["weights_seed"])
    inp = make_inputs(seed=g["inputs_seed"], **g["inputs"])
    mask = inp["mask"] if g["inputs"].get("lens") is not None else None
    sd = {k: (v.clone().requires_grad_(True) if k != "y_embedder.y_embedding" else v) for k, v in sd.items()}
    diff = po.GaussianDiffusionOracle()
    terms = diff.training_losses(lambda xt, t: po.forward(sd, cfg, xt, t, inp["y"], mask, data_info=g.get("data_info")), inp["x"], g["t"], inp["noise"])
    (terms["loss"].mean() * LOSS_SCALE).backward()      # the fp16 path's loss scale: without it the small gradients sit in fp16's subnormal range
    return {k: v.grad.clone() for k, v in sd.items() if "cross_attn.q_linear" in k or "cross_attn.kv_linear" in k}


rel = lambda a, b: float((a - b).norm() / b.norm())
torch.manual_seed(0)
ROUND.clear()
ref = grads()
for k in ref:                    # the custom backward is the analytic one: it must reproduce the golden's autograd gradients
    gg = g["grads"][k]
    print(f"fp32 custom backward vs reference golden  {k:45s} {rel(ref[k] / LOSS_SCALE, gg):.2e}" if torch.is_tensor(gg) and gg.shape == ref[k].shape else f"(sampled in the golden) {k}")
print(f"\n{gname}, loss scale {LOSS_SCALE:g}: rel-L2 error of the gradient caused by ONE fp16 rounding (every block's cross-attention), worst over the blocks")
print(f"{'rounded tensor':34s} {'q_linear.weight':>16s} {'q_linear.bias':>14s} {'kv_linear.weight':>17s}")
for name, what in (("O_in_delta", "O inside delta = rowsum(dO O)"), ("dO", "dO (upstream gradient)"), ("q", "q"), ("k", "k"), ("v", "v"), ("P", "P (second products)"),
                   ("dS", "dS (second products)"), ("dq", "dq as stored"), ("O_out", "O as handed to proj"), ("x1b", "x1b (16-bit residual copy)"),
                   ("all", "all of the above")):
    ROUND.clear()
    ROUND.update({"O_in_delta", "dO", "q", "k", "v", "P", "dS", "dq", "O_out", "x1b"} if name == "all" else {name})
    gr = grads()
    w = max(rel(gr[k], ref[k]) for k in ref if k.endswith("q_linear.weight"))
    b = max(rel(gr[k], ref[k]) for k in ref if k.endswith("q_linear.bias"))
    kvw = max(rel(gr[k], ref[k]) for k in ref if k.endswith("kv_linear.weight"))
    print(f"{what:34s} {w:16.2e} {b:14.2e} {kvw:17.2e}")
