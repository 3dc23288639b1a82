"""TEST INFRASTRUCTURE ONLY - what the REFERENCE's own mixed-precision path does to the gradients.
Runs the unmodified reference (/root/reference under oracle/ref_stubs.py) on a training golden's inputs twice - plain fp32, and under
torch.autocast(float16) with a 65536 loss scale, the reference's training mode (configs/PixArt_xl2_internal.py:57 mixed_precision='fp16',
train_scripts/train.py:318-326) - and prints, per parameter tensor, rel-L2(fp16-path gradient, fp32 gradient).  The xformers stub keeps softmax and
both attention products in fp32 and rounds only its inputs / outputs to fp16, i.e. it is at least as accurate as any fp16 attention kernel.
    python -m oracle.ref_fp16_noise [golden-case ...]      (build container only: needs /root/reference)
Round 6: `--json tests/golden/ref_fp16_noise.json` also writes the figure of EVERY parameter tensor (and of the loss) per case - the yardstick the GPU tier prints
next to this repo's own errors and asserts against (tests/test_model_gpu.py, __graft_entry__.smoke()).  The case "smoke" is the step smoke() runs."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_stubs  # noqa: E402
from oracle.make_golden import CASES, build_reference  # noqa: E402
from oracle.pixart_oracle import OracleCfg  # noqa: E402
from oracle.weights import make_inputs, make_state_dict  # noqa: E402


SMOKE_CASE = (dict(depth=2, input_size=16, model_max_length=20), dict(B=2, Hl=16, Wl=16, L=20, lens=[20, 9]))      # __graft_entry__._smoke_once


def grads(name, half):
    ckw, ikw = SMOKE_CASE if name == "smoke" else CASES[name]
    cfg = OracleCfg(**ckw)
    sd = make_state_dict(cfg, seed=0)
    inp = make_inputs(seed=1, **ikw)
    m = build_reference(cfg, sd).train()
    from diffusion import IDDPM
    diff = IDDPM(str(1000), learn_sigma=True, pred_sigma=True, snr=False)
    t = inp["t"].clone()
    if t.numel() > 1 and name != "smoke":
        t[0] = 0
    hw = torch.tensor([[inp["x"].shape[-2] * 8.0, inp["x"].shape[-1] * 8.0]] * inp["x"].shape[0])
    kw = dict(y=inp["y"], mask=inp["mask"][:, None, None, :], data_info={"img_hw": hw, "aspect_ratio": torch.ones(inp["x"].shape[0], 1)})
    scale = 65536.0 if half else 1.0
    with torch.autocast("cpu", dtype=torch.float16, enabled=half):
        terms = diff.training_losses(m, inp["x"], t, model_kwargs=kw, noise=inp["noise"])
        loss = terms["loss"].mean()
    (loss.float() * scale).backward()
    return {k: p.grad.float() / scale for k, p in m.named_parameters()}, terms["loss"].detach().float()


def main():
    out = {}
    args = list(sys.argv[1:])
    jpath = None
    if "--json" in args:
        i = args.index("--json")
        jpath = args[i + 1]
        del args[i:i + 2]
    full = json.load(open(jpath)) if jpath and os.path.exists(jpath) else {}
    for name in args or ["train_d2", "train_d2_plain"]:
        t0 = time.time()
        g32, l32 = grads(name, False)
        g16, l16 = grads(name, True)
        rows = []
        for k in g32:
            n = g32[k].norm().item()
            if n < 1e-9:
                continue
            rows.append((((g16[k] - g32[k]).norm() / n).item(), k))
        rows.sort(reverse=True)
        fam = {}
        for e, k in rows:
            f = ".".join(p for p in k.split(".") if not p.isdigit())
            fam[f] = max(fam.get(f, 0.0), e)
        out[name] = {"loss_rel": ((l16 - l32).norm() / l32.norm()).item(), "worst": [(k, e) for e, k in rows[:8]],
                     "by_family": dict(sorted(fam.items(), key=lambda x: -x[1])[:12]), "seconds": time.time() - t0}
        print(name, json.dumps(out[name], indent=1), flush=True)
        full[name] = {"loss_rel": out[name]["loss_rel"], "grad_rel": {k: e for e, k in rows}, "worst": rows[0][0],
                      "what": "rel-L2 of the unmodified reference under torch.autocast(float16) + 65536 loss scale against its own fp32 run, same inputs"}
        if jpath:
            json.dump(full, open(jpath, "w"), indent=0, sort_keys=True)
    return out


if __name__ == "__main__":
    main()
