"""Optimizer step time on the full XL/2 parameter set (610.9 M parameters, flat store): fused AdamW vs fused CAME (clip included).
Usage (GPU box): python tools/bench_opt.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pixart_sigma_amd import PixArtMS_XL_2  # noqa: E402
from pixart_sigma_amd.dp import FusedAdamW, FusedCAME  # noqa: E402


def main():
    torch.manual_seed(0)
    m = PixArtMS_XL_2(input_size=32, pe_interpolation=0.5, model_max_length=300).cuda().train()
    m.prepare(torch.device("cuda"))
    n = sum(p.numel() for p in m.parameters())
    out = {"parameters": n}
    for name, cls in (("adamw", FusedAdamW), ("came", FusedCAME)):
        opt = cls(m)
        opt.zero_grad()
        m._store.grad.normal_(std=1e-3)
        for _ in range(2):
            opt.step()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            opt.step()
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / 10
        state = sum(v.numel() * v.element_size() for v in opt.state_dict().values() if torch.is_tensor(v))
        out[name] = {"ms_per_step": ms, "state_GB": state / 1e9, "finite": bool(torch.isfinite(m._store.master).all())}
        del opt
    print(json.dumps(out))


if __name__ == "__main__":
    main()
