"""Forward parity of the fp16-operand build against the fp32 reference goldens (run with PXA_OPERAND_DTYPE=f16; also works for bf16).
Prints one JSON line: {"operand": ..., "cases": {name: rel_l2}}.  Used by tests/test_f16_parity_gpu.py in a subprocess, because the
operand type is a per-process choice (one library per type)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oracle import pixart_oracle as po
from oracle.weights import make_inputs, make_state_dict
from pixart_sigma_amd import DPMS, build_model, lib

def rel_l2(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm()).item()

out = {"operand": lib.OPERAND, "cases": {}}
for name in sys.argv[1:] or ["fwd_d2_sq", "fwd_d2_kvconv", "fwd_d2_qknorm", "cfg1_xl2_256"]:
    g = torch.load(os.path.join(ROOT, "tests", "golden", name + ".pt"), weights_only=False)
    cfg = po.OracleCfg(**g["cfg"])
    sd = make_state_dict(cfg, seed=g["weights_seed"])
    inp = make_inputs(seed=g["inputs_seed"], **g["inputs"])
    kvc = None
    if cfg.kv_sampling is not None:
        kvc = {"sampling": cfg.kv_sampling, "scale_factor": cfg.kv_scale_factor, "kv_compress_layer": list(cfg.kv_layers)}
    m = build_model("PixArtMS", depth=cfg.depth, hidden_size=1152, num_heads=16, input_size=cfg.input_size, pe_interpolation=cfg.pe_interpolation,
                    model_max_length=cfg.model_max_length, class_dropout_prob=0.0, kv_compress_config=kvc, qk_norm=cfg.qk_norm)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    mask = inp["mask"] if g["inputs"].get("lens") is not None else None
    with torch.no_grad():
        y = m(inp["x"].cuda(), inp["t"].cuda(), inp["y"].cuda(), mask=mask).cpu()
        key = "y" if "y" in g else "fwd"
        out["cases"][name] = rel_l2(y, g[key])
        if "sample" in g:          # BASELINE config 1: 2-step DPM-Solver++ with CFG 4.5
            gen = torch.Generator().manual_seed(g["null_seed"])
            null_y = torch.randn(1, 1, g["inputs"]["L"], 4096, generator=gen).repeat(inp["x"].shape[0], 1, 1, 1).cuda()
            s = DPMS(m.forward_with_dpmsolver, condition=inp["y"].cuda(), uncondition=null_y, cfg_scale=4.5,
                     model_kwargs=dict(data_info=None, mask=mask)).sample(inp["x"].cuda(), steps=2, order=2, skip_type="time_uniform", method="multistep")
            out["cases"][name + ":sample"] = rel_l2(s.cpu(), g["sample"])
    del m
print(json.dumps(out))
