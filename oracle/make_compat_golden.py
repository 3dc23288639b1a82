"""TEST INFRASTRUCTURE ONLY - golden vectors for the callers' side of the path, produced by running the
UNMODIFIED reference under oracle/ref_stubs.py in the build container:   python -m oracle.make_compat_golden
Writes tests/golden/compat.pt with
  converter : the state dict tools/convert_pixart_to_diffusers.py:23-155 builds from a miniature depth-28 checkpoint (hidden size 6: the
              script only renames / chunks tensors), captured at its Transformer2DModel.load_state_dict call, alpha + micro_condition
  dataset   : InternalDataMSSigma.getdata() on a three-image feature directory written by write_feature_dir() below
  sampler   : one epoch of AspectRatioBatchSampler over 103 fake image sizes
"""
import argparse
import json
import os
import random
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_stubs  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden", "compat.pt")
RATIOS = {"0.5": [704.0, 1408.0], "1.0": [1024.0, 1024.0], "2.0": [1408.0, 704.0]}      # a small bucket table (reference: ASPECT_RATIO_1024)


def mini_state_dict(depth=28, D=6, micro=True, seed=0):
    """The reference's key set (tools/convert_pixart_to_diffusers.py:29-155 pops every one of them) with tiny tensors."""
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    sd = {"x_embedder.proj.weight": r(D, 4, 2, 2), "x_embedder.proj.bias": r(D), "y_embedder.y_proj.fc1.weight": r(D, 8), "y_embedder.y_proj.fc1.bias": r(D),
          "y_embedder.y_proj.fc2.weight": r(D, D), "y_embedder.y_proj.fc2.bias": r(D), "t_embedder.mlp.0.weight": r(D, 4), "t_embedder.mlp.0.bias": r(D),
          "t_embedder.mlp.2.weight": r(D, D), "t_embedder.mlp.2.bias": r(D), "t_block.1.weight": r(6 * D, D), "t_block.1.bias": r(6 * D),
          "final_layer.linear.weight": r(32, D), "final_layer.linear.bias": r(32), "final_layer.scale_shift_table": r(2, D),
          "y_embedder.y_embedding": r(3, 8), "pos_embed": r(1, 4, D)}
    if micro:
        for e in ("csize_embedder", "ar_embedder"):
            sd.update({f"{e}.mlp.0.weight": r(2, 4), f"{e}.mlp.0.bias": r(2), f"{e}.mlp.2.weight": r(2, 2), f"{e}.mlp.2.bias": r(2)})
    for i in range(depth):
        b = f"blocks.{i}."
        sd.update({b + "scale_shift_table": r(6, D), b + "attn.qkv.weight": r(3 * D, D), b + "attn.qkv.bias": r(3 * D), b + "attn.proj.weight": r(D, D),
                   b + "attn.proj.bias": r(D), b + "cross_attn.q_linear.weight": r(D, D), b + "cross_attn.q_linear.bias": r(D),
                   b + "cross_attn.kv_linear.weight": r(2 * D, D), b + "cross_attn.kv_linear.bias": r(2 * D), b + "cross_attn.proj.weight": r(D, D),
                   b + "cross_attn.proj.bias": r(D), b + "mlp.fc1.weight": r(4 * D, D), b + "mlp.fc1.bias": r(4 * D), b + "mlp.fc2.weight": r(D, 4 * D),
                   b + "mlp.fc2.bias": r(D)})
    return sd


def gen_converter(tmp):
    ref_stubs.install()
    import diffusers
    captured = {}

    class Stop(Exception):
        pass

    class T2D:
        def __init__(self, **kw):
            captured["ctor"] = kw

        def load_state_dict(self, sd, strict=True):
            captured["sd"] = {k: v.clone() for k, v in sd.items()}
            raise Stop()
    diffusers.Transformer2DModel = T2D
    sys.modules.setdefault("transformers", types.ModuleType("transformers"))
    tr = sys.modules["transformers"]
    for n in ("T5EncoderModel", "T5Tokenizer"):
        if not hasattr(tr, n):
            setattr(tr, n, type(n, (), {}))
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_converter", os.path.join(ref_stubs.REFERENCE_ROOT, "tools", "convert_pixart_to_diffusers.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    ck = os.path.join(tmp, "mini.pth")
    torch.save({"state_dict": mini_state_dict()}, ck)
    args = argparse.Namespace(orig_ckpt_path=ck, version="alpha", image_size=1024, micro_condition=True, qk_norm=False, kv_compress=False,
                              only_transformer=True, dump_path=tmp)
    try:
        mod.main(args)
    except Stop:
        pass
    return {"converted": captured["sd"], "ctor": {k: v for k, v in captured["ctor"].items()}, "mini_args": dict(depth=28, D=6, micro=True, seed=0)}


def write_feature_dir(root, table, res=1024, seed=0):
    """Three images in the reference's feature layout; caption lengths 7 / 300 / 120, one image without an attention mask."""
    g = torch.Generator().manual_seed(seed)
    meta = [{"path": "a/img0.png", "height": 700, "width": 1400, "ratio": 0.5, "prompt": "p0", "sharegpt4v": "s0"},
            {"path": "a/img1.jpg", "height": 1024, "width": 1000, "ratio": 1.024, "prompt": "p1"},
            {"path": "b/img2.webp", "height": 3000, "width": 600, "ratio": 5.0, "prompt": "dropped: ratio > 4.5"},
            {"path": "b/img3.png", "height": 1500, "width": 700, "ratio": 2.14, "prompt": "p3"}]
    os.makedirs(os.path.join(root, "caption_features_new"), exist_ok=True)
    os.makedirs(os.path.join(root, f"img_sdxl_vae_features_{res}resolution_ms_new"), exist_ok=True)
    with open(os.path.join(root, "data_info.json"), "w") as f:
        json.dump(meta, f)
    for it, Lc in zip(meta, (7, 300, 11, 120)):
        n = "_".join(it["path"].rsplit("/", 1)).rsplit(".", 1)[0]
        key = min(table, key=lambda k: abs(float(k) - it["height"] / it["width"]))
        h, w = int(table[key][0]) // 8, int(table[key][1]) // 8
        np.save(os.path.join(root, f"img_sdxl_vae_features_{res}resolution_ms_new", n + ".npy"), torch.randn(8, h // 8, w // 8, generator=g).numpy())
        feats = {"caption_feature": torch.randn(1, Lc, 64, generator=g).half().numpy()}     # 64 channels: the format, not the width, is under test
        if it["path"] != "a/img1.jpg":
            feats["attention_mask"] = np.ones((1, Lc), dtype=np.int64)
        np.savez(os.path.join(root, "caption_features_new", n + ".npz"), **feats)


def gen_dataset(tmp):
    ref_stubs.install()
    import diffusion.data.datasets.InternalData_ms as ms
    table = dict(ms.ASPECT_RATIO_1024)                # the reference's own 1024px bucket table (diffusion/data/datasets/utils.py)
    root = os.path.join(tmp, "InternData")
    write_feature_dir(root, table)
    ds = ms.InternalDataMSSigma(root, resolution=1024, load_vae_feat=True, load_t5_feat=True, max_length=300, aspect_ratio_type="ASPECT_RATIO_1024")
    items = []
    for i in range(len(ds)):
        torch.manual_seed(100 + i)
        random.seed(0)
        img, txt, mask, info = ds.getdata(i)
        items.append({"img": img.clone(), "txt": txt.clone(), "mask": mask.clone(), "img_hw": info["img_hw"].clone(), "aspect_ratio": info["aspect_ratio"],
                      "mask_type": info["mask_type"]})
    return {"items": items, "len": len(ds), "ori": ds.ori_imgs_nums, "ratio_nums": dict(ds.ratio_nums), "ratios": table}


def gen_sampler():
    ref_stubs.install()
    from diffusion.utils.data_sampler import AspectRatioBatchSampler
    from torch.utils.data import SequentialSampler
    rnd = random.Random(0)
    sizes = [rnd.choice([(512, 1024), (800, 800), (1024, 512), (900, 1000), (1400, 600)]) for _ in range(103)]

    class DS:
        def __len__(self):
            return len(sizes)

        def get_data_info(self, i):
            return {"height": sizes[i][0], "width": sizes[i][1]}
    out = {"sizes": sizes, "runs": []}
    for bs, valid_num, drop_last, nums in ((8, 0, False, {"0.5": 30, "1.0": 50, "2.0": 23}), (5, 25, True, {"0.5": 30, "1.0": 50, "2.0": 23})):
        s = AspectRatioBatchSampler(SequentialSampler(DS()), DS(), bs, RATIOS, drop_last=drop_last, valid_num=valid_num, ratio_nums=nums)
        out["runs"].append({"batch_size": bs, "valid_num": valid_num, "drop_last": drop_last, "ratio_nums": nums, "batches": [list(b) for b in s]})
    return out


def gen_lr():
    """The reference's own schedule / scaling functions (diffusion/utils/lr_scheduler.py:43-88, optimizer.py:18-28) on a torch optimizer."""
    ref_stubs.install()
    mo = sys.modules.get("mmcv")
    for name in ("mmcv.runner", "mmcv.utils"):          # optimizer.py imports registries this stub set does not model
        pass
    from diffusion.utils.lr_scheduler import get_cosine_decay_to_constant_with_warmup
    out = {"cosine_decay": []}
    for warm, total, final in ((10, 200, 0.25), (0, 90, 0.5)):
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.SGD([p], lr=2e-5)
        sch = get_cosine_decay_to_constant_with_warmup(opt, num_warmup_steps=warm, num_training_steps=total, final_lr=final)
        lrs = []
        for _ in range(total + 5):
            lrs.append(opt.param_groups[0]["lr"])
            opt.step()
            sch.step()
        out["cosine_decay"].append({"warm": warm, "total": total, "final": final, "lrs": lrs})
    return out


def main():
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        obj = {"converter": gen_converter(tmp), "dataset": gen_dataset(tmp), "sampler": gen_sampler(), "lr": gen_lr()}
    torch.save(obj, GOLDEN)
    print(f"{GOLDEN}: {os.path.getsize(GOLDEN) / 1024:.1f} KiB; converter keys {len(obj['converter']['converted'])}, dataset items {obj['dataset']['len']}, "
          f"sampler batches {[len(r['batches']) for r in obj['sampler']['runs']]}")


if __name__ == "__main__":
    main()
