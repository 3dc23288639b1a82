#!/usr/bin/env python
"""Training entry point with the reference's launch shape (reference train_scripts/train.py:244-296:
`python -m torch.distributed.run --nproc_per_node=N train_scripts/train.py <config.py> --work-dir ... [--load-from ...]
[--resume-from ...] [--debug]`), driving the MI355X denoiser + fused AdamW + overlapped RCCL all-reduce.

Data: precomputed features in the reference's layout (tools/extract_features.py: `<name>.npy` = cat[mean,std] latent,
`<name>.npz` = caption_feature + attention_mask) listed in `config.data_root`, or `--synthetic`.  With `load_vae_feat = False`
the batches are images and the latents come from the HIP VAE on the fly, as in reference train.py:144-153 (`vae_pretrained` = a
diffusers AutoencoderKL directory; synthetic images and a random-init VAE when it is absent).  T5 on the fly, mmcv
configs with `_base_` inheritance, tensorboard/wandb trackers and validation image logging are outside this repo's scope
(SURVEY.md section 2): the config is a plain Python file whose module-level names are the keys of section 5 of the survey.
"""
import argparse
import glob
import os
import runpy
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixart_sigma_amd import IDDPM, build_model  # noqa: E402
from pixart_sigma_amd.dp import FusedAdamW, FusedCAME  # noqa: E402

DEFAULTS = dict(model="PixArtMS_XL_2", image_size=1024, train_batch_size=16, num_epochs=1, model_max_length=300, pred_sigma=True,
                learn_sigma=True, class_dropout_prob=0.1, kv_compress=False, kv_compress_config=None, micro_condition=False,
                grad_checkpointing=False, fp32_attention=False, gc_step=1, scale_factor=0.13025, gradient_clip=0.01,
                optimizer=dict(type="AdamW", lr=2e-5, weight_decay=3e-2, eps=1e-10), train_sampling_steps=1000, snr_loss=False,
                log_interval=20, save_model_steps=1000, seed=43, data_root=None, load_vae_feat=True,
                vae_pretrained="output/pretrained_models/pixart_sigma_sdxlvae_T5_diffusers/vae")


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("config", nargs="?", default=None)
    p.add_argument("--work-dir", "--work_dir", default="output/debug")
    p.add_argument("--resume-from", default=None)
    p.add_argument("--load-from", default=None)
    p.add_argument("--debug", action="store_true")
    p.add_argument("--synthetic", action="store_true")
    p.add_argument("--max-steps", type=int, default=None)
    return p.parse_args()


def batches(cfg, B, lat, L, dev, rank, world, synthetic):
    if synthetic or not cfg["data_root"]:
        g = torch.Generator().manual_seed(cfg["seed"] + rank)
        while not cfg["load_vae_feat"]:                                            # images in, latents from the VAE inside the step
            yield torch.randn(B, 3, lat * 8, lat * 8, generator=g).clamp(-1, 1).to(dev), torch.randn(B, 1, L, 4096, generator=g).to(dev), torch.ones(B, L, dtype=torch.int64)
        while True:
            yield torch.randn(B, 4, lat, lat, generator=g).to(dev), torch.randn(B, 1, L, 4096, generator=g).to(dev), torch.ones(B, L, dtype=torch.int64)
    names = sorted(glob.glob(os.path.join(cfg["data_root"], "*.npz")))[rank::world]
    while True:
        for i in range(0, len(names) - B + 1, B):
            zs, ys, ms = [], [], []
            for n in names[i:i + B]:
                f = np.load(n)
                lat_stats = np.load(n[:-4] + ".npy")                               # cat[mean, std] (InternalData.py:296-301)
                mean, std = np.split(lat_stats.reshape(8, lat, lat), 2)
                zs.append(torch.from_numpy(mean + std * np.random.randn(*mean.shape)).float())
                ys.append(torch.from_numpy(f["caption_feature"]).float().reshape(1, -1, 4096)[:, :L])
                ms.append(torch.from_numpy(f["attention_mask"]).reshape(-1)[:L])
            yield torch.stack(zs).to(dev), torch.stack(ys).to(dev), torch.stack(ms)


def main():
    a = parse_args()
    cfg = dict(DEFAULTS)
    if a.config:
        cfg.update({k: v for k, v in runpy.run_path(a.config).items() if not k.startswith("_")})
    world, rank, local = int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    if a.debug:
        cfg.update(train_batch_size=2, log_interval=1)
    torch.manual_seed(cfg["seed"])
    lat, L = cfg["image_size"] // 8, cfg["model_max_length"]
    model = build_model(cfg["model"], cfg["grad_checkpointing"], cfg["fp32_attention"], gc_step=cfg["gc_step"], input_size=lat,
                        pe_interpolation=cfg["image_size"] / 512, model_max_length=L, micro_condition=cfg["micro_condition"],
                        kv_compress_config=cfg["kv_compress_config"] if cfg["kv_compress"] else None,
                        pred_sigma=cfg["pred_sigma"], learn_sigma=cfg["learn_sigma"], class_dropout_prob=cfg["class_dropout_prob"])
    start_step = 0
    ck = a.resume_from or a.load_from
    if ck:
        sd = torch.load(ck, map_location="cpu")
        model.load_state_dict(sd.get("state_dict", sd), strict=False)
    model = model.to(dev).train()
    model.prepare(dev)
    o = cfg["optimizer"]
    if o.get("type", "AdamW") in ("CAMEWrapper", "CAME"):      # the optimizer of the PixArt-Sigma configs (reference optimizer.py:242-246)
        opt = FusedCAME(model, lr=o["lr"], weight_decay=o.get("weight_decay", 0.0), eps=o.get("eps", (1e-30, 1e-16)),
                        betas=o.get("betas", (0.9, 0.999, 0.9999)), max_grad_norm=cfg["gradient_clip"])
    else:
        opt = FusedAdamW(model, lr=o["lr"], weight_decay=o["weight_decay"], eps=o.get("eps", 1e-8), betas=o.get("betas", (0.9, 0.999)),
                         max_grad_norm=cfg["gradient_clip"])
    if a.resume_from and "optimizer" in sd:
        opt.load_state_dict(sd["optimizer"])
        start_step = int(os.path.basename(a.resume_from).split("_step_")[-1].split(".")[0]) if "_step_" in a.resume_from else sd.get("step", 0)
    diff = IDDPM(str(cfg["train_sampling_steps"]), learn_sigma=cfg["learn_sigma"], pred_sigma=cfg["pred_sigma"], snr=cfg["snr_loss"])
    vae = None
    if not cfg["load_vae_feat"]:                                                   # reference train.py:351-354
        from pixart_sigma_amd.vae import AutoencoderKL
        have = os.path.isdir(str(cfg["vae_pretrained"]))
        vae = (AutoencoderKL.from_pretrained(cfg["vae_pretrained"], torch_dtype=torch.float16) if have else AutoencoderKL(scaling_factor=cfg["scale_factor"])).to(dev)
        cfg["scale_factor"] = vae.config.scaling_factor
    os.makedirs(os.path.join(a.work_dir, "checkpoints"), exist_ok=True)
    it = batches(cfg, cfg["train_batch_size"], lat, L, dev, rank, world, a.synthetic)
    t0, step = time.time(), start_step
    while a.max_steps is None or step < start_step + a.max_steps:
        z, y, mask = next(it)
        if vae is not None:                                                        # reference train.py:147-153
            z = vae.encode(z).latent_dist.sample().float()
        x0 = z * cfg["scale_factor"]
        t = torch.randint(0, cfg["train_sampling_steps"], (z.shape[0],), device=dev).long()
        opt.zero_grad()
        loss = diff.training_losses(model, x0, t, model_kwargs=dict(y=y, mask=mask, data_info=None))["loss"].mean()
        loss.backward()
        opt.step()
        step += 1
        if step % cfg["log_interval"] == 0 and rank == 0:      # host sync only here (the reference syncs every step, train.py:187)
            print(f"step {step} loss {loss.item():.4f} grad_norm {opt.last_norm.item():.4f} {(time.time() - t0) / cfg['log_interval']:.3f} s/step", flush=True)
            t0 = time.time()
        if step % cfg["save_model_steps"] == 0 and rank == 0:
            torch.save({"state_dict": model.state_dict(), "optimizer": opt.state_dict(), "step": step},
                       os.path.join(a.work_dir, "checkpoints", f"epoch_1_step_{step}.pth"))
    if rank == 0:
        print(f"finished at step {step}: loss {loss.item():.4f} grad_norm {opt.last_norm.item():.4f}", flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
