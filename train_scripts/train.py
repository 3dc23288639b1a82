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

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # RCCL across processes needs dmabuf IPC on this driver; set before the HIP runtime starts

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
def _run_config(path, _seen=()):
    """A reference config file as a dict, with its `_base_` parents (mmcv convention, paths relative to the file: 13 of the reference configs inherit
    ../PixArt_xl2_internal.py) resolved first and overridden by the child."""
    path = os.path.abspath(path)
    if path in _seen:
        raise SystemExit(f"{path}: circular _base_")
    ns = runpy.run_path(path)
    out = {}
    bases = ns.get("_base_", [])
    for b in ([bases] if isinstance(bases, str) else bases):
        out.update(_run_config(os.path.join(os.path.dirname(path), b), _seen + (path,)))
    out.update({k: v for k, v in ns.items() if not k.startswith("_") and not callable(v) and not isinstance(v, type(os))})
    return out


def _early_dtype():
    """The MFMA operand type is a per-process choice (one library per type) and must be known before pixart_sigma_amd is imported:
    `mixed_precision = 'fp16'` in the config (every reference config: configs/PixArt_xl2_internal.py:57) selects the fp16-operand build +
    dynamic loss scaling, 'bf16' the bf16 build, 'no' / 'fp32' are refused (there is no fp32-operand MFMA path)."""
    cfgs = [a for a in sys.argv[1:] if a.endswith(".py") and os.path.exists(a)]
    mp = "bf16"
    if cfgs:
        mp = _run_config(cfgs[0]).get("mixed_precision", "bf16")
    early = argparse.ArgumentParser(add_help=False)          # both `--mixed-precision fp16` and `--mixed-precision=fp16`
    early.add_argument("--mixed-precision", default=None)
    mp = early.parse_known_args(sys.argv[1:])[0].mixed_precision or mp
    if mp not in ("fp16", "bf16"):
        raise SystemExit(f"mixed_precision={mp!r}: this path computes with bf16 or fp16 MFMA operands only")
    if mp == "fp16":
        os.environ["PXA_OPERAND_DTYPE"] = "f16"
    return mp


MIXED_PRECISION = _early_dtype() if __name__ == "__main__" else "bf16"
from pixart_sigma_amd import IDDPM, build_model  # noqa: E402
from pixart_sigma_amd.dp import FusedAdamW, FusedCAME, LossScaler  # noqa: E402
from pixart_sigma_amd.lr_schedule import LRSchedule, auto_scale_lr  # noqa: E402

DEFAULTS = dict(model="PixArtMS_XL_2", image_size=1024, train_batch_size=16, num_epochs=1, model_max_length=300, pred_sigma=True,
                learn_sigma=True, class_dropout_prob=0.1, kv_compress=False, kv_compress_config=None, micro_condition=False,
                grad_checkpointing=False, fp32_attention=False, gc_step=1, scale_factor=0.13025, gradient_clip=0.01,
                optimizer=dict(type="AdamW", lr=2e-5, weight_decay=3e-2, eps=1e-10), train_sampling_steps=1000, snr_loss=False,
                log_interval=20, save_model_steps=1000, seed=43, data_root=None, load_vae_feat=True,
                vae_pretrained="output/pretrained_models/pixart_sigma_sdxlvae_T5_diffusers/vae",
                # schedule / scaling keys of the reference configs (configs/PixArt_xl2_internal.py:34-57)
                gradient_accumulation_steps=1, auto_lr=None, lr_schedule="constant", lr_schedule_args=dict(num_warmup_steps=0),
                mixed_precision="bf16", aspect_ratio_type=None, valid_num=0, num_steps_per_epoch=None)
# keys of reference configs that name subsystems outside this path and cannot change what is trained: accepted and ignored.
# Anything else that is not in DEFAULTS raises: a recognised-but-unsupported option must not silently change what is trained.
IGNORED_KEYS = {"data", "image_list_json", "num_workers", "work_dir", "log_interval", "eval_sampling_steps", "visualize", "resume_from", "load_from",
                "validation_prompts", "save_model_epochs", "window_block_indexes", "window_size", "use_rel_pos", "lewei_scale", "pe_interpolation", "roots",
                "vae_pretrained", "tracker_project_name", "name", "num_ddim_timesteps", "w_max", "w_min", "cfg_scale", "image_size", "data_root",
                "aspect_ratio_type", "eval_metric", "model_max_length", "max_length"}
# keys that DO change what is trained and that this path implements for one value only: that value (or absence) is accepted, anything else refused
# (round-2 ADVICE: real_prompt_ratio = 0.5 of the Sigma configs mixes sharegpt4v captions in; FeatureDatasetMS always takes the real prompt).
SEMANTIC_KEYS = {"real_prompt_ratio": (1.0,), "qk_norm": (False,), "mask_loss_coef": (0.0, 0), "mask_type": ("null", None), "load_mask_index": (False,),
                 "loss_type": (None, "mse", "l2"), "huber_c": None, "ema_rate": None, "ema_decay": None, "skip_step": (0,), "conditional_dropout": (False, None),
                 "multi_scale": None, "use_fsdp": (False,)}      # None = any value: EMA is not kept, huber_c only matters under loss_type 'huber', multi_scale is the
                                                                  # dataset class choice (the bucketed FeatureDatasetMS handles both)


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("config", nargs="?", default=None)
    p.add_argument("--work-dir", "--work_dir", default="output/debug")
    p.add_argument("--resume-from", default=None)
    p.add_argument("--load-from", default=None)
    p.add_argument("--debug", action="store_true")
    p.add_argument("--synthetic", action="store_true")
    p.add_argument("--max-steps", type=int, default=None)
    p.add_argument("--mixed-precision", choices=["fp16", "bf16"], default=None, help="overrides the config's mixed_precision")
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


def load_config(path, debug=False):
    cfg = dict(DEFAULTS)
    if path:
        user = _run_config(path)
        unknown = sorted(k for k in user if k not in DEFAULTS and k not in IGNORED_KEYS and k not in SEMANTIC_KEYS)
        if unknown:
            raise SystemExit(f"{path}: config keys this training path does not implement: {unknown} (supported: {sorted(DEFAULTS)})")
        bad = {k: user[k] for k, ok in SEMANTIC_KEYS.items() if k in user and ok is not None and user[k] not in ok}
        if bad:
            raise SystemExit(f"{path}: these settings change what is trained and only one value is implemented here: "
                             + ", ".join(f"{k}={v!r} (implemented: {SEMANTIC_KEYS[k][0]!r})" for k, v in bad.items()))
        cfg.update({k: v for k, v in user.items() if k in DEFAULTS})
    if debug:
        cfg.update(train_batch_size=2, log_interval=1)
    if cfg["lr_schedule"] not in ("constant", "cosine", "cosine_decay_to_constant"):
        raise SystemExit(f"Unrecognized lr schedule {cfg['lr_schedule']}.")
    return cfg


def feature_batches(cfg, B, L, dev, rank, world):
    """Feature files in the reference's layout (InternalDataMSSigma + AspectRatioBatchSampler, reference train.py:404-421): every batch has
    one latent shape; yields (z, y, mask, data_info)."""
    from pixart_sigma_amd.data import AspectRatioBatchSampler, FeatureDatasetMS
    ratios = cfg["aspect_ratio_type"]
    if not isinstance(ratios, dict):
        raise SystemExit("aspect_ratio_type must be the bucket table itself: {ratio string: [H, W]} (the reference's ASPECT_RATIO_* dicts)")
    ds = FeatureDatasetMS(cfg["data_root"], ratios, resolution=cfg["image_size"], max_length=L)
    g = torch.Generator().manual_seed(cfg["seed"] + rank)
    while True:
        order = torch.randperm(len(ds), generator=g).tolist()[rank::world]
        for idx in AspectRatioBatchSampler(order, ds, B, ratios, drop_last=True, valid_num=cfg["valid_num"], ratio_nums={str(k): v for k, v in ds.ratio_nums.items()}):
            items = [ds.__getitem__(i, g) for i in idx]
            z = torch.stack([it[0] for it in items]).float().to(dev)
            y = torch.stack([it[1].float() for it in items]).to(dev)                    # (B, 1, L, 4096)
            mask = torch.cat([it[2].reshape(1, -1) for it in items]).long()             # host mask: no device sync for y_lens
            info = {"img_hw": torch.stack([it[3]["img_hw"] for it in items]), "aspect_ratio": torch.tensor([[it[3]["aspect_ratio"]] for it in items])}
            yield z, y, mask, info


def load_optimizer_state(opt, sd, rank):
    """Our own flat-buffer state (validated against the parameter layout) or a reference checkpoint's torch / CAME state_dict
    ('state' / 'param_groups': per-parameter tensors in the reference's parameter order) - the latter is skipped with a warning."""
    o = sd.get("optimizer")
    if o is None:
        return False
    if "param_groups" in o or "state" in o:
        if rank == 0:
            print("resume: the checkpoint carries a torch-format optimizer state (reference checkpoint); moments are re-initialised", flush=True)
        return False
    opt.load_state_dict(o)
    return True


def main():
    a = parse_args()
    cfg = load_config(a.config, a.debug)
    cfg["mixed_precision"] = MIXED_PRECISION
    world, rank, local = int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    torch.manual_seed(cfg["seed"])
    lat, L, B, accum = cfg["image_size"] // 8, cfg["model_max_length"], cfg["train_batch_size"], int(cfg["gradient_accumulation_steps"])
    model = build_model(cfg["model"], cfg["grad_checkpointing"], cfg["fp32_attention"], gc_step=cfg["gc_step"], input_size=lat,
                        pe_interpolation=cfg["image_size"] / 512, model_max_length=L, micro_condition=cfg["micro_condition"],
                        kv_compress_config=cfg["kv_compress_config"] if cfg["kv_compress"] else None,
                        pred_sigma=cfg["pred_sigma"], learn_sigma=cfg["learn_sigma"], class_dropout_prob=cfg["class_dropout_prob"])
    start_step, sd = 0, None
    ck = a.resume_from or a.load_from
    if ck:
        sd = torch.load(ck, map_location="cpu", weights_only=False)
        model.load_state_dict(sd.get("state_dict", sd), strict=False)
    model = model.to(dev).train()
    model.prepare(dev)
    # ---- learning rate: batch-size scaling (reference train.py:448-452) + warm-up schedule (lr_scheduler.py:10-40)
    o = dict(cfg["optimizer"])
    ratio = 1.0
    if cfg["auto_lr"]:
        o["lr"], ratio = auto_scale_lr(B * world * accum, o["lr"], **cfg["auto_lr"])
    # accelerate steps the prepared scheduler `world` times per optimizer step (split_batches=False, the reference's setting): steps_per_call
    sched = LRSchedule(o["lr"], cfg["lr_schedule"], lr_scale_ratio=ratio, steps_per_call=world,
                       num_training_steps=(cfg["num_steps_per_epoch"] or 0) * cfg["num_epochs"] or None, **(cfg["lr_schedule_args"] or {}))
    scaler = LossScaler(dev) if cfg["mixed_precision"] == "fp16" else None      # GradScaler protocol of accelerate's fp16 mode, on the device
    if o.get("type", "AdamW") in ("CAMEWrapper", "CAME"):      # the optimizer of the PixArt-Sigma configs (reference optimizer.py:242-246)
        opt = FusedCAME(model, lr=o["lr"], weight_decay=o.get("weight_decay", 0.0), eps=o.get("eps", (1e-30, 1e-16)),
                        betas=o.get("betas", (0.9, 0.999, 0.9999)), max_grad_norm=cfg["gradient_clip"], scaler=scaler)
    else:
        opt = FusedAdamW(model, lr=o["lr"], weight_decay=o["weight_decay"], eps=o.get("eps", 1e-8), betas=o.get("betas", (0.9, 0.999)),
                         max_grad_norm=cfg["gradient_clip"], scaler=scaler)
    if a.resume_from and sd is not None:
        if load_optimizer_state(opt, sd, rank):
            start_step = int(sd.get("step", 0))
        if "lr_scheduler" in sd:
            sched.load_state_dict(sd["lr_scheduler"], optimizer_step=sd.get("step"))
        elif "scheduler" in sd:                                 # reference checkpoint: torch LambdaLR state (save_checkpoint)
            sched.load_state_dict(sd["scheduler"])
        if scaler is not None and "loss_scaler" in sd:
            scaler.load_state_dict(sd["loss_scaler"])
        if "_step_" in a.resume_from and not start_step:
            start_step = int(os.path.basename(a.resume_from).split("_step_")[-1].split(".")[0])
    if rank == 0:
        print(f"lr {o['lr']:.3e} (auto_lr x{ratio:.3f}), schedule {cfg['lr_schedule']} {cfg['lr_schedule_args']}, accumulation {accum}, "
              f"operands {cfg['mixed_precision']}" + (f", loss scale {scaler.value:g}" if scaler else ""), flush=True)
    diff = IDDPM(str(cfg["train_sampling_steps"]), learn_sigma=cfg["learn_sigma"], pred_sigma=cfg["pred_sigma"], snr=cfg["snr_loss"])
    vae = None
    if not cfg["load_vae_feat"]:                                                   # reference train.py:351-354
        from pixart_sigma_amd.vae import AutoencoderKL
        have = os.path.isdir(str(cfg["vae_pretrained"]))
        vae = (AutoencoderKL.from_pretrained(cfg["vae_pretrained"], torch_dtype=torch.float16) if have else AutoencoderKL(scaling_factor=cfg["scale_factor"])).to(dev)
        cfg["scale_factor"] = vae.config.scaling_factor
    os.makedirs(os.path.join(a.work_dir, "checkpoints"), exist_ok=True)
    use_ds = bool(cfg["data_root"]) and not a.synthetic and os.path.exists(os.path.join(cfg["data_root"], "data_info.json"))
    it = feature_batches(cfg, B, L, dev, rank, world) if use_ds else batches(cfg, B, lat, L, dev, rank, world, a.synthetic)
    t0, step = time.time(), start_step
    sched_skips_seen = scaler.steps_skipped if scaler else 0
    while a.max_steps is None or step < start_step + a.max_steps:
        opt.zero_grad()
        opt.lr = sched.lr
        for micro in range(accum):                                                 # accelerator.accumulate (reference train.py:175)
            batch = next(it)
            z, y, mask = batch[:3]
            # micro-conditioning inputs (reference train.py:156): from the dataset, or the latent's own size for flat / synthetic data
            info = batch[3] if len(batch) > 3 else {"img_hw": torch.tensor([[z.shape[-2] * 8.0, z.shape[-1] * 8.0]] * z.shape[0]),
                                                    "aspect_ratio": torch.tensor([[z.shape[-2] / z.shape[-1]]] * z.shape[0])}
            if vae is not None:                                                    # reference train.py:147-153
                z = vae.encode(z).latent_dist.sample().float()
            x0 = z * cfg["scale_factor"]
            t = torch.randint(0, cfg["train_sampling_steps"], (z.shape[0],), device=dev).long()
            loss = diff.training_losses(model, x0, t, model_kwargs=dict(y=y, mask=mask, data_info=info))["loss"].mean() / accum
            if micro + 1 < accum:
                with opt.reducer.no_sync():                                        # local accumulation, no all-reduce yet
                    (scaler.scale(loss) if scaler else loss).backward()
            else:
                (scaler.scale(loss) if scaler else loss).backward()
        opt.step()
        sched.step()
        step += 1
        saving = step % cfg["save_model_steps"] == 0
        if scaler and (step % cfg["log_interval"] == 0 or saving):   # accelerate does not advance the schedule on a step the GradScaler skipped; the skip count
            skipped = scaler.steps_skipped                     # lives on the device, so the schedule is reconciled where the host syncs anyway (skips are rare) -
            sched.last_step -= world * (skipped - sched_skips_seen)   # and ALWAYS in front of a checkpoint: a resumed run starts its own count from the loaded
            sched_skips_seen = skipped                         # scaler's total, so skips not yet subtracted at save time would never be
        if step % cfg["log_interval"] == 0 and rank == 0:      # host sync only here (the reference syncs every step, train.py:187)
            extra = f" loss_scale {scaler.value:g} skipped {scaler.steps_skipped}" if scaler else ""
            print(f"step {step} loss {loss.item() * accum:.4f} grad_norm {opt.last_norm.item():.4f} lr {opt.lr:.3e}{extra} "
                  f"{(time.time() - t0) / cfg['log_interval']:.3f} s/step", flush=True)
            t0 = time.time()
        if saving and rank == 0:
            torch.save({"state_dict": model.state_dict(), "optimizer": opt.state_dict(), "lr_scheduler": sched.state_dict(), "step": step,
                        **({"loss_scaler": scaler.state_dict()} if scaler else {})},
                       os.path.join(a.work_dir, "checkpoints", f"epoch_1_step_{step}.pth"))
    if rank == 0:
        print(f"finished at step {step}: loss {loss.item() * accum:.4f} grad_norm {opt.last_norm.item():.4f} lr {opt.lr:.3e}", flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
