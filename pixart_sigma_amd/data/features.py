"""Precomputed-feature dataset in the reference's on-disk layout (what tools/extract_features.py writes and
diffusion/data/datasets/InternalData_ms.py:170-341 `InternalDataMSSigma` reads with load_vae_feat = load_t5_feat = True):

    <root>/data_info.json                                             [{"path", "height", "width", "ratio", "prompt", ["sharegpt4v"]}, ...]
    <root>/caption_features_new/<dir>_<name>.npz                      caption_feature (1, L_c, 4096) + attention_mask (1, L_c)
    <root>/img_sdxl_vae_features_<res>resolution_ms_new/<dir>_<name>.npy   cat[mean, std] of the VAE posterior, (8, h, w)

`__getitem__` returns what the reference's `getdata` returns: (latent sample = mean + std * N(0,1), caption features padded to
max_length by repeating the last token, attention mask (1, 1, max_length) int16 zero-padded, data_info = {img_hw: [H, W] of the original
image, aspect_ratio: the closest ratio of the bucket table, mask_type}).  Items with ratio > 4.5 are dropped, as there.
`ratio_nums` (samples per bucket over the first third of the list, InternalData_ms.py:276-281) feeds AspectRatioBatchSampler."""
import json
import os

import numpy as np
import torch

from .sampler import closest_ratio

_IMG_EXT = (".png", ".jpg", ".webp", ".jpeg", ".JPEG", ".JPG")


def replace_img_ext(path, dst_ext):
    for e in _IMG_EXT:
        path = path.replace(e, dst_ext)
    return path


def vae_feat_loader(path, generator=None):
    """cat[mean, std] -> one posterior sample (InternalData.py:296-301)."""
    mean, std = torch.from_numpy(np.load(path)).chunk(2)
    return mean + std * torch.randn(mean.shape, generator=generator, dtype=mean.dtype)


class FeatureDatasetMS(torch.utils.data.Dataset):
    def __init__(self, root, aspect_ratios, resolution=1024, image_list_json="data_info.json", max_length=300, mask_type="null",
                 real_prompt_ratio=1.0, weight_dtype=torch.float16):
        self.root, self.aspect_ratio, self.max_length, self.mask_type = root, aspect_ratios, max_length, mask_type
        self.weight_dtype = weight_dtype if real_prompt_ratio > 0 else torch.float32
        lists = image_list_json if isinstance(image_list_json, (list, tuple)) else [image_list_json]
        self.meta, self.txt_feat, self.vae_feat = [], [], []
        self.ori_imgs_nums = 0
        for jf in lists:
            with open(os.path.join(root, jf)) as f:
                meta = json.load(f)
            self.ori_imgs_nums += len(meta)
            keep = [it for it in meta if it["ratio"] <= 4.5]
            self.meta += keep
            flat = [replace_img_ext("_".join(it["path"].rsplit("/", 1)), "") for it in keep]
            self.txt_feat += [os.path.join(root, "caption_features_new", n + ".npz") for n in flat]
            self.vae_feat += [os.path.join(root, f"img_sdxl_vae_features_{resolution}resolution_ms_new", n + ".npy") for n in flat]
        self.ratio_nums = {float(k): 0 for k in aspect_ratios}
        for it in self.meta[: len(self.meta) // 3]:
            self.ratio_nums[float(closest_ratio(it["height"], it["width"], aspect_ratios)[0])] += 1

    def __len__(self):
        return len(self.meta)

    def get_data_info(self, idx):
        return {"height": self.meta[idx]["height"], "width": self.meta[idx]["width"]}

    def __getitem__(self, idx, generator=None):
        it = self.meta[idx]
        key, _ = closest_ratio(it["height"], it["width"], self.aspect_ratio)
        img = vae_feat_loader(self.vae_feat[idx], generator)
        data_info = {"img_hw": torch.tensor([it["height"], it["width"]], dtype=torch.float32), "aspect_ratio": float(key), "mask_type": self.mask_type}
        z = np.load(self.txt_feat[idx])
        txt = torch.from_numpy(z["caption_feature"])
        mask = torch.from_numpy(z["attention_mask"])[None] if "attention_mask" in z.files else torch.ones(1, 1, self.max_length)
        if txt.shape[1] != self.max_length:                 # pad by repeating the last token; the mask marks the pad as invalid
            pad = self.max_length - txt.shape[1]
            txt = torch.cat([txt, txt[:, -1:].repeat(1, pad, 1)], dim=1).to(self.weight_dtype)
            mask = torch.cat([mask, torch.zeros(1, 1, self.max_length - mask.shape[-1])], dim=-1)
        return img, txt, mask.to(torch.int16), data_info
