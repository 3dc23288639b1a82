"""Callers' side of the path against vectors produced by the UNMODIFIED reference under stubs
(oracle/make_compat_golden.py -> tests/golden/compat.pt):
  * checkpoint_compat.to_diffusers == the dict tools/convert_pixart_to_diffusers.py:23-155 hands to Transformer2DModel.load_state_dict,
    key for key and bit for bit, for a depth-28 alpha checkpoint with micro-conditioning;
  * data.FeatureDatasetMS.__getitem__ == InternalDataMSSigma.getdata on the same feature directory (same RNG stream for the posterior
    sample): latent sample, padded caption features, int16 mask, img_hw, bucket ratio, the ratio > 4.5 filter, ratio_nums;
  * data.AspectRatioBatchSampler == the reference sampler's first epoch (valid_num and drop_last included)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _g(golden):
    return golden("compat")


def test_to_diffusers_equals_reference_converter(golden):
    from oracle.make_compat_golden import mini_state_dict
    from pixart_sigma_amd.model.checkpoint_compat import from_diffusers, to_diffusers
    g = _g(golden)["converter"]
    sd = mini_state_dict(**g["mini_args"])
    mine = to_diffusers({"state_dict": dict(sd)})
    ref = g["converted"]
    assert set(mine) == set(ref), (sorted(set(mine) ^ set(ref))[:8])
    for k in ref:
        assert torch.equal(mine[k], ref[k]), k
    back = from_diffusers(mine)
    assert all(torch.equal(back[k], sd[k]) for k in back) and set(sd) - set(back) == {"pos_embed", "y_embedder.y_embedding"}
    assert g["ctor"]["use_additional_conditions"] is True and g["ctor"]["attention_head_dim"] == 72     # the layout the keys belong to


def test_feature_dataset_equals_reference_getdata(golden, tmp_path):
    from oracle.make_compat_golden import write_feature_dir
    from pixart_sigma_amd.data import FeatureDatasetMS
    g = _g(golden)["dataset"]
    root = str(tmp_path / "InternData")
    write_feature_dir(root, g["ratios"])
    ds = FeatureDatasetMS(root, g["ratios"], resolution=1024, max_length=300)
    assert len(ds) == g["len"] == 3 and ds.ori_imgs_nums == g["ori"] == 4            # the ratio-5.0 image is dropped
    assert {k: v for k, v in ds.ratio_nums.items() if v} == {k: v for k, v in g["ratio_nums"].items() if v}
    for i, ref in enumerate(g["items"]):
        torch.manual_seed(100 + i)
        img, txt, mask, info = ds[i]
        assert torch.equal(img, ref["img"]) and img.dtype == ref["img"].dtype
        assert torch.equal(txt, ref["txt"]) and txt.dtype == ref["txt"].dtype and txt.shape[1] == 300
        assert torch.equal(mask, ref["mask"]) and mask.dtype == torch.int16 and mask.shape == (1, 1, 300)
        assert torch.equal(info["img_hw"], ref["img_hw"]) and info["aspect_ratio"] == ref["aspect_ratio"] and info["mask_type"] == ref["mask_type"]
    assert int(g["items"][0]["mask"].sum()) == 7 and int(g["items"][1]["mask"].sum()) == 300


def test_bucket_sampler_equals_reference_first_epoch(golden):
    from pixart_sigma_amd.data import AspectRatioBatchSampler
    g = _g(golden)["sampler"]
    sizes = g["sizes"]

    class DS:
        def get_data_info(self, i):
            return {"height": sizes[i][0], "width": sizes[i][1]}
    ratios = {"0.5": [704.0, 1408.0], "1.0": [1024.0, 1024.0], "2.0": [1408.0, 704.0]}
    for run in g["runs"]:
        s = AspectRatioBatchSampler(range(len(sizes)), DS(), run["batch_size"], ratios, drop_last=run["drop_last"], valid_num=run["valid_num"],
                                    ratio_nums=run["ratio_nums"])
        assert [list(b) for b in s] == run["batches"]
