"""Aspect-ratio bucket sampler (behaviour of reference diffusion/utils/data_sampler.py:10-76)."""
import random

from pixart_sigma_amd.data import AspectRatioBatchSampler, closest_ratio

RATIOS = {"0.5": [704.0, 1408.0], "1.0": [1024.0, 1024.0], "2.0": [1408.0, 704.0]}


class DS:
    def __init__(self, sizes):
        self.sizes = sizes

    def get_data_info(self, i):
        return {"height": self.sizes[i][0], "width": self.sizes[i][1]}


def test_closest_ratio():
    assert closest_ratio(700, 1400, RATIOS) == ("0.5", [704.0, 1408.0])
    assert closest_ratio(1000, 1100, RATIOS)[0] == "1.0" and closest_ratio(3000, 1000, RATIOS)[0] == "2.0"


def test_batches_share_one_bucket_and_cover_everything():
    rnd = random.Random(0)
    sizes = [rnd.choice([(512, 1024), (800, 800), (1024, 512), (900, 1000)]) for _ in range(103)]
    ds = DS(sizes)
    nums = {"0.5": 30, "1.0": 50, "2.0": 23}
    s = AspectRatioBatchSampler(range(len(sizes)), ds, 8, RATIOS, ratio_nums=nums)
    batches = list(s)
    seen = sorted(i for b in batches for i in b)
    assert seen == list(range(103))
    for b in batches:
        assert len({closest_ratio(*sizes[i], RATIOS)[0] for i in b}) == 1 and 1 <= len(b) <= 8
    assert sum(len(b) < 8 for b in batches) <= 3                  # at most one short batch per bucket
    full_first = [len(b) for b in batches][: len(batches) - 3]
    assert all(n == 8 for n in full_first)
    dropped = list(AspectRatioBatchSampler(range(len(sizes)), ds, 8, RATIOS, drop_last=True, ratio_nums=nums))
    assert all(len(b) == 8 for b in dropped) and len(dropped) == sum(len(b) == 8 for b in batches)
    few = list(AspectRatioBatchSampler(range(len(sizes)), ds, 8, RATIOS, ratio_nums=nums, valid_num=25))      # the "2.0" bucket is ignored
    assert all(closest_ratio(*sizes[i], RATIOS)[0] != "2.0" for b in few for i in b)
