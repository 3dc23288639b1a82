"""Aspect-ratio bucketing for multi-scale training: every batch the denoiser sees has ONE latent shape (h, w), so batches are
formed per aspect-ratio bucket.  Behaviour of the reference's `AspectRatioBatchSampler` (`diffusion/utils/data_sampler.py:10-76`):
walk the base sampler, put each index into the bucket of the closest predefined ratio (height / width, keys are the ratio as a
string, values the [H, W] the images of that bucket are resized to), emit a batch the moment a bucket holds `batch_size` indices,
flush the partly filled buckets at the end (unless drop_last), and ignore buckets that hold fewer than `valid_num` samples in the
dataset-wide count `ratio_nums`.  The ratio table itself (e.g. the reference's ASPECT_RATIO_1024) is an input, as it is there."""


def closest_ratio(height, width, aspect_ratios):
    """Key of `aspect_ratios` (ratio strings -> [H, W]) nearest to height / width, and its [H, W]
    (reference `get_closest_ratio`, `diffusion/data/datasets/utils.py`)."""
    r = height / width
    key = min(aspect_ratios.keys(), key=lambda k: abs(float(k) - r))
    return key, aspect_ratios[key]


class AspectRatioBatchSampler:
    def __init__(self, sampler, dataset, batch_size, aspect_ratios, drop_last=False, config=None, valid_num=0, ratio_nums=None, **kwargs):
        if not isinstance(batch_size, int) or batch_size <= 0:
            raise ValueError(f"batch_size should be a positive integer value, but got batch_size={batch_size}")
        if not ratio_nums:
            raise AssertionError("ratio_nums (samples per aspect-ratio bucket over the dataset) is required")
        self.sampler, self.dataset, self.batch_size, self.aspect_ratios, self.drop_last = sampler, dataset, batch_size, aspect_ratios, drop_last
        self.valid = {str(k) for k, n in ratio_nums.items() if n >= valid_num}
        self.buckets = {k: [] for k in aspect_ratios}

    def __iter__(self):
        for idx in self.sampler:
            info = self.dataset.get_data_info(idx)
            key, _ = closest_ratio(info["height"], info["width"], self.aspect_ratios)
            if key not in self.valid:
                continue
            b = self.buckets[key]
            b.append(idx)
            if len(b) == self.batch_size:
                yield list(b)
                b.clear()
        for key, b in self.buckets.items():                  # leftovers: always shorter than a batch (full ones left above)
            self.buckets[key] = []
            if b and not self.drop_last:
                yield b
