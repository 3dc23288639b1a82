"""Helpers with the reference's names (diffusion/model/utils.py:17-45)."""
from collections.abc import Iterable
from itertools import repeat

import torch.nn as nn


def _ntuple(n):
    def parse(x):
        if isinstance(x, Iterable) and not isinstance(x, str):
            return x
        return tuple(repeat(x, n))
    return parse


to_2tuple = _ntuple(2)


def set_grad_checkpoint(model, use_fp32_attention=False, gc_step=1):
    """Sets the three attributes the reference sets on every submodule (utils.py:28-35).  In this implementation
    `grad_checkpointing` switches the engine to keep only each block's input and recompute the block in backward;
    `fp32_attention` is accepted for compatibility (softmax statistics are always fp32 here, operands bf16)."""
    assert isinstance(model, nn.Module)

    def set_attr(module):
        module.grad_checkpointing = True
        module.fp32_attention = use_fp32_attention
        module.grad_checkpointing_step = gc_step
    model.apply(set_attr)
