"""Learning-rate plumbing of the training entry point: schedule values equal torch's LambdaLR running the reference's /
diffusers' lambdas, auto-lr equals reference diffusion/utils/optimizer.py:18-28, state round-trips."""
import math

import pytest

import torch

from pixart_sigma_amd.lr_schedule import LRSchedule, auto_scale_lr


def _torch_lrs(lmbda, base, n):
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=base)
    sch = torch.optim.lr_scheduler.LambdaLR(opt, lmbda)
    out = []
    for _ in range(n):
        out.append(opt.param_groups[0]["lr"])
        opt.step()
        sch.step()
    return out


def test_constant_with_warmup_equals_diffusers_lambda():
    warm = 500          # configs/PixArt_xl2_internal.py:50
    s = LRSchedule(2e-5, "constant", num_warmup_steps=warm)
    ref = _torch_lrs(lambda k: float(k) / float(max(1.0, warm)) if k < warm else 1.0, 2e-5, 620)      # diffusers get_constant_schedule_with_warmup
    mine = []
    for _ in range(620):
        mine.append(s.lr)
        s.step()
    assert mine == ref and mine[0] == 0.0 and mine[warm] == 2e-5


def test_cosine_equals_diffusers_lambda():
    warm, total = 10, 100
    s = LRSchedule(1e-4, "cosine", num_warmup_steps=warm, num_training_steps=total)

    def lam(k):
        if k < warm:
            return float(k) / float(max(1, warm))
        prog = float(k - warm) / float(max(1, total - warm))
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * 0.5 * 2.0 * prog)))
    ref = _torch_lrs(lam, 1e-4, total)
    assert [s.base_lr * s.factor(k) for k in range(total)] == ref


def test_cosine_decay_to_constant_equals_reference(golden):
    for case in golden("compat")["lr"]["cosine_decay"]:
        s = LRSchedule(2e-5, "cosine_decay_to_constant", num_warmup_steps=case["warm"], num_training_steps=case["total"], lr_scale_ratio=1.0 / case["final"])
        mine = []
        for _ in case["lrs"]:
            mine.append(s.lr)
            s.step()
        assert all(abs(a - b) <= 1e-12 * max(abs(b), 1e-30) + 1e-18 for a, b in zip(mine, case["lrs"])), case["warm"]


def test_auto_scale_lr_rules():
    # reference: lr *= sqrt(effective_bs / 256) or effective_bs / 256, effective_bs = train_batch_size * world * accumulation
    lr, r = auto_scale_lr(16 * 8 * 1, 2e-5, rule="sqrt")
    assert r == math.sqrt(128 / 256) and lr == 2e-5 * r
    lr, r = auto_scale_lr(64 * 8 * 2, 1e-4, rule="linear")
    assert r == 4.0 and lr == 4e-4


def test_state_roundtrip_and_reference_checkpoint_key():
    s = LRSchedule(2e-5, "constant", num_warmup_steps=100)
    for _ in range(37):
        s.step()
    t = LRSchedule(2e-5, "constant", num_warmup_steps=100)
    t.load_state_dict(s.state_dict())
    assert t.lr == s.lr and t.last_step == 37
    t.load_state_dict({"last_epoch": 50})          # torch LambdaLR state of a reference checkpoint
    assert t.last_step == 50 and t.lr == 1e-5


def test_world_steps_per_optimizer_step_and_skipped_steps():
    """accelerate's AcceleratedScheduler (the reference prepares its scheduler without split_batches): `world` scheduler steps per optimizer step,
    none when the GradScaler skipped the step - a 1000-step warm-up is over after 125 optimizer steps on 8 GPUs."""
    s = LRSchedule(2e-5, "constant", num_warmup_steps=1000, steps_per_call=8)
    for _ in range(124):
        s.step()
    assert s.last_step == 992 and s.lr < 2e-5
    s.step(applied=False)                     # overflow step: the schedule does not move
    assert s.last_step == 992
    s.step()
    assert s.last_step == 1000 and s.lr == 2e-5
    t = LRSchedule(2e-5, "constant", num_warmup_steps=1000, steps_per_call=8)
    t.load_state_dict({"last_epoch": 1000})   # a reference checkpoint written after 125 optimizer steps on 8 GPUs
    assert t.lr == s.lr


def test_cosine_decay_to_constant_rejects_ratio_below_one():
    import pytest
    with pytest.raises(AssertionError):
        LRSchedule(2e-5, "cosine_decay_to_constant", num_warmup_steps=10, num_training_steps=100, lr_scale_ratio=0.7)


def test_state_dict_records_its_unit_and_rescales_on_load():
    """`last_step` is in world x optimizer-step units; a checkpoint in another unit (round-2 format = optimizer steps, or another GPU count)
    resumes at the same point of the schedule."""
    from pixart_sigma_amd.lr_schedule import LRSchedule
    a = LRSchedule(1e-4, "constant", num_warmup_steps=1000, steps_per_call=8)
    for _ in range(50):
        a.step()
    sd = a.state_dict()
    assert sd["last_step"] == 400 and sd["steps_per_call"] == 8
    b = LRSchedule(1e-4, "constant", num_warmup_steps=1000, steps_per_call=8)
    b.load_state_dict(sd)
    assert b.last_step == 400 and b.lr == a.lr
    c = LRSchedule(1e-4, "constant", num_warmup_steps=1000, steps_per_call=2)      # resumed on 2 GPUs: 50 optimizer steps = 100 scheduler steps there
    c.load_state_dict(sd)
    assert c.last_step == 100
    # a checkpoint WITHOUT its unit is not guessed at: round-3 files already count world x steps ...
    d = LRSchedule(1e-4, "constant", num_warmup_steps=1000, steps_per_call=8)
    d.load_state_dict({"last_step": 400, "base_lr": 1e-4})
    assert d.last_step == 400
    d.load_state_dict({"last_step": 400, "base_lr": 1e-4}, optimizer_step=50)        # ... also when the checkpoint's own step count is known
    assert d.last_step == 400
    # ... and a round-2 file (optimizer steps) is rescaled only when the checkpoint's own `step` proves the unit
    d.load_state_dict({"last_step": 50, "base_lr": 1e-4}, optimizer_step=50)
    assert d.last_step == 400
    d.load_state_dict({"last_step": 50, "base_lr": 1e-4})
    assert d.last_step == 50
    one = LRSchedule(1e-4, "constant", num_warmup_steps=1000, steps_per_call=1)       # one GPU: both readings coincide
    one.load_state_dict({"last_step": 50, "base_lr": 1e-4}, optimizer_step=50)
    assert one.last_step == 50
    e = LRSchedule(1e-4, "constant", num_warmup_steps=1000, steps_per_call=8)
    e.load_state_dict({"last_epoch": 400})                                          # a reference checkpoint's LambdaLR state: already world x steps
    assert e.last_step == 400


def test_unitless_checkpoint_heuristic_warns_and_can_be_overridden(monkeypatch):
    """when last_step == step decides the unit of a unit-less checkpoint, the schedule says so; PXA_LR_CKPT_UNIT states the unit explicitly."""
    import warnings
    from pixart_sigma_amd.lr_schedule import LRSchedule
    s = LRSchedule(1e-4, "constant", num_warmup_steps=1000, steps_per_call=8)
    with pytest.warns(UserWarning, match="no scheduler unit"):
        s.load_state_dict({"last_step": 50}, optimizer_step=50)
    assert s.last_step == 400
    monkeypatch.setenv("PXA_LR_CKPT_UNIT", "scheduler")          # "it already counts scheduler steps": loaded as it stands, no warning
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        s.load_state_dict({"last_step": 50}, optimizer_step=50)
    assert s.last_step == 50
    monkeypatch.setenv("PXA_LR_CKPT_UNIT", "optimizer")          # "optimizer steps": scaled, whatever `step` says
    s.load_state_dict({"last_step": 50})
    assert s.last_step == 400
