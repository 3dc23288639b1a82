"""Learning-rate schedules and batch-size scaling of the reference's training loop, as host scalars: the fused optimizers take `lr` as a
plain float per step (no parameter groups to walk), so a schedule is a pure function of the step count.

  * `auto_scale_lr`  - reference diffusion/utils/optimizer.py:18-28 (`auto_lr = dict(rule='sqrt'|'linear')`, base batch 256; the effective
    batch is train_batch_size x world_size x gradient_accumulation_steps, train_scripts/train.py:448-452)
  * `LRSchedule`     - reference diffusion/utils/lr_scheduler.py:10-40: 'constant' (diffusers get_constant_schedule_with_warmup: linear
    warm-up over num_warmup_steps, then 1), 'cosine' (get_cosine_schedule_with_warmup) and 'cosine_decay_to_constant' (lines 43-88).
    `step()` is called once per optimizer step, like `lr_scheduler.step()` at train.py:184; state = the scheduler's step count (checkpointed).
    What one call advances (round 3): the reference passes its scheduler through `accelerator.prepare` with the default
    `split_batches=False`, and accelerate's AcceleratedScheduler then steps the wrapped scheduler `num_processes` times per optimizer step and
    not at all when the fp16 GradScaler skipped the step.  So on 8 GPUs the 1000-step warm-up of the Sigma configs is over after 125 optimizer
    steps, and a reference checkpoint's `scheduler.last_epoch` counts world x applied steps.  `steps_per_call` (= world size in train.py) and
    `step(applied=...)` reproduce both.
"""
import os
import warnings
import math


def auto_scale_lr(effective_bs, base_lr, rule="linear", base_batch_size=256):
    if rule not in ("linear", "sqrt"):
        raise ValueError(f"auto_lr rule must be 'linear' or 'sqrt', got {rule!r}")
    ratio = math.sqrt(effective_bs / base_batch_size) if rule == "sqrt" else effective_bs / base_batch_size
    return base_lr * ratio, ratio


class LRSchedule:
    def __init__(self, base_lr, schedule="constant", num_warmup_steps=0, num_training_steps=None, lr_scale_ratio=1.0, num_decay=0.667,
                 num_cycles=0.5, steps_per_call=1):
        if schedule not in ("constant", "cosine", "cosine_decay_to_constant"):
            raise RuntimeError(f"Unrecognized lr schedule {schedule}.")          # same error as lr_scheduler.py:39
        if schedule != "constant" and not num_training_steps:
            raise ValueError(f"lr schedule {schedule!r} needs num_training_steps")
        if schedule == "cosine_decay_to_constant" and lr_scale_ratio < 1.0:
            raise AssertionError(f"lr_scale_ratio {lr_scale_ratio} < 1")           # reference lr_scheduler.py:31 asserts the same
        self.steps_per_call = int(steps_per_call)
        self.base_lr, self.schedule = base_lr, schedule
        self.warmup, self.total = int(num_warmup_steps), num_training_steps
        self.final = 1.0 / lr_scale_ratio
        self.num_decay, self.num_cycles = num_decay, num_cycles
        self.last_step = 0

    def factor(self, step):
        if step < self.warmup:
            return float(step) / float(max(1, self.warmup))
        if self.schedule == "constant":
            return 1.0
        if self.schedule == "cosine":
            prog = float(step - self.warmup) / float(max(1, self.total - self.warmup))
            return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(self.num_cycles) * 2.0 * prog)))
        decay_steps = int(self.total * self.num_decay)
        if step > decay_steps:
            return self.final
        prog = float(step - self.warmup) / float(max(1, decay_steps - self.warmup))
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(self.num_cycles) * 2.0 * prog))) * (1 - self.final) + self.final

    @property
    def lr(self):
        """learning rate of the NEXT optimizer step (torch LambdaLR semantics: lr after k scheduler steps = base * factor(k))"""
        return self.base_lr * self.factor(self.last_step)

    def step(self, applied=True):
        """one optimizer step of the job; `applied=False` = the loss scaler skipped it (the schedule does not move)"""
        if applied:
            self.last_step += self.steps_per_call
        return self.lr

    def state_dict(self):
        return {"last_step": self.last_step, "base_lr": self.base_lr, "steps_per_call": self.steps_per_call}

    def load_state_dict(self, sd, optimizer_step=None):
        """`last_step` counts scheduler steps = steps_per_call x applied optimizer steps (the reference's `last_epoch` on a `world`-GPU job, see the module
        header).  A checkpoint of this repo that RECORDED its unit (`steps_per_call`, round 4 on) and is resumed on another number of GPUs is rescaled to this
        run's unit.  A checkpoint without the key is never guessed at (round-3 files already counted world x steps and carried no unit, so reading
        every unit-less file as round 2's optimizer steps multiplied them by world a second time): the value is loaded as it stands - right for round 3
        and for the reference's `last_epoch` - unless the caller passes the checkpoint's own `step` (`optimizer_step`) and the two say unambiguously that the
        counter is in optimizer steps (last_step == step, world > 1): only then is it scaled by this run's steps_per_call."""
        last = int(sd.get("last_step", sd.get("last_epoch", 0)))             # 'last_epoch' = torch LambdaLR's name for the same counter
        if "steps_per_call" in sd:
            saved_unit = max(1, int(sd["steps_per_call"]))
            if saved_unit != self.steps_per_call:
                last = last * self.steps_per_call // saved_unit
        else:
            # unit-less checkpoint.  PXA_LR_CKPT_UNIT states it explicitly: "optimizer" = optimizer steps (round 2's files: scaled by this run's
            # steps_per_call), "scheduler" = scheduler steps (round 3's files, the reference's last_epoch: loaded as it stands).  Without it the heuristic below
            # decides - and says so, because last_step == step can also be a coincidence (a scheduler restarted mid-run).
            unit = os.environ.get("PXA_LR_CKPT_UNIT", "").lower()
            if unit == "optimizer":
                last = last * self.steps_per_call
            elif unit != "scheduler" and optimizer_step and self.steps_per_call > 1 and last == int(optimizer_step):
                warnings.warn(f"LRSchedule.load_state_dict: the checkpoint carries no scheduler unit and last_step == step == {last}: reading it as OPTIMIZER steps and "
                              f"scaling by steps_per_call = {self.steps_per_call}; set PXA_LR_CKPT_UNIT=scheduler (or optimizer) to state the unit")
                last = last * self.steps_per_call
        self.last_step = last
