"""Data-parallel training runtime for the denoiser: one process per GPU, RCCL (torch.distributed backend "nccl" on ROCm)
over xGMI, gradient all-reduce overlapped with the hand-sequenced backward, fused clip + AdamW on flat buffers.

Replaces what the reference gets from accelerate's DDP wrapper + torch AdamW (train_scripts/train.py:180-184,318-326,486):
  * the flat gradient buffer of engine.ParamStore is partitioned into buckets that follow the *completion order* of our
    backward: final layer, block L-1 ... block 0 (each ~21.3 M fp32 = 85 MB, one collective per bucket: large messages keep
    the 7 xGMI links per GPU busy; DDP's 25 MB default would issue ~4x more, smaller collectives), then the
    "cond" group (embedders + all scale_shift_tables, whose gradients PyTorch autograd finishes after the token path);
  * a bucket's all-reduce (SUM) is launched the moment the engine reports it complete and runs on RCCL's stream while the
    next block's backward kernels run on the compute stream;
  * averaging (1/world), global-norm clipping and AdamW are one pass: sumsq -> clip_coef (device scalar) -> adamw_step,
    which also refreshes the bf16 shadow weights the GEMMs read.  No host synchronisation anywhere in the step.
"""
import torch
import torch.distributed as dist


class GradReducer:
    """Bucketed, overlapped gradient all-reduce over the flat gradient buffer.  Backend-agnostic (nccl on GPUs, gloo in the
    CPU tests)."""

    def __init__(self, store, group=None):
        self.store = store
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.pending = []
        self.launched = set()

    def on_group_ready(self, name):
        """Engine hook: gradients of parameter group `name` are final for this step."""
        if self.world == 1 or name not in self.store.groups or name in self.launched:
            return
        s, e = self.store.groups[name]
        self.launched.add(name)
        self.pending.append(dist.all_reduce(self.store.grad[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Reduce whatever was not launched from hooks (the 'cond' group, or everything if hooks are unused) and wait."""
        if self.world > 1:
            for name in self.store.groups:
                self.on_group_ready(name)
            for w in self.pending:
                w.wait()
        self.pending, self.launched = [], set()
        return 1.0 / self.world     # multiplier that turns the summed gradient into the DDP average


class FusedAdamW:
    """torch.optim.AdamW semantics (configs/PixArt_xl2_internal.py:48) + clip_grad_norm_ (train.py:182) on the flat buffers,
    as HIP kernels.  step() never synchronises with the host; the gradient norm stays on the device (`.last_norm`)."""

    def __init__(self, model, lr=2e-5, betas=(0.9, 0.999), eps=1e-10, weight_decay=3e-2, max_grad_norm=0.01, reducer=None):
        assert model._store is not None, "call model.prepare(device) (or run one forward) before building the optimizer"
        self.model, self.store = model, model._store
        self.lr, self.betas, self.eps, self.wd, self.max_norm = lr, betas, eps, weight_decay, max_grad_norm
        dev = self.store.device
        self.m = torch.zeros_like(self.store.master)
        self.v = torch.zeros_like(self.store.master)
        self.sumsq = torch.zeros(1, device=dev)
        self.coef = torch.zeros(2, device=dev)   # [multiplier, total_norm]
        self.t = 0
        self.reducer = reducer or GradReducer(self.store)
        model._engine.grad_ready_hook = self.reducer.on_group_ready

    def zero_grad(self, set_to_none=False):
        self.store.attach_grads()
        self.store.grad.zero_()

    @property
    def last_norm(self):
        return self.coef[1]

    def step(self):
        from . import ops
        self.store.attach_grads()            # adopt gradients autograd may have allocated outside the flat buffer
        inv_world = self.reducer.finish()
        self.t += 1
        self.sumsq.zero_()
        ops.sumsq(self.store.grad, self.sumsq)
        ops.clip_coef(self.sumsq, self.coef, self.max_norm if self.max_norm else 0.0, inv_world)
        ops.adamw_step(self.store.master, self.store.grad, self.m, self.v, self.store.shadow, self.lr, self.betas[0], self.betas[1],
                       self.eps, self.wd, self.t, gscale=self.coef)

    def state_dict(self):
        return {"m": self.m, "v": self.v, "t": self.t, "lr": self.lr}

    def load_state_dict(self, sd):
        self.m.copy_(sd["m"])
        self.v.copy_(sd["v"])
        self.t, self.lr = sd["t"], sd.get("lr", self.lr)
