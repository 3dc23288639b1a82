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
import os
import sys
import time

import torch
import torch.distributed as dist

# Process-wide count of ACTIVE reducers: the persistent GEMMs' item hand-out is one switch for the whole process, so it is turned to the dynamic
# cursors when the first active reducer is built and put back to what was found when the last one is closed - whatever order reducers are created / collected in.
_ACTIVE_REDUCERS = {"n": 0, "prev": None}


def _pg_world(group=None):
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def broadcast_parameters(model, src=0, group=None):
    """DDP's wrap-time semantics (what `accelerator.prepare(model)` does for the reference, train_scripts/train.py:486): every rank takes rank `src`'s
    parameters and buffers.  The flat fp32 master buffer goes as ONE broadcast (2.4 GB for XL/2), then the module buffers (y_embedding, pos_embed); the 16-bit
    shadow and everything derived from the weights are re-cast afterwards.  No-op without a process group."""
    if _pg_world(group) <= 1 and os.environ.get("PXA_DP_FORCE_COLLECTIVES") != "1":
        return False
    if not (dist.is_available() and dist.is_initialized()):
        return False
    st = model._store
    gsrc = dist.get_global_rank(group, src) if group is not None else src
    dist.broadcast(st.master, src=gsrc, group=group)
    for b in model.buffers():
        dist.broadcast(b, src=gsrc, group=group)
    st.refresh_shadow(force=True)
    return True


def check_replicas(store, group=None):
    """Replica-equality check (cheap: two doubles per rank): sum and sum of squares of the flat master buffer must agree on every rank.  Identical seeds used to be
    ASSUMED; a rank that loaded another checkpoint, or skipped prepare()'s broadcast, diverges silently otherwise."""
    world = _pg_world(group)
    if world <= 1:
        return True
    m = store.master.double()
    mine = torch.stack([m.sum(), (m * m).sum()])
    if dist.get_backend(group) == "gloo":          # (gloo moves GPU tensors through the host for broadcast / all-reduce only)
        mine = mine.cpu()
    every = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(every, mine, group=group)
    if not all(torch.equal(e, every[0]) for e in every):
        raise RuntimeError("data-parallel replicas hold different parameters: " + ", ".join(f"rank {i}: sum {e[0].item():.9e}" for i, e in enumerate(every))
                           + " - call model.prepare(device) after init_process_group (it broadcasts rank 0's weights) or load the same checkpoint everywhere")
    return True


class GradReducer:
    """Bucketed, overlapped gradient all-reduce over the flat gradient buffer.  Backend-agnostic (nccl on GPUs, gloo in the
    CPU tests).

    Gradient accumulation (the reference's `accelerator.accumulate(model)` / gradient_accumulation_steps, train.py:170,486): wrap the
    non-final micro-steps in `no_sync()` - the engine's hooks are then ignored, gradients keep accumulating locally in the flat
    buffer, and the buckets are reduced from the hooks of the LAST backward (or all of them in finish()).  A hook that fires for a
    bucket already in flight in the same step would add local gradients on top of a reduced buffer: that is an error, not a no-op.

    bucket_dtype=torch.bfloat16 (1.22 GB instead of 2.44 GB per step on the wire): a bucket is cast into a bf16
    staging buffer, reduced there, and written back to the fp32 gradients in finish() (staging = one extra copy of the gradients
    in bf16, 1.22 GB of 288 GB)."""

    def __init__(self, store, group=None, bucket_dtype=None):
        self.store = store
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        # PXA_DP_FORCE_COLLECTIVES=1: run the bucket collectives even in a one-rank group (a real RCCL all-reduce of every bucket from
        # the engine's hooks on a single GPU - the only multi-GPU code path a one-GPU box can execute; tests/test_training_runtime_gpu.py::test_bench_under_torchrun_forced_collectives)
        self.active = self.world > 1 or (dist.is_available() and dist.is_initialized() and os.environ.get("PXA_DP_FORCE_COLLECTIVES") == "1")
        self._counted = False
        if self.active:
            # bucket all-reduces will run beside the backward's GEMMs: their persistent kernels hand items out dynamically, so that a workgroup kept off its CU
            # by a collective does not double the kernel's time (include/pixart_hip.h: pxa_gemm_set_dynamic_items; one GPU alone keeps the faster static split)
            # The switch is process-wide and reference-counted over the active reducers (see _ACTIVE_REDUCERS): 0 -> 1 turns the cursors on and remembers what
            # was there, 1 -> 0 puts it back.  PXA_DP_STATIC_ITEMS=1 keeps the static split under collectives (A/B of the cursors' standing cost).
            if os.environ.get("PXA_DP_STATIC_ITEMS") != "1":
                from . import lib as _lib
                if _ACTIVE_REDUCERS["n"] == 0:
                    _ACTIVE_REDUCERS["prev"] = _lib.load().pxa_gemm_set_dynamic_items(1)
                _ACTIVE_REDUCERS["n"] += 1
                self._counted = True
        # PXA_DP_TRACE=1: per bucket, when its gradients were complete (all-reduce launched) and when the compute stream got past its wait; plus the end of
        # backward - the first SCALE run then explains its own exposed-communication time (bench.py prints it).  Events on GPUs, perf_counter on CPU (gloo).
        self.trace_on = os.environ.get("PXA_DP_TRACE") == "1"
        self.last_trace = None
        self._tr = []
        self.bucket_dtype = bucket_dtype
        self.stage = torch.empty(store.total, dtype=bucket_dtype, device=store.device) if bucket_dtype not in (None, torch.float32) and self.active else None
        self.pending = []
        self.launched = []          # bucket names in launch order (the order every rank must agree on)
        self._sync = True

    def close(self):
        """Tear-down: the last active reducer to close restores the GEMM item hand-out the first one found (idempotent; also run when the reducer is collected)."""
        if getattr(self, "_counted", False):
            self._counted = False
            _ACTIVE_REDUCERS["n"] -= 1
            if _ACTIVE_REDUCERS["n"] == 0 and _ACTIVE_REDUCERS["prev"] is not None:
                from . import lib as _lib
                _lib.load().pxa_gemm_set_dynamic_items(_ACTIVE_REDUCERS["prev"])
                _ACTIVE_REDUCERS["prev"] = None

    def _stamp(self):
        if torch.device(self.store.device).type == "cuda":
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            return ev
        return time.perf_counter()

    def __del__(self):
        try:
            self.close()
        except Exception:      # interpreter shutdown: the library may already be gone
            pass

    class _NoSync:
        def __init__(self, r):
            self.r = r

        def __enter__(self):
            self.prev, self.r._sync = self.r._sync, False

        def __exit__(self, *exc):
            self.r._sync = self.prev

    def no_sync(self):
        """Context manager for the non-final micro-steps of gradient accumulation (DDP.no_sync semantics)."""
        return self._NoSync(self)

    def on_group_ready(self, name):
        """Engine hook: gradients of parameter group `name` are final for this backward."""
        if not self.active or not self._sync or name not in self.store.groups:
            return
        if name in self.launched:
            raise RuntimeError(f"gradient bucket {name!r} was completed twice before optimizer.step(): wrap the non-final micro-steps "
                               "of gradient accumulation in reducer.no_sync()")
        s, e = self.store.groups[name]
        self.launched.append(name)
        if self.trace_on:
            self._tr.append(dict(bucket=name, bytes=(e - s) * (2 if self.stage is not None else 4), ready=self._stamp()))
        if self.stage is not None:
            buf = self.stage[s:e]
            buf.copy_(self.store.grad[s:e])
        else:
            buf = self.store.grad[s:e]
        self.pending.append((dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True), s, e))

    def finish(self):
        """Reduce whatever was not launched from hooks (the 'cond' group, or everything if hooks are unused) and wait."""
        if self.active:
            launched = set(self.launched)
            late = [name for name in self.store.groups if name not in launched]
            for name in late:
                self.on_group_ready(name)
            t_bwd = self._stamp() if self.trace_on else None          # everything the backward enqueued is in front of this point of the compute stream
            for i, (w, s, e) in enumerate(self.pending):
                w.wait()
                if self.stage is not None:
                    self.store.grad[s:e].copy_(self.stage[s:e])
                if self.trace_on:
                    self._tr[i]["done"] = self._stamp()
            if self.trace_on:
                self._finish_trace(t_bwd, late)
        self.pending, self.launched = [], []
        return 1.0 / self.world     # multiplier that turns the summed gradient into the DDP average

    def _finish_trace(self, t_bwd, late):
        """Resolve the step's stamps (one host sync - tracing only) into milliseconds relative to the first bucket's launch."""
        tr, self._tr = self._tr, []
        if not tr:
            return
        if not isinstance(t_bwd, float):
            torch.cuda.synchronize()
            t0 = tr[0]["ready"]
            ms = lambda ev: t0.elapsed_time(ev)          # noqa: E731
        else:
            t0 = tr[0]["ready"]
            ms = lambda t: (t - t0) * 1e3                # noqa: E731
        rows = [dict(bucket=r["bucket"], MB=round(r["bytes"] / 1e6, 1), ready_ms=round(ms(r["ready"]), 3), passed_ms=round(ms(r["done"]), 3),
                     launched_from="finish()" if r["bucket"] in late else "hook") for r in tr]
        bwd = ms(t_bwd)
        self.last_trace = dict(backward_done_ms=round(bwd, 3), all_reduced_ms=rows[-1]["passed_ms"], exposed_ms=round(max(0.0, rows[-1]["passed_ms"] - bwd), 3),
                               bytes=sum(r["bytes"] for r in tr), world=self.world, buckets=rows)
        if os.environ.get("PXA_DP_TRACE_PRINT", "1") == "1" and (not dist.is_initialized() or dist.get_rank() == 0):
            t = self.last_trace
            print(f"[dp trace] backward done {t['backward_done_ms']:.2f} ms after the first bucket; last bucket passed at {t['all_reduced_ms']:.2f} ms; "
                  f"exposed communication {t['exposed_ms']:.2f} ms; {t['bytes'] / 1e9:.2f} GB in {len(rows)} buckets, world {self.world}", file=sys.stderr)


class LossScaler:
    """Dynamic loss scaling for fp16-operand training - torch.cuda.amp.GradScaler's protocol (what accelerate runs for the reference's
    mixed_precision='fp16', configs/PixArt_xl2_internal.py:57, train_scripts/train.py:180-184) kept entirely on the device:

        loss = scaler.scale(terms['loss'].mean());  loss.backward();  opt.step()          # opt built with scaler=...

    `scale()` multiplies by the current device-side scale; the fused optimizers fold 1/scale into the clip coefficient, skip the update
    when the gradient norm is inf/nan and apply GradScaler.update()'s growth / backoff in the same tiny kernel (pxa_clip_coef_scaled).
    Nothing synchronises with the host; `found_inf`, `value`, `skipped` read the record back on demand (logging only)."""

    def __init__(self, device, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
        self.state = torch.tensor([init_scale, 0.0, 0.0, 0.0, 0.0], dtype=torch.float32, device=device)
        self.growth_factor, self.backoff_factor, self.growth_interval = growth_factor, backoff_factor, growth_interval

    def scale(self, loss):
        return loss * self.state[0]

    @property
    def value(self):
        return float(self.state[0].item())

    @property
    def found_inf(self):
        return bool(self.state[2].item() != 0)

    @property
    def steps_applied(self):
        return int(self.state[3].item())

    @property
    def steps_skipped(self):
        return int(self.state[4].item())

    def state_dict(self):
        return {"state": self.state.clone(), "growth_factor": self.growth_factor, "backoff_factor": self.backoff_factor,
                "growth_interval": self.growth_interval}

    def load_state_dict(self, sd):
        self.state.copy_(sd["state"])
        self.growth_factor, self.backoff_factor, self.growth_interval = sd["growth_factor"], sd["backoff_factor"], sd["growth_interval"]


class FusedAdamW:
    """torch.optim.AdamW semantics (configs/PixArt_xl2_internal.py:48) + clip_grad_norm_ (train.py:182) on the flat buffers,
    as HIP kernels.  step() never synchronises with the host; the gradient norm stays on the device (`.last_norm`)."""

    def __init__(self, model, lr=2e-5, betas=(0.9, 0.999), eps=1e-10, weight_decay=3e-2, max_grad_norm=0.01, reducer=None, scaler=None):
        assert model._store is not None, "call model.prepare(device) (or run one forward) before building the optimizer"
        self.model, self.store = model, model._store
        self.lr, self.betas, self.eps, self.wd, self.max_norm = lr, betas, eps, weight_decay, max_grad_norm
        self.scaler = scaler
        dev = self.store.device
        self.m = torch.zeros_like(self.store.master)
        self.v = torch.zeros_like(self.store.master)
        self.sumsq = torch.zeros(1, device=dev)
        self.coef = torch.zeros(2, device=dev)   # [multiplier, total_norm]
        self.t = 0
        self.reducer = reducer or GradReducer(self.store)
        model._engine.grad_ready_hook = self.reducer.on_group_ready
        if self.reducer.world > 1:
            check_replicas(self.store, self.reducer.group)

    def zero_grad(self, set_to_none=False):
        self.store.attach_grads()
        self.store.grad.zero_()

    @property
    def last_norm(self):
        return self.coef[1]

    def _check_store(self):
        if self.model._store is not self.store:
            raise RuntimeError("the model rebuilt its flat parameter store after this optimizer was created (model moved to another device, or a "
                               "parameter's storage was replaced): the optimizer would update orphaned buffers - build the optimizer after the move")

    def step(self):
        from . import ops
        self._check_store()
        self.store.attach_grads()            # adopt gradients autograd may have allocated outside the flat buffer
        inv_world = self.reducer.finish()
        self.t += 1
        self._clip(inv_world)
        if self.scaler is not None:       # loss-scaled step: skipped on inf/nan, bias correction from the device-side applied-step count
            ops.adamw_step_scaled(self.store.master, self.store.grad, self.m, self.v, self.store.shadow, self.lr, self.betas[0], self.betas[1],
                                  self.eps, self.wd, self.coef, self.scaler.state)
        else:
            ops.adamw_step(self.store.master, self.store.grad, self.m, self.v, self.store.shadow, self.lr, self.betas[0], self.betas[1],
                           self.eps, self.wd, self.t, gscale=self.coef)
        self.store.bump()                    # the kernel rewrote the shadow weights: caches keyed on them (Engine._text_cache) are stale, derived buffers are rewritten

    def _clip(self, inv_world):
        """sumsq -> device-side clip coefficient (x 1/world, x 1/loss-scale) in self.coef = [multiplier, total_norm]."""
        from . import ops
        self.sumsq.zero_()
        ops.sumsq(self.store.grad, self.sumsq)
        mx = self.max_norm if self.max_norm else 0.0
        if self.scaler is not None:
            sc = self.scaler
            ops.clip_coef_scaled(self.sumsq, self.coef, mx, inv_world, sc.state, sc.growth_factor, sc.backoff_factor, sc.growth_interval)
        else:
            ops.clip_coef(self.sumsq, self.coef, mx, inv_world)

    def _layout(self):
        """(name, offset, shape) of every parameter in the flat buffers: saved with the state and checked on load, so a changed
        parameter order / alignment cannot silently shift the moments onto other parameters."""
        st = self.store
        return [(n, st.offset[n], tuple(st.shape[n])) for n in st.names]

    def _check_layout(self, sd):
        if "layout" in sd and [tuple(x) if not isinstance(x, tuple) else x for x in sd["layout"]] != self._layout():
            raise RuntimeError("optimizer state was saved for a different flat parameter layout (parameter set, order or alignment changed)")

    def state_dict(self):
        return {"m": self.m, "v": self.v, "t": self.t, "lr": self.lr, "layout": self._layout()}

    def load_state_dict(self, sd):
        self._check_layout(sd)
        self.m.copy_(sd["m"])
        self.v.copy_(sd["v"])
        self.t, self.lr = sd["t"], sd.get("lr", self.lr)
        if self.scaler is not None:
            # the loss-scaled step takes its bias-correction count from the scaler's device record (applied steps): seed it from this state, so warmed-up
            # moments resumed without a 'loss_scaler' entry (a bf16 run's checkpoint) are not corrected as if t = 1.  A scaler state loaded AFTERWARDS
            # (train.py's order) overwrites it with the checkpoint's own count.
            self.scaler.state[3] = float(self.t)


def came_tables(names, offset, shape, numel, tile_elems):
    """Host-side tables of pxa_came_step (include/pixart_hip.h): per tensor its [batch][R][C] view and the offsets of its row /
    column / row-mean / full second-moment state; the tile list (whole rows of one tensor, ~tile_elems elements each; 1-D tensors:
    element ranges); 1/R per column-state entry.  Pure Python (tested on CPU)."""
    tensors, tiles, inv_r, layout = [], [], [], {}
    n_row = n_col = n_rm = n_nf = 0
    for ti, name in enumerate(names):
        shp, n = shape[name], numel[name]
        if len(shp) >= 2:                                 # came_pytorch: factored second moments over the last two dims
            R, Cc = shp[-2], shp[-1]
            batch = n // (R * Cc)
            pad = -(batch * Cc) % 4                       # keep every tensor's column state 16-byte aligned
            tensors.append(dict(off=offset[name], batch=batch, R=R, C=Cc, factored=1, row_off=n_row, col_off=n_col, rm_off=n_rm, nf_off=0))
            layout[name] = dict(factored=True, row=(n_row, batch * R), col=(n_col, batch * Cc))
            inv_r += [1.0 / R] * (batch * Cc) + [0.0] * pad
            rows = batch * R
            per = max(4, tile_elems // Cc // 4 * 4)
            if batch > 1 or Cc % 4 or Cc > 256 * 18:      # the kernel's scalar path (came.hip, MAXJ == 0): one wave walks its rows one after the other,
                per = min(per, 64)                        # every row a chain of dependent loads and atomics - the patch-embed conv weight, viewed as
                                                          # [4608][2][2], was ONE tile of 9,216 such rows = ~2 ms per pass, the whole launch waiting for it
            tiles += [(ti, r0, min(per, rows - r0)) for r0 in range(0, rows, per)]
            n_row, n_col, n_rm = n_row + rows, n_col + batch * Cc + pad, n_rm + batch
        else:
            tensors.append(dict(off=offset[name], batch=1, R=1, C=n, factored=0, row_off=0, col_off=0, rm_off=0, nf_off=n_nf))
            layout[name] = dict(factored=False, nf=(n_nf, n))
            tiles += [(ti, e0, min(tile_elems, n - e0)) for e0 in range(0, n, tile_elems)]
            n_nf += n
    return dict(tensors=tensors, tiles=tiles, col_inv_r=inv_r, layout=layout, n_row=n_row, n_col=n_col, n_rm=n_rm, n_nf=n_nf)


class FusedCAME(FusedAdamW):
    """came_pytorch.CAME semantics (the reference's CAMEWrapper, diffusion/utils/optimizer.py:242-246; config defaults of
    configs/pixart_sigma_config/*.py: lr 2e-5, weight_decay 0, betas (0.9, 0.999, 0.9999), eps (1e-30, 1e-16)) + clip_grad_norm_ on
    the flat buffers: one pxa_came_step call per step (6 launches over all tensors).  Memory: exp_avg (fp32, one per parameter) +
    row / column statistics - the factored second moments are 0.1 % of AdamW's `v`."""
    TILE_ELEMS = 262144      # per workgroup: every tile ends in C same-address atomics on the column partials, so tiles are large

    def __init__(self, model, lr=2e-5, betas=(0.9, 0.999, 0.9999), eps=(1e-30, 1e-16), clip_threshold=1.0, weight_decay=0.0,
                 max_grad_norm=0.01, reducer=None, scaler=None):
        from .lib import CameTensor, CameTile
        import ctypes as C
        assert model._store is not None, "call model.prepare(device) (or run one forward) before building the optimizer"
        self.model, self.store = model, model._store
        self.lr, self.betas, self.eps, self.clip, self.wd, self.max_norm = lr, betas, eps, clip_threshold, weight_decay, max_grad_norm
        self.scaler = scaler
        st, dev = self.store, self.store.device
        tb = came_tables(st.names, st.offset, st.shape, st.numel, self.TILE_ELEMS)
        self.layout = tb["layout"]                        # name -> dict(kind, offsets): for state_dict / tests
        tiles, inv_r, n_row, n_col, n_rm, n_nf = tb["tiles"], tb["col_inv_r"], tb["n_row"], tb["n_col"], tb["n_rm"], tb["n_nf"]
        tensors = []
        for d in tb["tensors"]:
            t = CameTensor()
            for k, v in d.items():
                setattr(t, k, v)
            tensors.append(t)
        self.n_col, self.n_rm, self.n_tensors, self.n_tiles = n_col, n_rm, len(tensors), len(tiles)
        tarr = (CameTensor * len(tensors))(*tensors)
        larr = (CameTile * len(tiles))(*[CameTile(a, b, c, 0) for a, b, c in tiles])
        self.tensors_dev = torch.frombuffer(bytearray(C.string_at(C.addressof(tarr), C.sizeof(tarr))), dtype=torch.uint8).to(dev)
        self.tiles_dev = torch.frombuffer(bytearray(C.string_at(C.addressof(larr), C.sizeof(larr))), dtype=torch.uint8).to(dev)
        self.col_inv_r = torch.tensor(inv_r if inv_r else [0.0], dtype=torch.float32, device=dev)
        z = lambda n: torch.zeros(max(n, 1), dtype=torch.float32, device=dev)
        self.m = torch.zeros_like(st.master)
        self.sq_row, self.res_row, self.sq_col, self.res_col, self.nf_sq = z(n_row), z(n_row), z(n_col), z(n_col), z(n_nf)
        from . import lib
        self.scratch = z(lib.load().pxa_came_scratch_elems(n_col, n_rm, len(tensors)))
        self.sumsq = torch.zeros(1, device=dev)
        self.coef = torch.zeros(2, device=dev)
        self.t = 0
        self.reducer = reducer or GradReducer(self.store)
        model._engine.grad_ready_hook = self.reducer.on_group_ready
        if self.reducer.world > 1:
            check_replicas(self.store, self.reducer.group)

    def step(self):
        from . import ops
        from .lib import CameArgs, call, ptr
        self._check_store()
        self.store.attach_grads()
        inv_world = self.reducer.finish()
        self.t += 1
        self._clip(inv_world)
        a = CameArgs()
        a.p, a.g, a.exp_avg, a.p_bf16 = ptr(self.store.master), ptr(self.store.grad), ptr(self.m), ptr(self.store.shadow)
        a.sq_row, a.sq_col, a.res_row, a.res_col, a.nf_sq = ptr(self.sq_row), ptr(self.sq_col), ptr(self.res_row), ptr(self.res_col), ptr(self.nf_sq)
        a.scratch, a.tensors, a.n_tensors, a.tiles, a.n_tiles = ptr(self.scratch), ptr(self.tensors_dev), self.n_tensors, ptr(self.tiles_dev), self.n_tiles
        a.col_inv_r, a.n_cols_total, a.n_rm_total = ptr(self.col_inv_r), self.n_col, self.n_rm
        a.lr, a.beta1, a.beta2, a.beta3 = self.lr, self.betas[0], self.betas[1], self.betas[2]
        a.eps0, a.eps1, a.clip_threshold, a.weight_decay = self.eps[0], self.eps[1], self.clip, self.wd
        a.gscale = ptr(self.coef)
        a.scaler = ptr(self.scaler.state) if self.scaler is not None else None
        call("pxa_came_step", a)                        # refreshes the bf16 shadow itself (no parameter version is bumped)
        self.store.bump()

    def state_dict(self):
        return {k: getattr(self, k) for k in ("m", "sq_row", "sq_col", "res_row", "res_col", "nf_sq")} | {"t": self.t, "lr": self.lr, "layout": self._layout()}

    def load_state_dict(self, sd):
        self._check_layout(sd)
        for k in ("m", "sq_row", "sq_col", "res_row", "res_col", "nf_sq"):
            getattr(self, k).copy_(sd[k])
        self.t, self.lr = sd["t"], sd.get("lr", self.lr)
