"""Data parallelism with the REAL kernels on two ranks (GPU box, one GPU): two processes share cuda:0 and reduce over gloo (which moves
GPU tensors through host memory) - the closest thing to the multi-GPU job that a one-GPU runner can execute.  Unlike tests/test_dp_gloo.py
(CPU, kernel stand-ins) every gradient here comes out of the HIP forward / backward, the buckets are all-reduced from the engine's hooks
while the remaining backward kernels are still being enqueued, and the fused clip + AdamW consumes the reduced buffer.

Checked, on different per-rank batches:
  * the ranks are seeded DIFFERENTLY and become replicas through model.prepare()'s broadcast of rank 0's parameters (round 6: DDP's wrap-time semantics);
  * both ranks fire the bucket collectives from the hooks in the order final -> blocks.1 -> blocks.0 -> cond (the last one from autograd's end-of-backward callback);
  * the reduced gradient buffer equals the sum of the two batches' gradients accumulated by ONE process into one buffer (per-tensor
    rel-L2; the kernels' fp32 atomics and the different summation order leave ~1e-6) - the DDP contract of the reference's accelerate wrapper
    (train_scripts/train.py:180-184,318-326), with 1 / world folded into the clip coefficient;
  * after two fused clip + AdamW steps the two ranks hold the same weights (Adam turns a gradient that is pure rounding noise - the key
    bias, to which softmax is invariant - into +-lr, so weights are compared between ranks, which see identical reduced gradients, and the
    gradients are compared against the single-process run)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

LR, STEPS = 1e-3, 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model(seed=0, broadcast=None):
    """broadcast=None: prepare() broadcasts rank 0's weights when a process group of > 1 ranks exists (a COLLECTIVE, like wrapping in DDP: every rank calls it);
    False: a local model (the single-process reference rank 0 builds on its own)."""
    from pixart_sigma_amd import PixArtMS
    torch.manual_seed(seed)
    m = PixArtMS(depth=2, input_size=16, model_max_length=16, class_dropout_prob=0.0,
                 kv_compress_config={"sampling": "conv", "scale_factor": 2, "kv_compress_layer": [1]})
    with torch.no_grad():
        for blk in m.blocks:
            blk.cross_attn.proj.weight.normal_(std=0.02)
        m.final_layer.linear.weight.normal_(std=0.02)
    m = m.cuda().train()
    m.prepare("cuda", broadcast=broadcast)
    return m


def _batch(step, rank):
    g = torch.Generator().manual_seed(1000 + 10 * step + rank)
    B, L = 2, 16
    x0 = torch.randn(B, 4, 16, 16, generator=g).cuda()
    noise = torch.randn(B, 4, 16, 16, generator=g).cuda()
    y = torch.randn(B, 1, L, 4096, generator=g).cuda()
    t = torch.randint(0, 1000, (B,), generator=g).cuda()
    mask = torch.ones(B, L, dtype=torch.int64)
    mask[1, 9:] = 0                                        # a ragged caption
    return x0, noise, y, t, mask


def _loss(diff, model, batch):
    x0, noise, y, t, mask = batch
    return diff.training_losses(model, x0, t, model_kwargs=dict(y=y, mask=mask, data_info=None), noise=noise)["loss"].mean()


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from pixart_sigma_amd import IDDPM
    from pixart_sigma_amd.dp import FusedAdamW, GradReducer
    diff = IDDPM("1000", learn_sigma=True, pred_sigma=True, snr=False)
    model = _model(seed=100 + rank)                        # ranks initialise DIFFERENTLY: prepare()'s broadcast of rank 0's weights is what makes them replicas
    store = model._store
    red = GradReducer(store)
    assert red.active and red.world == world
    model._engine.grad_ready_hook = red.on_group_ready
    # ---- phase A: one backward, reduced gradients vs the single-process sum
    _loss(diff, model, _batch(7, rank)).backward()
    assert model._store is store                            # prepare("cuda") and the forward's "cuda:0" are the same store (no rebuild)
    order_a = list(red.launched)
    inv = red.finish()
    g_dp = store.grad.detach().clone()
    res = None
    if rank == 0:
        ref = _model(seed=100, broadcast=False)             # rank 0's own initialisation = what every rank holds after the broadcast
        assert torch.equal(ref._store.master, store.master)
        for r in range(world):
            _loss(diff, ref, _batch(7, r)).backward()
        torch.cuda.synchronize()
        worst = (0.0, "")
        for n in store.names:
            a, b = store.view(g_dp, n).float(), ref._store.g(n).float()
            den = float(b.norm())
            if den > 1e-7:                                  # tensors whose gradient is rounding noise (k bias) have no relative scale
                worst = max(worst, (float((a - b).norm()) / den, n))
        res = {"order_a": order_a, "inv": inv, "grad_worst": worst, "grad_rel": float((g_dp - ref._store.grad).norm() / ref._store.grad.norm())}
    # ---- phase B: two optimizer steps, ranks must agree
    store.grad.zero_()
    opt = FusedAdamW(model, lr=LR, weight_decay=3e-2, eps=1e-10, max_grad_norm=0.01, reducer=red)
    before = store.master.detach().clone()
    orders, losses = [], []
    for step in range(STEPS):
        opt.zero_grad()
        loss = _loss(diff, model, _batch(step, rank))
        loss.backward()
        orders.append(list(red.launched))                   # what the hooks launched during backward, before finish()
        opt.step()
        losses.append(float(loss.detach()))
    torch.cuda.synchronize()
    mine = store.master.detach().cpu()
    both = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(both, mine)
    if rank == 0:
        res.update({"orders": orders, "losses": losses, "ranks_diff": float((both[0] - both[1]).abs().max()),
                    "moved": float((both[0] - before.cpu()).abs().max())})
        torch.save(res, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_real_kernels_match_single_process(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    out = str(tmp_path / "r.pt")
    here = os.path.dirname(os.path.abspath(__file__))
    os.environ["PYTHONPATH"] = here + os.pathsep + os.path.dirname(here) + os.pathsep + os.environ.get("PYTHONPATH", "")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    res = torch.load(out, weights_only=False)
    print(f"two-rank DP with the HIP kernels: reduced gradients vs single-process sum rel-L2 {res['grad_rel']:.2e} (worst tensor "
          f"{res['grad_worst'][1]} {res['grad_worst'][0]:.2e}); after {STEPS} steps |rank0 - rank1| {res['ranks_diff']:.2e}, weights moved by "
          f"{res['moved']:.2e}; losses {res['losses']}")
    want = ["final", "blocks.1", "blocks.0", "cond"]        # 'cond' from the autograd end-of-backward callback (round 6), i.e. before finish()
    assert res["order_a"] == want and all(o == want for o in res["orders"]), (res["order_a"], res["orders"])
    assert res["inv"] == 0.5
    assert res["grad_rel"] < 1e-5 and res["grad_worst"][0] < 1e-4, res["grad_worst"]
    assert all(torch.isfinite(torch.tensor(res["losses"])))
    assert res["moved"] > 0.5 * LR                           # the optimizer really stepped
    assert res["ranks_diff"] <= STEPS * LR * 1e-4            # identical reduced gradients; the norm's atomic order is the only difference
