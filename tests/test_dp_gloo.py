"""Data-parallel path on CPU processes (gloo, world_size 2): the bucketed, hook-driven gradient all-reduce of
pixart_sigma_amd.dp.GradReducer over the flat ParamStore buffer must reproduce single-process large-batch gradients
(DDP-average semantics), whatever the order in which buckets complete."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.Tanh(), torch.nn.Linear(32, 8), torch.nn.Tanh(), torch.nn.Linear(8, 4))


def _group_of(name):
    return {"0": "cond", "2": "blocks.0", "4": "final"}[name.split(".")[0]]


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pixart_sigma_amd.dp import GradReducer
    from pixart_sigma_amd.engine import ParamStore
    m = _model()
    store = ParamStore(list(m.named_parameters()), torch.device("cpu"), group_of=_group_of)
    red = GradReducer(store)
    g = torch.Generator().manual_seed(123)
    x, y = torch.randn(8, 16, generator=g), torch.randn(8, 4, generator=g)
    xs, ys = x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4]
    for step in range(2):                      # two steps: the reducer state must reset cleanly
        store.grad.zero_()
        ((m(xs) - ys) ** 2).mean().backward()
        # buckets complete in backward order; fire two of the three hooks "during backward", leave 'cond' to finish()
        red.on_group_ready("final")
        red.on_group_ready("blocks.0")
        inv = red.finish()
        store.grad.mul_(inv)
    if rank == 0:
        torch.save({"grad": store.grad.clone(), "groups": dict(store.groups)}, out)
    dist.destroy_process_group()


def test_bucketed_allreduce_matches_large_batch(tmp_path):
    out = str(tmp_path / "g.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out, weights_only=False)
    from pixart_sigma_amd.engine import ParamStore
    m = _model()
    store = ParamStore(list(m.named_parameters()), torch.device("cpu"), group_of=_group_of)
    g = torch.Generator().manual_seed(123)
    x, y = torch.randn(8, 16, generator=g), torch.randn(8, 4, generator=g)
    ((m(x) - y) ** 2).mean().backward()       # mean over the global batch == average of the two per-rank means
    assert set(got["groups"]) == {"cond", "blocks.0", "final"}
    assert torch.allclose(got["grad"], store.grad, rtol=1e-5, atol=1e-7)


def test_reducer_is_noop_single_process():
    from pixart_sigma_amd.dp import GradReducer
    from pixart_sigma_amd.engine import ParamStore
    m = _model()
    store = ParamStore(list(m.named_parameters()), torch.device("cpu"), group_of=_group_of)
    red = GradReducer(store)
    red.on_group_ready("final")
    assert red.finish() == 1.0 and red.pending == []
