"""GPU parity of the fused CAME step (pxa_came_step through pixart_sigma_amd.dp.FusedCAME) against oracle/came_ref.py, the
restatement of came_pytorch.CAME.step() (parity unpinned: the package is an un-vendored dependency of the reference).
Both sides are fp32; they differ only in summation order and rsqrt rounding -> 1e-6 on the parameters, 1e-4 on the parameter UPDATES of every step."""
from types import SimpleNamespace

import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import rel_l2  # noqa: E402

SHAPES = {"a.weight": (300, 200), "a.bias": (300,), "table": (6, 1152), "conv.weight": (48, 4, 2, 2), "wide.weight": (96, 4096),
          "odd.weight": (33, 7), "tall.weight": (4608, 64), "b.bias": (7,), "huge_row.weight": (8, 9000), "long.bias": (600000,), "big.weight": (1200, 1152)}


@pytest.mark.parametrize("wd,max_norm", [(0.0, 0.01), (0.03, 0.0)])
def test_fused_came_matches_oracle(wd, max_norm):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle.came_ref import CAMERef
    from pixart_sigma_amd import ops
    from pixart_sigma_amd.dp import FusedCAME
    from pixart_sigma_amd.engine import ParamStore
    g = torch.Generator().manual_seed(0)
    params = [(n, torch.nn.Parameter((torch.randn(s, generator=g) * 0.05).cuda())) for n, s in SHAPES.items()]
    store = ParamStore(params, torch.device("cuda"))
    model = SimpleNamespace(_store=store, _engine=SimpleNamespace(grad_ready_hook=None))
    opt = FusedCAME(model, lr=1e-3, weight_decay=wd, max_grad_norm=max_norm)
    ref_p = [p.detach().clone() for _, p in params]
    ref = CAMERef(ref_p, lr=1e-3, weight_decay=wd)
    for step in range(4):
        opt.zero_grad()
        grads = []
        for i, (n, p) in enumerate(params):
            gr = torch.randn(p.shape, generator=g).cuda() * (10.0 ** (i % 3 - 1))
            if n == "b.bias":
                gr.zero_()                                   # a parameter that receives no gradient signal
            p.grad.copy_(gr)
            grads.append(gr)
        total = torch.sqrt(sum((x.double() ** 2).sum() for x in grads)).float()
        coef = torch.clamp(max_norm / (total + 1e-6), max=1.0) if max_norm else torch.tensor(1.0, device="cuda")
        before = [p.detach().clone() for _, p in params]
        before_ref = [p.clone() for p in ref_p]
        opt.step()
        ref.step([x * coef for x in grads])
        assert rel_l2(opt.last_norm, total) < 1e-5
        for (n, p), b, rp, rb in zip(params, before, ref_p, before_ref):
            want = rp - rb
            assert rel_l2(p.detach(), rp) < 1e-6, (step, n)
            if want.abs().max() == 0:
                assert torch.equal(p.detach(), b), n
            elif want.norm() > 1e-3 * rp.norm():             # (a zero-gradient tensor only moves by lr * wd * p: fp32 noise of p - b)
                assert rel_l2(p.detach() - b, want) < 1e-4, (step, n)   # fp32 cancellation in p - b: ulp(p) / |update|
            assert torch.equal(store.view(store.shadow, n), p.detach().to(ops.BF16)), n
    lay = opt.layout["conv.weight"]
    assert lay["row"][1] == 48 * 4 * 2 and lay["col"][1] == 48 * 4 * 2       # [48*4] matrices of 2 x 2
    sd = opt.state_dict()
    opt.load_state_dict({k: (v.clone() if torch.is_tensor(v) else v) for k, v in sd.items()})
