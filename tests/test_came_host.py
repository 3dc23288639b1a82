"""CPU check of oracle/came_ref.py (the checker of the fused CAME step; parity unpinned: came_pytorch is not installed and the reference has no test for it)
against a SECOND, structurally different restatement: scalar Python loops over the PUBLISHED algorithm (CAME, Luo et al., ACL 2023, Algorithm 2) in the paper's
own form - row / column SUMS and v = r c^T / (1^T r) - where came_pytorch (and oracle/came_ref.py after it) keeps row / column MEANS and multiplies
(r / mean(r))^-1/2 by c^-1/2.  The two forms are algebraically equal; a slip of an axis, of a mean against a sum or of the clipping's RMS in either shows as a
mismatch.  Non-factored (1-D) tensors follow the package (plain second moment, no confidence statistics)."""
import math

import torch

from oracle.came_ref import CAMERef

LR, EPS1, EPS2, D, B1, B2, B3 = 1e-2, 1e-30, 1e-16, 1.0, 0.9, 0.999, 0.9999


def _paper_step_matrix(theta, G, st):
    """One step of Algorithm 2 on an n x m matrix, scalar arithmetic in double precision; st = dict(r, c, m, R, C) (lists), updated in place."""
    n, m = len(G), len(G[0])
    sq = [[G[i][j] ** 2 + EPS1 for j in range(m)] for i in range(n)]
    st["r"] = [B2 * st["r"][i] + (1 - B2) * sum(sq[i]) for i in range(n)]                               # r_t = beta2 r + (1 - beta2) (G^2 + eps1) 1_m
    st["c"] = [B2 * st["c"][j] + (1 - B2) * sum(sq[i][j] for i in range(n)) for j in range(m)]          # c_t = beta2 c + (1 - beta2) 1_n^T (G^2 + eps1)
    tot = sum(st["r"])
    u = [[G[i][j] / math.sqrt(st["r"][i] * st["c"][j] / tot) for j in range(m)] for i in range(n)]      # u = G / sqrt(v), v = r c^T / (1^T r)
    rms = math.sqrt(sum(x * x for row in u for x in row) / (n * m))
    u = [[x / max(1.0, rms / D) for x in row] for row in u]                                              # u^ = u / max(1, RMS(u) / d)
    st["m"] = [[B1 * st["m"][i][j] + (1 - B1) * u[i][j] for j in range(m)] for i in range(n)]
    U = [[(u[i][j] - st["m"][i][j]) ** 2 + EPS2 for j in range(m)] for i in range(n)]                   # instability of the update
    st["R"] = [B3 * st["R"][i] + (1 - B3) * sum(U[i]) for i in range(n)]
    st["C"] = [B3 * st["C"][j] + (1 - B3) * sum(U[i][j] for i in range(n)) for j in range(m)]
    totR = sum(st["R"])
    return [[theta[i][j] - LR * st["m"][i][j] / math.sqrt(st["R"][i] * st["C"][j] / totR) for j in range(m)] for i in range(n)]


def _package_step_vector(theta, g, st):
    """came_pytorch on a 1-D tensor: second moment per element, clipping, first moment; the update IS the first moment."""
    n = len(g)
    st["v"] = [B2 * st["v"][i] + (1 - B2) * (g[i] ** 2 + EPS1) for i in range(n)]
    u = [g[i] / math.sqrt(st["v"][i]) for i in range(n)]
    rms = math.sqrt(sum(x * x for x in u) / n)
    u = [x / max(1.0, rms / D) for x in u]
    st["m"] = [B1 * st["m"][i] + (1 - B1) * u[i] for i in range(n)]
    return [theta[i] - LR * st["m"][i] for i in range(n)]


def test_came_restatement_matches_the_published_algorithm_in_its_own_form():
    g = torch.Generator().manual_seed(0)
    n, m = 3, 5
    W = torch.randn(n, m, generator=g, dtype=torch.float64)
    b = torch.randn(7, generator=g, dtype=torch.float64)
    grads = [(torch.randn(n, m, generator=g, dtype=torch.float64) * s, torch.randn(7, generator=g, dtype=torch.float64) * s) for s in (1.0, 0.1, 3.0, 1e-3)]
    pw, pb = W.clone().float(), b.clone().float()
    ref = CAMERef([pw, pb], lr=LR, eps=(EPS1, EPS2), clip_threshold=D, betas=(B1, B2, B3), weight_decay=0.0)
    tw, tb = W.tolist(), b.tolist()
    sw = dict(r=[0.0] * n, c=[0.0] * m, m=[[0.0] * m for _ in range(n)], R=[0.0] * n, C=[0.0] * m)
    sb = dict(v=[0.0] * 7, m=[0.0] * 7)
    for gw, gb in grads:
        ref.step(grads=[gw.float(), gb.float()])
        tw = _paper_step_matrix(tw, gw.tolist(), sw)
        tb = _package_step_vector(tb, gb.tolist(), sb)
        assert torch.allclose(pw.double(), torch.tensor(tw, dtype=torch.float64), rtol=2e-5, atol=2e-7), (pw, tw)
        assert torch.allclose(pb.double(), torch.tensor(tb, dtype=torch.float64), rtol=2e-5, atol=2e-7)
    assert (pw.double() - W).abs().max() > 1e-3                     # the parameters really moved


def test_came_restatement_batched_matrices_factor_the_last_two_dims():
    """came_pytorch factors over the LAST TWO dims of a >= 2-D tensor (a conv weight (Co, Ci, kh, kw) is Co x Ci matrices of kh x kw): each trailing matrix
    follows the 2-D algorithm on its own statistics, but the clipping RMS is taken over the WHOLE tensor."""
    g = torch.Generator().manual_seed(1)
    T = torch.randn(2, 3, 4, generator=g, dtype=torch.float64)
    G = torch.randn(2, 3, 4, generator=g, dtype=torch.float64)
    p = T.clone().float()
    CAMERef([p], lr=LR, eps=(EPS1, EPS2), clip_threshold=D, betas=(B1, B2, B3)).step(grads=[G.float()])
    # by hand for the first step: r, c from zero state; u = G / sqrt(v); ONE rms over all 24 elements
    us = []
    for k in range(2):
        Gk = G[k].tolist()
        sq = [[x * x + EPS1 for x in row] for row in Gk]
        r = [(1 - B2) * sum(row) for row in sq]
        c = [(1 - B2) * sum(sq[i][j] for i in range(3)) for j in range(4)]
        us.append([[Gk[i][j] / math.sqrt(r[i] * c[j] / sum(r)) for j in range(4)] for i in range(3)])
    rms = math.sqrt(sum(x * x for u in us for row in u for x in row) / 24)
    out = []
    for k in range(2):
        u = [[x / max(1.0, rms / D) for x in row] for row in us[k]]
        mom = [[(1 - B1) * x for x in row] for row in u]
        U = [[(u[i][j] - mom[i][j]) ** 2 + EPS2 for j in range(4)] for i in range(3)]
        R = [(1 - B3) * sum(row) for row in U]
        C = [(1 - B3) * sum(U[i][j] for i in range(3)) for j in range(4)]
        out.append([[T[k][i][j].item() - LR * mom[i][j] / math.sqrt(R[i] * C[j] / sum(R)) for j in range(4)] for i in range(3)])
    assert torch.allclose(p.double(), torch.tensor(out, dtype=torch.float64), rtol=2e-5, atol=2e-7)
