"""TEST INFRASTRUCTURE ONLY - plain-torch restatement of `came_pytorch.CAME.step()`.

**Parity unpinned.**  The reference's optimizer for every PixArt-Sigma config is `CAMEWrapper`
(`configs/pixart_sigma_config/PixArt_sigma_xl2_img1024_internalms.py:29`: lr 2e-5, weight_decay 0, betas (0.9, 0.999, 0.9999),
eps (1e-30, 1e-16)), a subclass that adds nothing to `came_pytorch.CAME` (`diffusion/utils/optimizer.py:15,242-246`).
came_pytorch is an unpinned pip dependency (`requirements.txt`), not vendored, not installed here, and the reference has no test
for it; this file restates the algorithm of the package's `step()` (CAME, Luo et al., ACL 2023, Algorithm 2; defaults
clip_threshold = 1.0) so that the HIP path has a checker.  Only tests/ may import it.
"""
import torch


class CAMERef:
    def __init__(self, params, lr=2e-5, eps=(1e-30, 1e-16), clip_threshold=1.0, betas=(0.9, 0.999, 0.9999), weight_decay=0.0):
        self.params = list(params)
        self.lr, self.eps, self.clip, self.betas, self.wd = lr, eps, clip_threshold, betas, weight_decay
        self.state = [dict() for _ in self.params]

    @staticmethod
    def _rms(t):
        return t.norm(2) / (t.numel() ** 0.5)

    @staticmethod
    def _approx_sq_grad(row, col):
        r_factor = (row / row.mean(dim=-1, keepdim=True)).rsqrt_().unsqueeze(-1)
        c_factor = col.unsqueeze(-2).rsqrt()
        return torch.mul(r_factor, c_factor)

    @torch.no_grad()
    def step(self, grads=None):
        for i, p in enumerate(self.params):
            grad = (p.grad if grads is None else grads[i])
            if grad is None:
                continue
            grad = grad.float()
            st, shape = self.state[i], grad.shape
            factored = len(shape) >= 2
            if not st:
                st["exp_avg"] = torch.zeros_like(grad)
                if factored:
                    st["exp_avg_sq_row"] = torch.zeros(shape[:-1], dtype=grad.dtype, device=grad.device)
                    st["exp_avg_sq_col"] = torch.zeros(shape[:-2] + shape[-1:], dtype=grad.dtype, device=grad.device)
                    st["exp_avg_res_row"] = torch.zeros(shape[:-1], dtype=grad.dtype, device=grad.device)
                    st["exp_avg_res_col"] = torch.zeros(shape[:-2] + shape[-1:], dtype=grad.dtype, device=grad.device)
                else:
                    st["exp_avg_sq"] = torch.zeros_like(grad)
            b1, b2, b3 = self.betas
            update = grad ** 2 + self.eps[0]
            if factored:
                st["exp_avg_sq_row"].mul_(b2).add_(update.mean(dim=-1), alpha=1.0 - b2)
                st["exp_avg_sq_col"].mul_(b2).add_(update.mean(dim=-2), alpha=1.0 - b2)
                update = self._approx_sq_grad(st["exp_avg_sq_row"], st["exp_avg_sq_col"])
                update.mul_(grad)
            else:
                st["exp_avg_sq"].mul_(b2).add_(update, alpha=1.0 - b2)
                update = st["exp_avg_sq"].rsqrt().mul_(grad)
            update.div_((self._rms(update) / self.clip).clamp_(min=1.0))
            exp_avg = st["exp_avg"]
            exp_avg.mul_(b1).add_(update, alpha=1.0 - b1)
            res = (update - exp_avg) ** 2 + self.eps[1]          # confidence-guided strategy: instability of the update
            if factored:
                st["exp_avg_res_row"].mul_(b3).add_(res.mean(dim=-1), alpha=1.0 - b3)
                st["exp_avg_res_col"].mul_(b3).add_(res.mean(dim=-2), alpha=1.0 - b3)
                update = self._approx_sq_grad(st["exp_avg_res_row"], st["exp_avg_res_col"]).mul_(exp_avg)
            else:
                update = exp_avg.clone()
            if self.wd != 0:
                p.add_(p, alpha=-self.wd * self.lr)
            update.mul_(self.lr)
            p.add_(-update)
