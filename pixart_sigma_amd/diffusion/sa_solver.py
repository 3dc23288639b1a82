"""SA-Solver sampling loop around the denoiser - the third `--sampling_algo` of scripts/inference.py (reference
scripts/inference.py:119-133 -> diffusion/sa_sampler.py:9-93 -> diffusion/model/sa_solver.py:325-358, 371-396, 398-418, 449-476, 478-560,
644-753, 755-909).

Same public surface (`SASolverSampler(model, device=...).sample(S=..., batch_size=..., shape=..., eta=..., conditioning=...,
unconditional_conditioning=..., unconditional_guidance_scale=..., model_kwargs=...)` -> `(samples, None)`), restricted to what the entry point
uses: discrete linear-beta VP schedule, model_type 'noise', classifier-free guidance, data prediction, mode 'few_steps' (the rescaled
stochastic Adams predictor / corrector of the paper's appendix D), PEC, predictor order 2, corrector order 2, time-uniform steps,
tau(t) = eta inside 0.2 <= t <= 0.8 and 0 outside.

As in dpm_solver.py every schedule scalar (alpha_t, sigma_t, lambda_t of the S + 1 time points, the Adams coefficients of each step) is
computed once on the host in float32 - the reference re-derives them on the device at every step through a sort / gather interpolation
(sa_solver.py:1099-1137) - so the loop issues only the denoiser call and a handful of fused elementwise updates per step and never
synchronises.  The Gaussian draws come from the device generator, one `randn_like` per step in the reference's order (including the unused draw
in front of the first evaluation, sa_solver.py:786), or from `normals_sequence` (the reference's own, unused, parameter name) when the caller -
the parity tests - supplies them."""
import numpy as np
import torch

from .dpm_solver import NoiseScheduleVP


class SASolver:
    def __init__(self, eps_fn, ns):
        """eps_fn(x, t_continuous: float) -> guided noise prediction (model_wrapper.model_fn, sa_solver.py:297-320)"""
        self.eps, self.ns = eps_fn, ns

    # ---- coefficient algebra, float32 torch scalars on the host (sa_solver.py:449-476, 478-494, 541-560)
    @staticmethod
    def _exp_int(order, a, b, tau):
        """integral over [a, b] of exp((1 + tau^2) x) x^order dx, order 0 / 1"""
        k = 1 + tau ** 2
        bc, ac = k * b, k * a
        if order == 0:
            return torch.exp(bc) * (1 - torch.exp(-(bc - ac))) / k
        assert order == 1
        return torch.exp(bc) * ((bc - 1) - (ac - 1) * torch.exp(-(bc - ac))) / (k ** 2)

    def _coefficients(self, order, a, b, lams, tau):
        """Adams coefficients of the `order` most recent evaluations: Lagrange basis through `lams`, integrated against the exponential weight"""
        if order == 1:
            return [self._exp_int(0, a, b, tau)]
        assert order == 2
        d01, d10 = lams[0] - lams[1], lams[1] - lams[0]
        basis = [[1 / d01, -lams[1] / d01], [1 / d10, -lams[0] / d10]]        # [node][power 1, power 0]
        return [basis[i][0] * self._exp_int(1, a, b, tau) + basis[i][1] * self._exp_int(0, a, b, tau) for i in range(2)]

    def _update(self, order, x, tau, models, i_prev, i_t, noise, corrector):
        """one stochastic Adams-Bashforth (predictor) / Adams-Moulton (corrector) update from time index i_prev[-1] to i_t (sa_solver.py:644-753).
        `models` / `i_prev`: data predictions and their time indices, oldest first; the corrector's node list also holds the new point."""
        lam, sig = self.lam, self.sig
        tau = torch.as_tensor(float(tau), dtype=torch.float32)
        lam_t, lam_p = lam[i_t], lam[i_prev[-1]]
        h = lam_t - lam_p
        nodes = (i_prev + [i_t]) if corrector else i_prev
        lams = [lam[nodes[-(i + 1)]] for i in range(order)]
        co = self._coefficients(order, lam_p, lam_t, lams, tau)
        if order == 2:        # the O(h^3) term of the few-steps variant (UniPC-like, sa_solver.py:665-678 / 723-733)
            k = 1 + tau ** 2
            r = (h * k - 1 + torch.exp(k * (-h))) / (k ** 2)
            adj = torch.exp(k * lam_t) * ((h / 2 - r / h) if corrector else (h ** 2 / 2 - r) / (lam[i_prev[-1]] - lam[i_prev[-2]]))
            co = [co[0] + adj, co[1] - adj]
        scale = (1 + tau ** 2) * sig[i_t] * torch.exp(-tau ** 2 * lam_t)
        grad = float(scale * co[0]) * models[-1]
        if order == 2:
            grad = grad + float(scale * co[1]) * models[-2]
        nz = float(sig[i_t] * torch.sqrt(1 - torch.exp(-2 * tau ** 2 * h)))
        return float(torch.exp(-tau ** 2 * h) * (sig[i_t] / sig[i_prev[-1]])) * x + grad + nz * noise

    def sample(self, x, tau_fn, steps, normals=None, predictor_order=2, corrector_order=2):
        """sample_few_steps with skip_type 'time', skip_order 1, pc_mode 'PEC' (sa_solver.py:755-909)"""
        assert predictor_order == 2 and corrector_order == 2 and steps >= 2
        ns = self.ns
        ts = torch.linspace(ns.T, 1.0 / ns.total_N, steps + 1)                 # get_time_steps, :412-414
        self.lam, self.sig, self.alp = ns.marginal_lambda(ts), ns.marginal_std(ts), ns.marginal_alpha(ts)
        tl = [float(v) for v in ts]
        draws = iter(normals) if normals is not None else None

        def draw():
            return next(draws).to(x) if draws is not None else torch.randn_like(x)

        def x0_pred(xx, i):                                                    # data_prediction_fn, :377-386
            return (xx - float(self.sig[i]) * self.eps(xx, tl[i])) / float(self.alp[i])

        with torch.no_grad():
            draw()                                                             # :786 (drawn, not used)
            models, i_prev = [x0_pred(x, 0)], [0]
            for step in range(1, steps + 1):
                last = step == steps
                p_ord = min(predictor_order, step, steps - step + 1)           # warm-up (:811) and lower_order_final (:848)
                c_ord = min(corrector_order, step + 1, steps - step + 2)
                noise = draw()
                tau = 0.0 if last else tau_fn(tl[step])                        # the final step is deterministic and is not corrected (:857-861, :877-884)
                x_p = self._update(p_ord, x, tau, models, i_prev, step, noise, corrector=False)
                if last:
                    x = x_p
                    break
                models.append(x0_pred(x_p, step))
                x = self._update(c_ord, x, tau, models, i_prev, step, noise, corrector=True)
                i_prev.append(step)
                if step >= max(predictor_order, corrector_order - 1):         # the history keeps predictor_order evaluations (:903)
                    del models[0]
        return x


class SASolverSampler:
    """diffusion/sa_sampler.py:9-93"""

    def __init__(self, model, noise_schedule="linear", diffusion_steps=1000, device="cuda"):
        assert noise_schedule == "linear"
        self.model, self.device = model, device
        betas = np.linspace(0.0001 * 1000 / diffusion_steps, 0.02 * 1000 / diffusion_steps, diffusion_steps, dtype=np.float64)
        self.alphas_cumprod = torch.from_numpy(np.cumprod(1.0 - betas, axis=0)).to(torch.float32)   # float64 product, then float32 (:22-24)

    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, eta=0.0, x_T=None, unconditional_guidance_scale=1.0, unconditional_conditioning=None,
               model_kwargs={}, normals_sequence=None, **unused):
        from .dpm_solver import DPM_Solver
        C, H, W = shape
        img = torch.randn((batch_size, C, H, W), device=self.device) if x_T is None else x_T
        ns = NoiseScheduleVP("discrete", alphas_cumprod=self.alphas_cumprod)
        wrap = DPM_Solver(self.model, ns, conditioning, unconditional_conditioning, unconditional_guidance_scale, dict(model_kwargs))
        solver = SASolver(wrap._eps, ns)
        x = solver.sample(img, lambda t: eta if 0.2 <= t <= 0.8 else 0, S, normals=normals_sequence)
        return x.to(self.device), None
