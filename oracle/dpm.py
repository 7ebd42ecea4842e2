"""TEST INFRASTRUCTURE ONLY - CPU restatement of the reference's flow-matching DPM-Solver++
sampler (omnivideo/utils/fm_solvers.py) in the configuration the pipeline builds
(x2x_gen_unified.py:417-426): solver_order 2, "dpmsolver++", "midpoint", flow_prediction,
lower_order_final, final sigma 0, sigmas from get_sampling_sigmas.

Pinned by tests/golden/dpm.npz (schedules and trajectories produced by the reference class
itself, oracle/gen_golden.py).  Only tests, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import this module.
"""
import numpy as np
import torch


def sampling_sigmas(steps, shift):
    """fm_solvers.py:23-27 (float64)."""
    sigma = np.linspace(1, 0, steps + 1)[:steps]
    return shift * sigma / (1 + (shift - 1) * sigma)


def schedule(steps, shift, num_train_timesteps=1000):
    """retrieve_timesteps(scheduler, sigmas=...) -> set_timesteps (fm_solvers.py:227-291) with
    the scheduler's own shift of 1: sigmas float32 [steps + 1] (last 0), timesteps int64."""
    sig = sampling_sigmas(steps, shift)
    sig = 1.0 * sig / (1 + (1.0 - 1) * sig)
    timesteps = torch.from_numpy(sig * num_train_timesteps).to(torch.int64)
    sigmas = torch.from_numpy(np.concatenate([sig, [0]]).astype(np.float32))
    return sigmas, timesteps


def _lam(sigma):
    return torch.log(1 - sigma) - torch.log(sigma)


def coefficients(sigmas, i, order):
    """The 0-dim float32 scalars of dpm_solver_first_order_update (fm_solvers.py:455-466) and
    multistep_dpm_solver_second_order_update (:527-560) at step i."""
    sigma_t, sigma_s0 = sigmas[i + 1], sigmas[i]
    h = _lam(sigma_t) - _lam(sigma_s0)
    c = dict(ratio=sigma_t / sigma_s0, b=(1 - sigma_t) * (torch.exp(-h) - 1.0))
    if order == 2:
        h_0 = _lam(sigma_s0) - _lam(sigmas[i - 1])
        c["inv_r0"] = 1.0 / (h_0 / h)
        c["half_b"] = 0.5 * c["b"]
    return c


class DPMpp:
    def __init__(self, steps, shift, solver_order=2):
        self.sigmas, self.timesteps = schedule(steps, shift)
        self.n = steps
        self.order = solver_order
        self.i = 0
        self.lower_order_nums = 0
        self.m_prev = None

    def step(self, model_output, sample):
        """fm_solvers.py:707-799: returns the next sample (float32, as the reference does from the
        first step on: prev_sample takes the dtype of the converted model output)."""
        i = self.i
        x0 = sample - self.sigmas[i] * model_output            # :383-385, promoted to float32
        sample = sample.to(torch.float32)
        final = i == self.n - 1                                 # final_sigmas_type == "zero"
        second_last_low = i == self.n - 2 and self.n < 15
        if self.order == 1 or self.lower_order_nums < 1 or final:
            c = coefficients(self.sigmas, i, 1)
            x = c["ratio"] * sample - c["b"] * x0
        else:   # order 2 (also when lower_order_second asks for it: with solver_order 2 it is the same branch)
            del second_last_low
            c = coefficients(self.sigmas, i, 2)
            d1 = c["inv_r0"] * (x0 - self.m_prev)
            x = c["ratio"] * sample - c["b"] * x0 - c["half_b"] * d1
        if self.lower_order_nums < self.order:
            self.lower_order_nums += 1
        self.m_prev = x0
        self.i += 1
        return x.to(x0.dtype)


def sample_loop(model_cond, model_uncond, noise, num_steps, shift, guide_scale):
    """The denoising loop of x2x_gen_unified.py:474-492 with the dpm++ solver.  model_*(latent, t)
    -> fp32 velocity with the latent's shape; noise is the initial latent in the run dtype."""
    sched = DPMpp(num_steps, shift)
    lat = noise
    for t in sched.timesteps:
        c = model_cond(lat, int(t))
        u = model_uncond(lat, int(t))
        lat = sched.step(u + guide_scale * (c - u), lat)
    return lat
