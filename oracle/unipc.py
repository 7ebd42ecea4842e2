"""TEST INFRASTRUCTURE ONLY - CPU oracle for the flow-matching UniPC sampler and the
classifier-free-guidance loop.  Product code never imports this file.

Restates, for the one configuration the pipeline uses (solver_order=2, "bh2",
predict_x0, flow_prediction, lower_order_final, final sigma 0; reference:
omnivideo/x2x_gen_unified.py:410-415), the schedule and update rules of
omnivideo/utils/fm_solvers_unipc.py.  Scalars are float32 0-dim torch tensors, as
in the reference, so the arithmetic rounds the same way.
"""
import numpy as np
import torch


def schedule(num_steps, shift, num_train_timesteps=1000):
    """sigmas float32 [steps+1] (last = 0) and int64 timesteps [steps]
    (fm_solvers_unipc.py:108-133 for sigma_max/min with shift=1, :183-212)."""
    alphas = np.linspace(1, 1 / num_train_timesteps, num_train_timesteps)[::-1].copy()
    base = torch.from_numpy(1.0 - alphas).to(torch.float32)
    sigma_max, sigma_min = base[0].item(), base[-1].item()
    s = np.linspace(sigma_max, sigma_min, num_steps + 1).copy()[:-1]
    s = shift * s / (1 + (shift - 1) * s)
    timesteps = torch.from_numpy(s * num_train_timesteps).to(torch.int64)
    sigmas = torch.from_numpy(np.concatenate([s, [0]]).astype(np.float32))
    return sigmas, timesteps


def _lam(sig):
    return torch.log(1 - sig) - torch.log(sig)


SCALAR_MODE = "device"


def _smul(scalar, x):
    """0-dim float32 scalar times a tensor in the sample dtype.  Type promotion keeps
    the tensor's dtype (bf16 samples stay bf16).  torch's device kernels keep the
    host scalar in float32 and round the product once ("device"); torch's CPU
    kernels first round the scalar to the tensor dtype ("cpu", what the CPU-made
    bf16 golden trajectories contain).  Identical for float32 samples."""
    if SCALAR_MODE == "cpu":
        return scalar * x
    return (scalar * x.float()).to(x.dtype)


class UniPC:
    """Predictor-corrector state machine (fm_solvers_unipc.py:656-740)."""

    def __init__(self, num_steps, shift):
        self.sigmas, self.timesteps = schedule(num_steps, shift)
        self.n = num_steps
        self.i = 0
        self.hist = [None, None]      # converted model outputs (x0 predictions), oldest first
        self.last_sample = None
        self.lower_order_nums = 0
        self.this_order = None

    def _common(self, sig_t, sig_s0):
        h = _lam(sig_t) - _lam(sig_s0)
        hh = -h
        h_phi_1 = torch.expm1(hh)
        return h, hh, h_phi_1, torch.expm1(hh)

    def _correct(self, m_t, x, order):
        # fm_solvers_unipc.py:487-627
        i = self.i
        sig_t, sig_s0 = self.sigmas[i], self.sigmas[i - 1]
        m0 = self.hist[-1]
        h, hh, h_phi_1, B_h = self._common(sig_t, sig_s0)
        rks, D1s = [], []
        for k in range(1, order):
            rk = (_lam(self.sigmas[i - (k + 1)]) - _lam(sig_s0)) / h
            rks.append(rk)
            D1s.append((self.hist[-(k + 1)] - m0) / rk)
        rks.append(1.0)
        rks = torch.tensor(rks)
        R, b = [], []
        h_phi_k = h_phi_1 / hh - 1
        fact = 1
        for k in range(1, order + 1):
            R.append(torch.pow(rks, k - 1))
            b.append(h_phi_k * fact / B_h)
            fact *= k + 1
            h_phi_k = h_phi_k / hh - 1 / fact
        R, b = torch.stack(R), torch.tensor(b)
        rhos = torch.tensor([0.5], dtype=x.dtype) if order == 1 else torch.linalg.solve(R, b).to(x.dtype)
        x_t_ = _smul(sig_t / sig_s0, x) - (1 - sig_t) * h_phi_1 * m0
        corr = 0
        for k, d in enumerate(D1s):
            corr = corr + rhos[k] * d
        x_t = x_t_ - (1 - sig_t) * B_h * (corr + rhos[-1] * (m_t - m0))
        return x_t.to(x.dtype)

    def _predict(self, x, order):
        # fm_solvers_unipc.py:351-485
        i = self.i
        sig_t, sig_s0 = self.sigmas[i + 1], self.sigmas[i]
        m0 = self.hist[-1]
        h, hh, h_phi_1, B_h = self._common(sig_t, sig_s0)
        x_t = _smul(sig_t / sig_s0, x) - (1 - sig_t) * h_phi_1 * m0
        if order == 2:
            rk = (_lam(self.sigmas[i - 1]) - _lam(sig_s0)) / h
            D1 = (self.hist[-2] - m0) / rk
            x_t = x_t - (1 - sig_t) * B_h * (torch.tensor([0.5], dtype=x.dtype)[0] * D1)
        return x_t.to(x.dtype)

    def step(self, model_output, sample):
        """model_output fp32 velocity, sample in the latent dtype; returns the next sample."""
        i = self.i
        m_t = sample - self.sigmas[i] * model_output          # :316-322
        if i > 0 and self.last_sample is not None:
            sample = self._correct(m_t, self.last_sample, self.this_order)
        self.hist = [self.hist[1], m_t]
        order = min(2, self.n - i)
        self.this_order = min(order, self.lower_order_nums + 1)
        self.last_sample = sample
        nxt = self._predict(sample, self.this_order)
        if self.lower_order_nums < 2:
            self.lower_order_nums += 1
        self.i += 1
        return nxt


def sample_loop(model_cond, model_uncond, noise, num_steps, shift, guide_scale):
    """The denoising loop of x2x_gen_unified.py:474-492.  model_*(latent, t) -> fp32
    velocity with the latent's shape; noise is the initial latent in the run dtype."""
    sched = UniPC(num_steps, shift)
    lat = noise
    for t in sched.timesteps:
        c = model_cond(lat, int(t))
        u = model_uncond(lat, int(t))
        v = u + guide_scale * (c - u)
        lat = sched.step(v, lat)
    return lat
