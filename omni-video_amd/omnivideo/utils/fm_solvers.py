"""Flow-matching DPM-Solver++ sampler with the reference scheduler's interface (reference:
omnivideo/utils/fm_solvers.py) for the configuration the pipeline uses
(x2x_gen_unified.py:417-426): solver_order 2, "dpmsolver++", "midpoint", flow_prediction,
lower_order_final, final sigma 0, sigmas supplied by get_sampling_sigmas.

As for UniPC, everything that depends only on the sigma schedule is evaluated once with the
reference's float32 scalar arithmetic and stored as one coefficient row per step; a step is
one fused launch (ovd_cfg_dpm_step).  The reference's samples are float32 from the first
step on (prev_sample takes the dtype of the converted model output, :793), so the loop keeps
the latent in a float32 buffer; a bf16 initial noise is read as is by the first step.
"""
import inspect
from typing import List, Optional

import numpy as np
import torch

from .. import _hip as H

NCOEF = H.DPM_COEF_STRIDE
DP_SIGMA, DP_ORDER, DP_RATIO, DP_B, DP_HALF_B, DP_INV_R0 = range(6)


class SchedulerOutput:
    def __init__(self, prev_sample):
        self.prev_sample = prev_sample


def get_sampling_sigmas(sampling_steps, shift):
    """reference :23-27"""
    sigma = np.linspace(1, 0, sampling_steps + 1)[:sampling_steps]
    return shift * sigma / (1 + (shift - 1) * sigma)


def retrieve_timesteps(scheduler, num_inference_steps=None, device=None, timesteps=None, sigmas=None, **kwargs):
    """reference :30-67"""
    if timesteps is not None and sigmas is not None:
        raise ValueError("Only one of `timesteps` or `sigmas` can be passed. Please choose one to set custom values")
    params = set(inspect.signature(scheduler.set_timesteps).parameters.keys())
    if timesteps is not None:
        if "timesteps" not in params:
            raise ValueError(f"The current scheduler class {scheduler.__class__}'s `set_timesteps` does not support "
                             f"custom timestep schedules. Please check whether you are using the correct scheduler.")
        scheduler.set_timesteps(timesteps=timesteps, device=device, **kwargs)
    elif sigmas is not None:
        if "sigmas" not in params:
            raise ValueError(f"The current scheduler class {scheduler.__class__}'s `set_timesteps` does not support "
                             f"custom sigmas schedules. Please check whether you are using the correct scheduler.")
        scheduler.set_timesteps(sigmas=sigmas, device=device, **kwargs)
    else:
        scheduler.set_timesteps(num_inference_steps, device=device, **kwargs)
    return scheduler.timesteps, len(scheduler.timesteps)


def _lam(s):
    return torch.log(1 - s) - torch.log(s)


def dpm_coefficients(sigmas, solver_order=2):
    """float32 table [steps, 8] of the per-step scalars of dpm_solver_first_order_update
    (reference :455-466) and multistep_dpm_solver_second_order_update (:527-560), with the
    order selection of step() (:733-770)."""
    n = sigmas.numel() - 1
    tab = torch.zeros(n, NCOEF, dtype=torch.float32)
    lower_order_nums = 0
    for i in range(n):
        sigma_t, sigma_s0 = sigmas[i + 1], sigmas[i]
        h = _lam(sigma_t) - _lam(sigma_s0)
        b = (1 - sigma_t) * (torch.exp(-h) - 1.0)
        final = i == n - 1                      # lower_order_final through final_sigmas_type == "zero"
        order = 1 if (solver_order == 1 or lower_order_nums < 1 or final) else 2
        row = tab[i]
        row[DP_SIGMA], row[DP_ORDER], row[DP_RATIO], row[DP_B] = sigma_s0, order, sigma_t / sigma_s0, b
        if order == 2:
            h_0 = _lam(sigma_s0) - _lam(sigmas[i - 1])
            row[DP_HALF_B] = 0.5 * b
            row[DP_INV_R0] = 1.0 / (h_0 / h)
        if lower_order_nums < solver_order:
            lower_order_nums += 1
    return tab


class FlowDPMSolverMultistepScheduler:
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, solver_order: int = 2,
                 prediction_type: str = "flow_prediction", shift: Optional[float] = 1.0,
                 use_dynamic_shifting=False, thresholding: bool = False, dynamic_thresholding_ratio: float = 0.995,
                 sample_max_value: float = 1.0, algorithm_type: str = "dpmsolver++", solver_type: str = "midpoint",
                 lower_order_final: bool = True, euler_at_final: bool = False,
                 final_sigmas_type: Optional[str] = "zero", lambda_min_clipped: float = -float("inf"),
                 variance_type: Optional[str] = None, invert_sigmas: bool = False):
        if (prediction_type, algorithm_type, solver_type, final_sigmas_type) != \
                ("flow_prediction", "dpmsolver++", "midpoint", "zero") or solver_order not in (1, 2) \
                or thresholding or use_dynamic_shifting or euler_at_final or invert_sigmas or not lower_order_final:
            raise NotImplementedError("only the pipeline's DPM-Solver++ configuration is built (order <= 2, "
                                      "dpmsolver++, midpoint, flow_prediction, lower_order_final, zero)")
        self.num_train_timesteps, self.solver_order, self.shift = num_train_timesteps, solver_order, shift
        alphas = np.linspace(1, 1 / num_train_timesteps, num_train_timesteps)[::-1].copy()     # reference :183-195
        sig = torch.from_numpy(1.0 - alphas).to(torch.float32)
        sig = shift * sig / (1 + (shift - 1) * sig)
        self.sigmas = sig
        self.timesteps = sig * num_train_timesteps
        self.sigma_min, self.sigma_max = sig[-1].item(), sig[0].item()
        self.num_inference_steps = None
        self._step_index = None
        self._state = None

    @property
    def step_index(self):
        return self._step_index

    def set_timesteps(self, num_inference_steps=None, device=None, sigmas=None, mu=None, shift=None):
        """reference :227-291"""
        if sigmas is None:
            sigmas = np.linspace(self.sigma_max, self.sigma_min, num_inference_steps + 1).copy()[:-1]
        if shift is None:
            shift = self.shift
        sigmas = shift * np.asarray(sigmas) / (1 + (shift - 1) * np.asarray(sigmas))
        timesteps = sigmas * self.num_train_timesteps
        self.sigmas = torch.from_numpy(np.concatenate([sigmas, [0]]).astype(np.float32))
        self.timesteps = torch.from_numpy(timesteps).to(device=device, dtype=torch.int64)
        self.num_inference_steps = len(timesteps)
        self._step_index = None
        self._state = None

    # ------------------------------------------------------------------ device state
    def prepare(self, sample):
        """Allocates the x0 history and uploads the coefficient table for latents shaped like
        `sample` (called implicitly by the first step)."""
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' "
                             "after creating the scheduler")
        dev = sample.device
        self._state = dict(coef=dpm_coefficients(self.sigmas, self.solver_order).to(dev),
                           m1=torch.zeros(sample.shape, dtype=torch.float32, device=dev),
                           tsteps=self.timesteps.to(dev, torch.float64), shape=tuple(sample.shape))
        self._step_index = 0
        return self._state

    def _check(self, sample):
        if self._state is None:
            self.prepare(sample)
        if tuple(sample.shape) != self._state["shape"]:
            raise ValueError("sample shape changed in the middle of a sampling run")
        return self._state

    def step_cfg(self, cond, uncond, guide_scale, sample, step_ptr=None, out=None):
        """Fused guidance + step.  `sample` fp32: updated IN PLACE unless `out` (fp32) is given;
        `sample` bf16 (the initial noise): `out` is required."""
        st = self._check(sample)
        if step_ptr is None and self._step_index >= self.num_inference_steps:
            raise IndexError("more steps than set_timesteps() prepared")
        if out is None:
            if sample.dtype != torch.float32:
                raise TypeError("a bf16 sample needs an fp32 `out`: DPM-Solver++ samples are float32 after a step")
            out = sample
        H.cfg_dpm_step(cond, uncond, sample, out, st["m1"], st["coef"], guide_scale, step=self._step_index,
                       step_ptr=step_ptr)
        self._step_index += 1
        return out

    def step(self, model_output, timestep, sample, generator=None, variance_noise=None, return_dict: bool = True):
        """reference :707-799: returns the next sample as a new float32 tensor."""
        x = sample.contiguous()
        out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        mo = model_output.to(torch.float32).contiguous()
        self.step_cfg(mo, mo, 1.0, x, out=out)
        return (out,) if not return_dict else SchedulerOutput(prev_sample=out)

    def scale_model_input(self, sample, *args, **kwargs):
        return sample

    def __len__(self):
        return self.num_train_timesteps
