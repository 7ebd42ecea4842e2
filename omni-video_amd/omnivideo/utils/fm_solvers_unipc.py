"""Flow-matching UniPC sampler with the reference scheduler's interface (reference:
omnivideo/utils/fm_solvers_unipc.py) for the configuration the pipeline uses:
solver_order 2, "bh2", predict_x0, flow_prediction, lower_order_final, final sigma 0.

Everything that depends only on the sigma schedule - which is everything except the
tensors - is evaluated once in `set_timesteps` with the same float32 scalar arithmetic as
the reference and stored as one coefficient row per step.  A step is then a single fused
launch (ovd_cfg_unipc_step): guidance combine, x0 conversion, corrector, history shift and
predictor in one pass over the latent, with no host round trip, so the loop can be
replayed from a captured graph with a device-side step counter.
"""
from typing import List, Optional, Tuple, Union

import numpy as np
import torch

from .. import _hip as H

NCOEF = H.UNIPC_COEF_STRIDE
(CF_SIGMA, CF_RESERVED, CF_USE_CORR, CF_C_ORDER, CF_C_R, CF_C_A, CF_C_AB, CF_C_RK, CF_C_RHO0, CF_C_RHO1,
 CF_P_ORDER, CF_P_R, CF_P_A, CF_P_AB, CF_P_RK) = range(15)


class SchedulerOutput:
    def __init__(self, prev_sample):
        self.prev_sample = prev_sample


def _lam(s):
    return torch.log(1 - s) - torch.log(s)


def unipc_coefficients(sigmas, solver_order=2, sample_dtype=torch.float32):
    """float32 table [steps, 16] of the per-step scalars of multistep_uni_c_bh_update
    (reference fm_solvers_unipc.py:549-617) and multistep_uni_p_bh_update (:406-474).
    `sample_dtype` matters because the reference casts the solved rho to the sample's
    dtype (:463,608)."""
    if solver_order != 2:
        raise NotImplementedError("only solver_order=2 is built (the pipeline's setting)")
    n = sigmas.numel() - 1
    tab = torch.zeros(n, NCOEF, dtype=torch.float32)
    lower_order_nums, prev_order = 0, None
    for i in range(n):
        row = tab[i]
        row[CF_SIGMA] = sigmas[i]
        row[CF_C_ORDER] = row[CF_P_ORDER] = 1
        row[CF_C_RK] = row[CF_P_RK] = 1
        if i > 0:
            order = prev_order
            sig_t, sig_s0 = sigmas[i], sigmas[i - 1]
            h = _lam(sig_t) - _lam(sig_s0)
            hh = -h
            h_phi_1 = torch.expm1(hh)
            B_h = torch.expm1(hh)
            row[CF_USE_CORR], row[CF_C_ORDER] = 1, order
            row[CF_C_R] = sig_t / sig_s0
            row[CF_C_A] = (1 - sig_t) * h_phi_1
            row[CF_C_AB] = (1 - sig_t) * B_h
            if order == 1:
                row[CF_C_RHO1] = 0.5
            else:
                rk = (_lam(sigmas[i - 2]) - _lam(sig_s0)) / h
                rks = torch.tensor([rk, 1.0])
                h_phi_k = h_phi_1 / hh - 1
                b0 = h_phi_k * 1 / B_h
                h_phi_k = h_phi_k / hh - 1 / 2
                b1 = h_phi_k * 2 / B_h
                R = torch.stack([torch.pow(rks, 0), torch.pow(rks, 1)])
                rhos = torch.linalg.solve(R, torch.tensor([b0, b1])).to(sample_dtype).float()
                row[CF_C_RK], row[CF_C_RHO0], row[CF_C_RHO1] = rk, rhos[0], rhos[1]
        this_order = min(min(solver_order, n - i), lower_order_nums + 1)
        sig_t, sig_s0 = sigmas[i + 1], sigmas[i]
        h = _lam(sig_t) - _lam(sig_s0)
        hh = -h
        h_phi_1 = torch.expm1(hh)
        B_h = torch.expm1(hh)
        row[CF_P_ORDER] = this_order
        row[CF_P_R] = sig_t / sig_s0
        row[CF_P_A] = (1 - sig_t) * h_phi_1
        row[CF_P_AB] = (1 - sig_t) * B_h
        if this_order == 2:
            row[CF_P_RK] = (_lam(sigmas[i - 1]) - _lam(sig_s0)) / h
        prev_order = this_order
        if lower_order_nums < solver_order:
            lower_order_nums += 1
    return tab


class FlowUniPCMultistepScheduler:
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, solver_order: int = 2,
                 prediction_type: str = "flow_prediction", shift: Optional[float] = 1.0,
                 use_dynamic_shifting=False, thresholding: bool = False, predict_x0: bool = True,
                 solver_type: str = "bh2", lower_order_final: bool = True, disable_corrector: List[int] = [],
                 timestep_spacing: str = "linspace", steps_offset: int = 0,
                 final_sigmas_type: Optional[str] = "zero", **unused):
        if (solver_order, prediction_type, predict_x0, solver_type, lower_order_final, final_sigmas_type) != \
                (2, "flow_prediction", True, "bh2", True, "zero") or thresholding or use_dynamic_shifting \
                or disable_corrector:
            raise NotImplementedError("only the pipeline's UniPC configuration is built "
                                      "(order 2, bh2, predict_x0, flow_prediction, lower_order_final, zero)")
        self.num_train_timesteps, self.solver_order, self.shift = num_train_timesteps, solver_order, shift
        # reference :108-133: sigma range of the unshifted (shift given at construction) schedule
        alphas = np.linspace(1, 1 / num_train_timesteps, num_train_timesteps)[::-1].copy()
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
        """reference :161-228"""
        if sigmas is None:
            sigmas = np.linspace(self.sigma_max, self.sigma_min, num_inference_steps + 1).copy()[:-1]
        if shift is None:
            shift = self.shift
        sigmas = shift * sigmas / (1 + (shift - 1) * sigmas)
        timesteps = sigmas * self.num_train_timesteps
        self.sigmas = torch.from_numpy(np.concatenate([sigmas, [0]]).astype(np.float32))
        self.timesteps = torch.from_numpy(timesteps).to(device=device, dtype=torch.int64)
        self.num_inference_steps = len(timesteps)
        self._step_index = None
        self._state = None
        self._device = device

    # ------------------------------------------------------------------ device state
    def prepare(self, sample):
        """Allocates the history buffers and uploads the coefficient table for latents
        shaped/typed like `sample` (called implicitly by the first step)."""
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' "
                             "after creating the scheduler")
        dev = sample.device
        self._state = dict(
            coef=unipc_coefficients(self.sigmas, self.solver_order, sample.dtype).to(dev),
            last=torch.zeros_like(sample), m1=torch.zeros(sample.shape, dtype=torch.float32, device=dev),
            m2=torch.zeros(sample.shape, dtype=torch.float32, device=dev),
            tsteps=self.timesteps.to(dev, torch.float64), shape=tuple(sample.shape), dtype=sample.dtype)
        self._step_index = 0
        return self._state

    def _check(self, sample):
        if self._state is None:
            self.prepare(sample)
        st = self._state
        if tuple(sample.shape) != st["shape"] or sample.dtype != st["dtype"]:
            raise ValueError("sample shape/dtype changed in the middle of a sampling run")
        return st

    def step_cfg(self, cond, uncond, guide_scale, sample, step_ptr=None):
        """Fused guidance + step, IN PLACE on `sample` (contiguous, latent dtype).  With
        step_ptr (device int32) the row index is read on the device."""
        st = self._check(sample)
        if step_ptr is None and self._step_index >= self.num_inference_steps:
            raise IndexError("more steps than set_timesteps() prepared")
        H.cfg_unipc_step(cond, uncond, sample, st["last"], st["m1"], st["m2"], st["coef"], guide_scale,
                         step=self._step_index, step_ptr=step_ptr)
        self._step_index += 1
        return sample

    def step(self, model_output, timestep, sample, return_dict: bool = True, generator=None):
        """reference :656-740: returns the next sample as a new tensor."""
        x = sample.contiguous().clone()
        mo = model_output.to(torch.float32).contiguous()
        self.step_cfg(mo, mo, 1.0, x)
        return (x,) if not return_dict else SchedulerOutput(prev_sample=x)

    def scale_model_input(self, sample, *args, **kwargs):
        return sample

    def __len__(self):
        return self.num_train_timesteps
