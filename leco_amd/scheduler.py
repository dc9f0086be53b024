"""Noise schedulers (DDIM on the benchmarked path; DDPM / LMS / Euler-ancestral for config completeness) with the duck-type the reference uses (``set_timesteps``, ``timesteps``,
``init_noise_sigma``, ``scale_model_input``, ``step(...).prev_sample``; train_lora.py:143-145,
195-199, train_util.py:55,153,184,190) and the constructor arguments of model_util.py:239-246
(scaled-linear betas 0.00085..0.012, 1000 train steps, clip_sample=False, epsilon / v_prediction).

DDIM with eta = 0 is linear in (sample, model_output):  x_prev = c_x(t) x + c_e(t) out.
``coef_table()`` exposes (c_x, c_e) for every step of the current schedule so that the fused
denoising loop (``leco_cfg_ddim_step``) reads them on the device and a captured hipGraph can be
replayed for every step."""
from __future__ import annotations

import numpy as np
import torch


class SchedulerOutput:
    def __init__(self, prev_sample, pred_original_sample=None):
        self.prev_sample = prev_sample
        self.pred_original_sample = pred_original_sample


class DDIMScheduler:
    order = 1

    def __init__(self, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", num_train_timesteps=1000,
                 clip_sample=False, prediction_type="epsilon", set_alpha_to_one=True, steps_offset=0):
        if beta_schedule != "scaled_linear":
            raise ValueError("only the scaled_linear schedule of the SD model family is implemented")
        if clip_sample:
            raise ValueError("clip_sample=True is not used by the reference and not implemented")
        self.num_train_timesteps = num_train_timesteps
        self.prediction_type = prediction_type
        self.steps_offset = steps_offset
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    def set_timesteps(self, num_inference_steps: int, device=None):
        if num_inference_steps > self.num_train_timesteps:
            raise ValueError("num_inference_steps > num_train_timesteps")
        self.num_inference_steps = num_inference_steps
        ratio = self.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64) + self.steps_offset
        self.timesteps = torch.from_numpy(ts).to(device)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def _coef(self, t: int):
        prev_t = t - self.num_train_timesteps // self.num_inference_steps
        a_t = float(self.alphas_cumprod[t])
        a_p = float(self.alphas_cumprod[prev_t]) if prev_t >= 0 else float(self.final_alpha_cumprod)
        sa, sb = a_t ** 0.5, (1.0 - a_t) ** 0.5
        spa, spb = a_p ** 0.5, (1.0 - a_p) ** 0.5
        if self.prediction_type == "epsilon":
            # x0 = (x - sb e)/sa ; x_prev = spa x0 + spb e
            return spa / sa, spb - spa * sb / sa
        if self.prediction_type == "v_prediction":
            # x0 = sa x - sb v ; e = sa v + sb x ; x_prev = spa x0 + spb e
            return spa * sa + spb * sb, spb * sa - spa * sb
        raise ValueError(f"unknown prediction_type {self.prediction_type}")

    def coef_table(self) -> torch.Tensor:
        """fp32 [len(timesteps)][2] = (c_x, c_e) per step of the current schedule."""
        return torch.tensor([self._coef(int(t)) for t in self.timesteps], dtype=torch.float32)

    def step(self, model_output, timestep, sample):
        if self.num_inference_steps is None:
            raise ValueError("call set_timesteps first")
        cx, ce = self._coef(int(timestep))
        return SchedulerOutput(cx * sample + ce * model_output)


# =============================================================================================
# The other three schedulers the reference accepts (model_util.py:247-274).  All of them are LINEAR in
# (sample, model_output, fresh noise, previous derivatives), so each exposes `rows()`: one fp32 row of
# SCHED_ROW coefficients per step that the fused denoising loop (`leco_cfg_sched_step`) reads on the device:
#     x_next = c_x x + c_e out + c_n noise + c_h1 h1 + c_h2 h2 + c_h3 h3
#     d      = d_x x + d_e out          (derivative pushed into the history h1 <- d, h2 <- h1, h3 <- h2)
#     x_in   = s_in x_next              (scale_model_input for the NEXT step, bf16 UNet input)
# =============================================================================================
SCHED_ROW = 12   # c_x, c_e, c_n, c_h1, c_h2, c_h3, s_in, d_x, d_e, 3 unused


class _SigmaBase:
    """beta schedule + sigma table shared by the sigma-space schedulers."""

    def __init__(self, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", num_train_timesteps=1000,
                 prediction_type="epsilon"):
        if beta_schedule != "scaled_linear":
            raise ValueError("only the scaled_linear schedule of the SD model family is implemented")
        self.num_train_timesteps = num_train_timesteps
        self.prediction_type = prediction_type
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.num_inference_steps = None
        self._all_sigmas = (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).numpy()
        self.set_timesteps(num_train_timesteps)
        self.num_inference_steps = None

    # "linspace" timestep spacing (the default of these schedulers in diffusers 0.20): float timesteps
    def set_timesteps(self, num_inference_steps: int, device=None):
        self.num_inference_steps = num_inference_steps
        ts = np.linspace(0, self.num_train_timesteps - 1, num_inference_steps, dtype=float)[::-1].copy()
        sig = np.interp(ts, np.arange(0, len(self._all_sigmas)), self._all_sigmas)
        self.sigmas = np.concatenate([sig, [0.0]]).astype(np.float32)
        self.timesteps = torch.from_numpy(ts.astype(np.float32)).to(device)
        self._reset()

    def _reset(self):
        pass

    @property
    def init_noise_sigma(self) -> float:
        return float(self.sigmas.max())

    def _index(self, timestep) -> int:
        t = float(timestep)
        idx = (self.timesteps.detach().cpu().double() - t).abs().argmin()
        return int(idx)

    def scale_model_input(self, sample, timestep):
        s = float(self.sigmas[self._index(timestep)])
        return sample / ((s * s + 1) ** 0.5)

    def _x0_coef(self, sigma: float):
        """pred_x0 = p_x x + p_e out."""
        if self.prediction_type == "epsilon":
            return 1.0, -sigma
        if self.prediction_type == "v_prediction":
            return 1.0 / (sigma * sigma + 1), -sigma / (sigma * sigma + 1) ** 0.5
        raise ValueError(f"unknown prediction_type {self.prediction_type}")

    def _deriv_coef(self, sigma: float):
        """derivative (x - pred_x0) / sigma = d_x x + d_e out."""
        px, pe = self._x0_coef(sigma)
        return (1.0 - px) / sigma, -pe / sigma

    def _scale_next(self, i: int) -> float:
        s = float(self.sigmas[i + 1]) if i + 1 < len(self.sigmas) - 1 else 0.0
        return 1.0 / (s * s + 1) ** 0.5


class EulerAncestralDiscreteScheduler(_SigmaBase):
    """x' = x + d (sigma_down - sigma) + noise sigma_up with d = (x - x0)/sigma (stochastic)."""
    order = 1
    needs_noise = True
    n_hist = 0

    def _sig(self, i: int):
        s_from, s_to = float(self.sigmas[i]), float(self.sigmas[i + 1])
        s_up = (s_to ** 2 * (s_from ** 2 - s_to ** 2) / s_from ** 2) ** 0.5
        s_down = (s_to ** 2 - s_up ** 2) ** 0.5
        return s_from, s_up, s_down

    def row(self, i: int):
        s, s_up, s_down = self._sig(i)
        dx, de = self._deriv_coef(s)
        dt = s_down - s
        return [1.0 + dx * dt, de * dt, s_up, 0.0, 0.0, 0.0, self._scale_next(i), dx, de, 0.0, 0.0, 0.0]

    def rows(self) -> torch.Tensor:
        return torch.tensor([self.row(i) for i in range(len(self.timesteps))], dtype=torch.float32)

    def step(self, model_output, timestep, sample, generator=None, noise=None):
        i = self._index(timestep)
        r = self.row(i)
        if noise is None:
            noise = torch.randn(model_output.shape, generator=generator, device=model_output.device, dtype=model_output.dtype)
        return SchedulerOutput(r[0] * sample + r[1] * model_output + r[2] * noise)


class LMSDiscreteScheduler(_SigmaBase):
    """linear multistep (order 4): x' = x + sum_j c_j d_{t-j}, c_j = integral of the Lagrange basis over
    [sigma_t, sigma_{t+1}] (scipy quad, epsrel 1e-4)."""
    order = 1
    needs_noise = False
    n_hist = 3

    def _reset(self):
        self.derivatives = []

    def lms_coefficient(self, order: int, t: int, current_order: int) -> float:
        from scipy import integrate

        def basis(tau):
            prod = 1.0
            for k in range(order):
                if current_order == k:
                    continue
                prod *= (tau - self.sigmas[t - k]) / (self.sigmas[t - current_order] - self.sigmas[t - k])
            return prod
        return integrate.quad(basis, self.sigmas[t], self.sigmas[t + 1], epsrel=1e-4)[0]

    def _coeffs(self, i: int, order: int = 4):
        o = min(i + 1, order)
        return [self.lms_coefficient(o, i, c) for c in range(o)]

    def row(self, i: int):
        s = float(self.sigmas[i])
        dx, de = self._deriv_coef(s)
        c = self._coeffs(i) + [0.0, 0.0, 0.0]
        return [1.0 + c[0] * dx, c[0] * de, 0.0, c[1], c[2], c[3], self._scale_next(i), dx, de, 0.0, 0.0, 0.0]

    def rows(self) -> torch.Tensor:
        return torch.tensor([self.row(i) for i in range(len(self.timesteps))], dtype=torch.float32)

    def step(self, model_output, timestep, sample, order: int = 4):
        i = self._index(timestep)
        s = float(self.sigmas[i])
        dx, de = self._deriv_coef(s)
        self.derivatives.append(dx * sample + de * model_output)
        if len(self.derivatives) > order:
            self.derivatives.pop(0)
        coeffs = self._coeffs(i, order)
        prev = sample + sum(c * d for c, d in zip(coeffs, reversed(self.derivatives)))
        return SchedulerOutput(prev)


class DDPMScheduler:
    """ancestral sampling with the fixed_small variance: x' = mean(x, out) + sqrt(var) noise (stochastic)."""
    order = 1
    needs_noise = True
    n_hist = 0

    def __init__(self, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", num_train_timesteps=1000,
                 clip_sample=False, prediction_type="epsilon"):
        if beta_schedule != "scaled_linear":
            raise ValueError("only the scaled_linear schedule of the SD model family is implemented")
        if clip_sample:
            raise ValueError("clip_sample=True is not used by the reference and not implemented")
        self.num_train_timesteps = num_train_timesteps
        self.prediction_type = prediction_type
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    def set_timesteps(self, num_inference_steps: int, device=None):
        self.num_inference_steps = num_inference_steps
        ratio = self.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts).to(device)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def _index(self, timestep) -> int:
        return int((self.timesteps.detach().cpu() - int(timestep)).abs().argmin())

    def _coef(self, t: int):
        n = self.num_inference_steps or self.num_train_timesteps
        prev_t = t - self.num_train_timesteps // n
        a_t = float(self.alphas_cumprod[t])
        a_p = float(self.alphas_cumprod[prev_t]) if prev_t >= 0 else 1.0
        b_t, b_p = 1 - a_t, 1 - a_p
        cur_a = a_t / a_p
        cur_b = 1 - cur_a
        if self.prediction_type == "epsilon":
            px, pe = 1 / a_t ** 0.5, -(b_t ** 0.5) / a_t ** 0.5
        elif self.prediction_type == "v_prediction":
            px, pe = a_t ** 0.5, -(b_t ** 0.5)
        else:
            raise ValueError(f"unknown prediction_type {self.prediction_type}")
        c0 = a_p ** 0.5 * cur_b / b_t          # weight of pred_x0
        ct = cur_a ** 0.5 * b_p / b_t          # weight of the current sample
        var = max(b_p / b_t * cur_b, 1e-20)
        return c0 * px + ct, c0 * pe, (var ** 0.5 if t > 0 else 0.0)

    def row(self, i: int):
        cx, ce, cn = self._coef(int(self.timesteps[i]))
        return [cx, ce, cn, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0]

    def rows(self) -> torch.Tensor:
        return torch.tensor([self.row(i) for i in range(len(self.timesteps))], dtype=torch.float32)

    def step(self, model_output, timestep, sample, generator=None, noise=None):
        cx, ce, cn = self._coef(int(timestep))
        if noise is None:
            noise = torch.randn(model_output.shape, generator=generator, device=model_output.device, dtype=model_output.dtype)
        return SchedulerOutput(cx * sample + ce * model_output + cn * noise)


def create_noise_scheduler(scheduler_name: str = "ddpm", prediction_type: str = "epsilon"):
    """model_util.py:230-278 (same names, same constructor arguments)."""
    name = scheduler_name.lower().replace(" ", "_")
    kw = dict(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", num_train_timesteps=1000,
              prediction_type=prediction_type)
    if name == "ddim":
        return DDIMScheduler(clip_sample=False, **kw)
    if name == "ddpm":
        return DDPMScheduler(clip_sample=False, **kw)
    if name == "lms":
        return LMSDiscreteScheduler(**kw)
    if name == "euler_a":
        return EulerAncestralDiscreteScheduler(**kw)
    raise ValueError(f"Unknown scheduler name: {name}")
