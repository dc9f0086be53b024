"""DDIM noise scheduler with the duck-type the reference uses (``set_timesteps``, ``timesteps``,
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


def create_noise_scheduler(scheduler_name: str = "ddpm", prediction_type: str = "epsilon"):
    """model_util.py:230-278.  Only DDIM is on the MI355X hot path; the other three names the
    reference accepts (ddpm / lms / euler_a) are outside every benchmarked configuration."""
    name = scheduler_name.lower().replace(" ", "_")
    if name == "ddim":
        return DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                             num_train_timesteps=1000, clip_sample=False, prediction_type=prediction_type)
    if name in ("ddpm", "lms", "euler_a"):
        raise NotImplementedError(f"noise scheduler '{name}' is not implemented on the MI355X path (only 'ddim')")
    raise ValueError(f"Unknown scheduler name: {name}")
