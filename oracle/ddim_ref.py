"""ORACLE (test infrastructure only) -- numpy/torch restatement of
``diffusers==0.20.0`` ``DDIMScheduler`` as the reference constructs and uses it.

PARITY UNPINNED: the scheduler lives in the un-vendored dependency diffusers==0.20.0
(requirements.txt:1); this restates its published algorithm.
Anchors: ctor args at model_util.py:239-246; call sites train_lora.py:143-145,195-199,
train_util.py:55,153,184,190.  The timestep tables [980..0] / [999..0] and the closed-form
alphas_cumprod are checked in tests/test_host.py (scheduler tests).
"""
import numpy as np
import torch


class _Step:
    def __init__(self, prev_sample, pred_original_sample):
        self.prev_sample = prev_sample
        self.pred_original_sample = pred_original_sample


class DDIMSchedulerRef:
    """beta_schedule='scaled_linear', clip_sample=False, set_alpha_to_one=True,
    steps_offset=0, timestep_spacing='leading', eta=0."""

    def __init__(self, beta_start=0.00085, beta_end=0.012, num_train_timesteps=1000,
                 prediction_type="epsilon"):
        self.num_train_timesteps = num_train_timesteps
        self.prediction_type = prediction_type
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps,
                               dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0)
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = num_inference_steps
        ratio = self.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts).to(device)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def step(self, model_output, timestep, sample):
        t = int(timestep)
        prev_t = t - self.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        b_t = 1 - a_t
        if self.prediction_type == "epsilon":
            x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
            eps = model_output
        elif self.prediction_type == "v_prediction":
            x0 = (a_t ** 0.5) * sample - (b_t ** 0.5) * model_output
            eps = (a_t ** 0.5) * model_output + (b_t ** 0.5) * sample
        else:
            raise ValueError(self.prediction_type)
        direction = (1 - a_p) ** 0.5 * eps
        prev = a_p ** 0.5 * x0 + direction
        return _Step(prev, x0)
