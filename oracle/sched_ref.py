"""CPU restatement of the three non-DDIM schedulers the reference can select (model_util.py:247-274):
diffusers 0.20.0 `DDPMScheduler`, `LMSDiscreteScheduler`, `EulerAncestralDiscreteScheduler` with the constructor
arguments used there (scaled-linear betas 0.00085..0.012, 1000 train steps, clip_sample=False).

TEST INFRASTRUCTURE ONLY (tests/, smoke, bench cpu_baseline).  PARITY UNPINNED: diffusers is not vendored in
/root/reference and not installed; the step rules are restated from the published algorithms (Ho et al. 2020 eq. 7
posterior with the "fixed_small" variance; Karras et al. 2022 ancestral Euler as in k-diffusion `sample_euler_ancestral`;
k-diffusion `sample_lms` with order 4 and quad(epsrel=1e-4)), written step-by-step (predict x0, form the update)
rather than as the pre-multiplied coefficient rows of leco_amd/scheduler.py, so the two derivations check each other.
Known answer pinned: sigma_max = 14.6146 for this beta schedule."""
import numpy as np
import torch
from scipy import integrate


def _alphas_cumprod(n=1000, b0=0.00085, b1=0.012):
    betas = torch.linspace(b0 ** 0.5, b1 ** 0.5, n, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


class SigmaSchedule:
    def __init__(self, n_steps, n_train=1000):
        ac = _alphas_cumprod(n_train)
        all_sig = (((1 - ac) / ac) ** 0.5).numpy()
        self.timesteps = np.linspace(0, n_train - 1, n_steps, dtype=float)[::-1].copy()
        self.sigmas = np.concatenate([np.interp(self.timesteps, np.arange(n_train), all_sig), [0.0]]).astype(np.float32)
        self.init_noise_sigma = float(self.sigmas.max())

    def scale(self, i):
        return 1.0 / float(self.sigmas[i] ** 2 + 1) ** 0.5


def pred_x0_sigma(x, out, sigma, prediction_type):
    if prediction_type == "epsilon":
        return x - sigma * out
    return out * (-sigma / (sigma ** 2 + 1) ** 0.5) + x / (sigma ** 2 + 1)


def euler_a_step(sch: SigmaSchedule, i, x, out, noise, prediction_type="epsilon"):
    s_from, s_to = float(sch.sigmas[i]), float(sch.sigmas[i + 1])
    x0 = pred_x0_sigma(x, out, s_from, prediction_type)
    s_up = (s_to ** 2 * (s_from ** 2 - s_to ** 2) / s_from ** 2) ** 0.5
    s_down = (s_to ** 2 - s_up ** 2) ** 0.5
    d = (x - x0) / s_from
    return x + d * (s_down - s_from) + noise * s_up


def lms_step(sch: SigmaSchedule, i, x, out, derivatives, prediction_type="epsilon", order=4):
    """`derivatives`: list of previous derivatives (oldest first), mutated like the scheduler's own state."""
    s = float(sch.sigmas[i])
    x0 = pred_x0_sigma(x, out, s, prediction_type)
    derivatives.append((x - x0) / s)
    if len(derivatives) > order:
        derivatives.pop(0)
    o = min(i + 1, order)

    def coeff(cur):
        def f(tau):
            p = 1.0
            for k in range(o):
                if k != cur:
                    p *= (tau - sch.sigmas[i - k]) / (sch.sigmas[i - cur] - sch.sigmas[i - k])
            return p
        return integrate.quad(f, sch.sigmas[i], sch.sigmas[i + 1], epsrel=1e-4)[0]
    return x + sum(coeff(c) * d for c, d in zip(range(o), reversed(derivatives)))


def ddpm_step(t, n_steps, x, out, noise, prediction_type="epsilon", n_train=1000):
    ac = _alphas_cumprod(n_train)
    prev_t = t - n_train // n_steps
    a_t = float(ac[t])
    a_p = float(ac[prev_t]) if prev_t >= 0 else 1.0
    b_t, b_p = 1 - a_t, 1 - a_p
    cur_a = a_t / a_p
    cur_b = 1 - cur_a
    if prediction_type == "epsilon":
        x0 = (x - b_t ** 0.5 * out) / a_t ** 0.5
    else:
        x0 = a_t ** 0.5 * x - b_t ** 0.5 * out
    mean = (a_p ** 0.5 * cur_b / b_t) * x0 + (cur_a ** 0.5 * b_p / b_t) * x
    var = max(b_p / b_t * cur_b, 1e-20)
    return mean + (var ** 0.5 * noise if t > 0 else 0.0)


# ---- duck-typed adapters (set_timesteps / timesteps / scale_model_input / step().prev_sample) for oracle/step_ref.py
class _Out:
    def __init__(self, prev_sample):
        self.prev_sample = prev_sample


class _SigmaRef:
    def __init__(self, prediction_type="epsilon", noises=None):
        self.prediction_type, self.noises = prediction_type, list(noises or [])
        self.set_timesteps(1000)

    def set_timesteps(self, n):
        self.n = n
        self.sch = SigmaSchedule(n)
        self.timesteps = torch.from_numpy(self.sch.timesteps.astype(np.float32))
        self.init_noise_sigma = self.sch.init_noise_sigma
        self.derivatives = []

    def _i(self, t):
        return int((self.timesteps.double() - float(t)).abs().argmin())

    def scale_model_input(self, x, t):
        return x * self.sch.scale(self._i(t))


class LMSRef(_SigmaRef):
    def step(self, out, t, x):
        return _Out(lms_step(self.sch, self._i(t), x, out, self.derivatives, self.prediction_type))


class EulerARef(_SigmaRef):
    def step(self, out, t, x):
        noise = self.noises.pop(0).reshape(x.shape).to(x.dtype)
        return _Out(euler_a_step(self.sch, self._i(t), x, out, noise, self.prediction_type))


class DDPMRef:
    def __init__(self, prediction_type="epsilon", noises=None):
        self.prediction_type, self.noises = prediction_type, list(noises or [])
        self.init_noise_sigma = 1.0
        self.set_timesteps(1000)

    def set_timesteps(self, n):
        self.n = n
        self.timesteps = torch.from_numpy((np.arange(0, n) * (1000 // n)).round()[::-1].copy().astype(np.int64))

    def scale_model_input(self, x, t):
        return x

    def step(self, out, t, x):
        noise = self.noises.pop(0).reshape(x.shape).to(x.dtype)
        return _Out(ddpm_step(int(t), self.n, x, out, noise, self.prediction_type))
