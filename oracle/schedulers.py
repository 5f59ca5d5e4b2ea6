"""fp32 CPU restatement of the three schedulers on the reference's hot path.

TEST INFRASTRUCTURE (see oracle/__init__.py) -- parity unpinned.

* ``DDIMOracle``      <- ``src/schedulers.py:190-192`` (empty subclass of diffusers DDIMScheduler)
* ``DPMSolverOracle`` <- ``src/schedulers.py:12-187`` (convert_model_output + step are repo code and
                         restated literally; the update formulas are diffusers 0.32.1's
                         DPMSolverMultistepScheduler)
* ``LCMOracle``       <- ``src/schedulers.py:195-197`` (empty subclass of diffusers LCMScheduler)

All tables are fp32 exactly as diffusers builds them
(``betas = linspace(sqrt(b0), sqrt(b1), T, float32)**2``; ``alphas_cumprod = cumprod(1-betas)``);
every scalar coefficient is a 0-d fp32 tensor as upstream, so the arithmetic order is the
published one.  The SD-1.5 checkpoint's scheduler config (PNDM: steps_offset=1,
set_alpha_to_one=False, timestep_spacing="leading", prediction_type="epsilon") is what
``BaseMethod.setup_scheduler`` forwards through ``from_config``
(``src/experiments/base_experiment.py:66-72``), hence the defaults below.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

# what runwayml/stable-diffusion-v1-5/scheduler/scheduler_config.json holds
SD15_SCHEDULER_CONFIG = dict(
    num_train_timesteps=1000,
    beta_start=0.00085,
    beta_end=0.012,
    beta_schedule="scaled_linear",
    set_alpha_to_one=False,
    skip_prk_steps=True,
    steps_offset=1,
    clip_sample=False,
    timestep_spacing="leading",
    prediction_type="epsilon",
)


def _alphas_cumprod(cfg) -> torch.Tensor:
    betas = torch.linspace(cfg["beta_start"] ** 0.5, cfg["beta_end"] ** 0.5,
                           cfg["num_train_timesteps"], dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


class _Base:
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, **config):
        cfg = dict(SD15_SCHEDULER_CONFIG)
        cfg.update(config)
        self.config = cfg
        self.alphas_cumprod = _alphas_cumprod(cfg)
        self.timesteps = None
        self.num_inference_steps = None

    @classmethod
    def from_config(cls, config, **overrides):
        cfg = dict(config)
        cfg.update(overrides)
        return cls(**cfg)

    def scale_model_input(self, sample, t=None):
        return sample


class DDIMOracle(_Base):
    """diffusers DDIMScheduler, eta=0 (reference passes eta=0.0: src/models.py:43,185)."""

    def __init__(self, **config):
        super().__init__(**config)
        self.final_alpha_cumprod = (torch.tensor(1.0) if self.config.get("set_alpha_to_one", True)
                                    else self.alphas_cumprod[0])

    def set_timesteps(self, num_inference_steps: int, device=None):
        T = self.config["num_train_timesteps"]
        self.num_inference_steps = num_inference_steps
        spacing = self.config.get("timestep_spacing", "leading")
        if spacing == "leading":
            step_ratio = T // num_inference_steps
            ts = (np.arange(0, num_inference_steps) * step_ratio).round()[::-1].copy().astype(np.int64)
            ts += self.config.get("steps_offset", 0)
        elif spacing == "linspace":
            ts = np.linspace(0, T - 1, num_inference_steps).round()[::-1].copy().astype(np.int64)
        elif spacing == "trailing":
            step_ratio = T / num_inference_steps
            ts = np.round(np.arange(T, 0, -step_ratio)).astype(np.int64) - 1
        else:
            raise ValueError(spacing)
        self.timesteps = torch.from_numpy(ts)

    def step(self, model_output, timestep, sample, eta: float = 0.0, generator=None, return_dict=False):
        assert eta == 0.0
        t = int(timestep)
        prev_t = t - self.config["num_train_timesteps"] // self.num_inference_steps
        alpha_prod_t = self.alphas_cumprod[t]
        alpha_prod_t_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        beta_prod_t = 1 - alpha_prod_t
        pred_original_sample = (sample - beta_prod_t ** 0.5 * model_output) / alpha_prod_t ** 0.5
        pred_epsilon = model_output
        pred_sample_direction = (1 - alpha_prod_t_prev) ** 0.5 * pred_epsilon
        prev_sample = alpha_prod_t_prev ** 0.5 * pred_original_sample + pred_sample_direction
        return (prev_sample, pred_original_sample)


class DPMSolverOracle(_Base):
    """``DPMSolverScheduler`` of the reference (src/schedulers.py:12-187).

    Appendix-B quirk #2: the reference's ``step`` unpacks two values from
    ``convert_model_output`` (src/schedulers.py:127) but its ``++`` branch returns ONE tensor
    (src/schedulers.py:61).  The intended behaviour -- restated here -- is
    ``model_output := x0_pred`` for ``++`` and ``(epsilon, x0_pred)`` for plain ``dpmsolver``.
    """

    def __init__(self, **config):
        config.setdefault("solver_order", 2)
        config.setdefault("algorithm_type", "dpmsolver++")
        config.setdefault("solver_type", "midpoint")
        config.setdefault("lower_order_final", True)
        config.setdefault("euler_at_final", False)
        config.setdefault("final_sigmas_type", "zero")
        super().__init__(**config)
        if (self.config["algorithm_type"] in ("dpmsolver", "sde-dpmsolver")
                and self.config["final_sigmas_type"] == "zero"):
            raise ValueError("`final_sigmas_type` zero is not supported for `algorithm_type` "
                             f"{self.config['algorithm_type']}. Please choose `sigma_min` instead.")
        self.model_outputs = [None] * self.config["solver_order"]
        self.lower_order_nums = 0
        self._step_index = None

    @property
    def step_index(self):
        return self._step_index

    def set_timesteps(self, num_inference_steps: int = None, device=None, timesteps=None):
        """``timesteps``: custom schedule (diffusers 0.32.1 DPMSolverMultistepScheduler.set_timesteps;
        used by the two-scheduler pipeline, ``src/models.py:488-492``) [upstream-recall]."""
        T = self.config["num_train_timesteps"]
        last_timestep = T            # lambda_min_clipped = -inf
        spacing = self.config.get("timestep_spacing", "linspace")
        if timesteps is not None:
            ts = np.array([int(t) for t in timesteps]).astype(np.int64)
        elif spacing == "linspace":
            ts = np.linspace(0, last_timestep - 1, num_inference_steps + 1).round()[::-1][:-1].copy().astype(np.int64)
        elif spacing == "leading":
            step_ratio = last_timestep // (num_inference_steps + 1)
            ts = (np.arange(0, num_inference_steps + 1) * step_ratio).round()[::-1][:-1].copy().astype(np.int64)
            ts += self.config.get("steps_offset", 0)
        elif spacing == "trailing":
            step_ratio = T / num_inference_steps
            ts = np.arange(last_timestep, 0, -step_ratio).round().copy().astype(np.int64) - 1
        else:
            raise ValueError(spacing)
        ac = self.alphas_cumprod.numpy()
        sigmas = np.array(((1 - ac) / ac) ** 0.5)
        sigmas = np.interp(ts, np.arange(0, len(sigmas)), sigmas)
        if self.config["final_sigmas_type"] == "sigma_min":
            sigma_last = ((1 - ac[0]) / ac[0]) ** 0.5
        elif self.config["final_sigmas_type"] == "zero":
            sigma_last = 0
        else:
            raise ValueError(self.config["final_sigmas_type"])
        self.sigmas = torch.from_numpy(np.concatenate([sigmas, [sigma_last]]).astype(np.float32))
        self.timesteps = torch.from_numpy(ts)
        self.num_inference_steps = len(ts)
        self.model_outputs = [None] * self.config["solver_order"]
        self.lower_order_nums = 0
        self._step_index = None

    @staticmethod
    def _sigma_to_alpha_sigma_t(sigma):
        alpha_t = 1 / ((sigma ** 2 + 1) ** 0.5)
        sigma_t = sigma * alpha_t
        return alpha_t, sigma_t

    def _init_step_index(self, timestep):
        idx = (self.timesteps == int(timestep)).nonzero()
        pos = 1 if len(idx) > 1 else 0
        self._step_index = idx[pos].item()

    # src/schedulers.py:14-96 (epsilon prediction, no thresholding)
    def convert_model_output(self, model_output, sample):
        # The variant pipelines call this on a scheduler that has not stepped yet (src/models.py:603-611):
        # the reference then indexes ``sigmas[None]`` and fails on a shape mismatch.  Restated with index 0
        # so that the hand-off runs; with the shipped configs the entry is shifted out before it is used.
        sigma = self.sigmas[self.step_index if self.step_index is not None else 0]
        alpha_t, sigma_t = self._sigma_to_alpha_sigma_t(sigma)
        x0_pred = (sample - sigma_t * model_output) / alpha_t
        if self.config["algorithm_type"] in ("dpmsolver++", "sde-dpmsolver++"):
            return x0_pred, x0_pred          # intended behaviour, see class docstring
        return model_output, x0_pred         # src/schedulers.py:92-96

    def _lambdas(self, *sig):
        out = []
        for s in sig:
            a, st = self._sigma_to_alpha_sigma_t(s)
            out.append((a, st, torch.log(a) - torch.log(st)))
        return out

    def dpm_solver_first_order_update(self, model_output, sample, noise=None):
        (alpha_t, sigma_t, lambda_t), (alpha_s, sigma_s, lambda_s) = self._lambdas(
            self.sigmas[self.step_index + 1], self.sigmas[self.step_index])
        h = lambda_t - lambda_s
        alg = self.config["algorithm_type"]
        if alg == "dpmsolver++":
            return (sigma_t / sigma_s) * sample - (alpha_t * (torch.exp(-h) - 1.0)) * model_output
        if alg == "dpmsolver":
            return (alpha_t / alpha_s) * sample - (sigma_t * (torch.exp(h) - 1.0)) * model_output
        if alg == "sde-dpmsolver++":     # diffusers 0.32.1 DPMSolverMultistepScheduler [upstream-recall]
            return ((sigma_t / sigma_s * torch.exp(-h)) * sample + (alpha_t * (1 - torch.exp(-2.0 * h))) * model_output
                    + sigma_t * torch.sqrt(1.0 - torch.exp(-2.0 * h)) * noise)
        assert alg == "sde-dpmsolver"
        return ((alpha_t / alpha_s) * sample - 2.0 * (sigma_t * (torch.exp(h) - 1.0)) * model_output
                + sigma_t * torch.sqrt(torch.exp(2.0 * h) - 1.0) * noise)

    def multistep_dpm_solver_second_order_update(self, model_output_list, sample, noise=None):
        (alpha_t, sigma_t, lambda_t), (alpha_s0, sigma_s0, lambda_s0), (_, _, lambda_s1) = self._lambdas(
            self.sigmas[self.step_index + 1], self.sigmas[self.step_index], self.sigmas[self.step_index - 1])
        m0, m1 = model_output_list[-1], model_output_list[-2]
        h, h_0 = lambda_t - lambda_s0, lambda_s0 - lambda_s1
        r0 = h_0 / h
        D0, D1 = m0, (1.0 / r0) * (m0 - m1)
        assert self.config["solver_type"] == "midpoint"
        alg = self.config["algorithm_type"]
        if alg == "dpmsolver++":
            return ((sigma_t / sigma_s0) * sample - (alpha_t * (torch.exp(-h) - 1.0)) * D0
                    - 0.5 * (alpha_t * (torch.exp(-h) - 1.0)) * D1)
        if alg == "dpmsolver":
            return ((alpha_t / alpha_s0) * sample - (sigma_t * (torch.exp(h) - 1.0)) * D0
                    - 0.5 * (sigma_t * (torch.exp(h) - 1.0)) * D1)
        if alg == "sde-dpmsolver++":
            return ((sigma_t / sigma_s0 * torch.exp(-h)) * sample + (alpha_t * (1 - torch.exp(-2.0 * h))) * D0
                    + 0.5 * (alpha_t * (1 - torch.exp(-2.0 * h))) * D1
                    + sigma_t * torch.sqrt(1.0 - torch.exp(-2.0 * h)) * noise)
        assert alg == "sde-dpmsolver"
        return ((alpha_t / alpha_s0) * sample - 2.0 * (sigma_t * (torch.exp(h) - 1.0)) * D0
                - (sigma_t * (torch.exp(h) - 1.0)) * D1 + sigma_t * torch.sqrt(torch.exp(2.0 * h) - 1.0) * noise)

    def multistep_dpm_solver_third_order_update(self, model_output_list, sample, noise=None):
        (alpha_t, sigma_t, lambda_t), (alpha_s0, sigma_s0, lambda_s0), (_, _, lambda_s1), (_, _, lambda_s2) = \
            self._lambdas(self.sigmas[self.step_index + 1], self.sigmas[self.step_index],
                          self.sigmas[self.step_index - 1], self.sigmas[self.step_index - 2])
        m0, m1, m2 = model_output_list[-1], model_output_list[-2], model_output_list[-3]
        h, h_0, h_1 = lambda_t - lambda_s0, lambda_s0 - lambda_s1, lambda_s1 - lambda_s2
        r0, r1 = h_0 / h, h_1 / h
        D0 = m0
        D1_0, D1_1 = (1.0 / r0) * (m0 - m1), (1.0 / r1) * (m1 - m2)
        D1 = D1_0 + (r0 / (r0 + r1)) * (D1_0 - D1_1)
        D2 = (1.0 / (r0 + r1)) * (D1_0 - D1_1)
        alg = self.config["algorithm_type"]
        if alg == "dpmsolver++":
            return ((sigma_t / sigma_s0) * sample - (alpha_t * (torch.exp(-h) - 1.0)) * D0
                    + (alpha_t * ((torch.exp(-h) - 1.0) / h + 1.0)) * D1
                    - (alpha_t * ((torch.exp(-h) - 1.0 + h) / h ** 2 - 0.5)) * D2)
        if alg == "sde-dpmsolver++":
            return ((sigma_t / sigma_s0 * torch.exp(-h)) * sample + (alpha_t * (1.0 - torch.exp(-2.0 * h))) * D0
                    + (alpha_t * ((1.0 - torch.exp(-2.0 * h)) / (-2.0 * h) + 1.0)) * D1
                    + (alpha_t * ((1.0 - torch.exp(-2.0 * h) - 2.0 * h) / (2.0 * h) ** 2 - 0.5)) * D2
                    + sigma_t * torch.sqrt(1.0 - torch.exp(-2.0 * h)) * noise)
        assert alg == "dpmsolver", "sde-dpmsolver has no third-order update upstream"
        return ((alpha_t / alpha_s0) * sample - (sigma_t * (torch.exp(h) - 1.0)) * D0
                - (sigma_t * ((torch.exp(h) - 1.0) / h - 1.0)) * D1
                - (sigma_t * ((torch.exp(h) - 1.0 - h) / h ** 2 - 0.5)) * D2)

    # src/schedulers.py:98-187
    def step(self, model_output, timestep, sample, generator=None, variance_noise=None, return_dict=False):
        if self.step_index is None:
            self._init_step_index(timestep)
        n = len(self.timesteps)
        lower_order_final = (self.step_index == n - 1) and (
            self.config["euler_at_final"]
            or (self.config["lower_order_final"] and n < 15)
            or self.config["final_sigmas_type"] == "zero")
        lower_order_second = ((self.step_index == n - 2) and self.config["lower_order_final"] and n < 15)

        model_output, x0_pred = self.convert_model_output(model_output, sample=sample)
        for i in range(self.config["solver_order"] - 1):
            self.model_outputs[i] = self.model_outputs[i + 1]
        self.model_outputs[-1] = model_output
        sample = sample.to(torch.float32)
        # src/schedulers.py:134-147
        if self.config["algorithm_type"] in ["sde-dpmsolver", "sde-dpmsolver++"] and variance_noise is None:
            noise = torch.randn(model_output.shape, generator=generator, dtype=torch.float32)
        elif self.config["algorithm_type"] in ["sde-dpmsolver", "sde-dpmsolver++"]:
            noise = variance_noise.to(dtype=torch.float32)
        else:
            noise = None
        so = self.config["solver_order"]
        if so == 1 or self.lower_order_nums < 1 or lower_order_final:
            prev_sample = self.dpm_solver_first_order_update(model_output, sample=sample, noise=noise)
        elif so == 2 or self.lower_order_nums < 2 or lower_order_second:
            prev_sample = self.multistep_dpm_solver_second_order_update(self.model_outputs, sample=sample, noise=noise)
        else:
            prev_sample = self.multistep_dpm_solver_third_order_update(self.model_outputs, sample=sample, noise=noise)
        if self.lower_order_nums < so:
            self.lower_order_nums += 1
        self._step_index += 1
        return (prev_sample, x0_pred)


class LCMOracle(_Base):
    """diffusers LCMScheduler."""

    def __init__(self, **config):
        config.setdefault("original_inference_steps", 50)
        config.setdefault("timestep_scaling", 10.0)
        super().__init__(**config)
        self.final_alpha_cumprod = (torch.tensor(1.0) if self.config.get("set_alpha_to_one", True)
                                    else self.alphas_cumprod[0])
        self._step_index = None

    def set_timesteps(self, num_inference_steps: int, device=None):
        T = self.config["num_train_timesteps"]
        original_steps = self.config["original_inference_steps"]
        k = T // original_steps
        origin = np.asarray(list(range(1, int(original_steps) + 1))) * k - 1
        origin = origin[::-1].copy()
        idx = np.floor(np.linspace(0, len(origin), num=num_inference_steps, endpoint=False)).astype(np.int64)
        self.timesteps = torch.from_numpy(origin[idx].astype(np.int64))
        self.num_inference_steps = num_inference_steps
        self._step_index = None

    def step(self, model_output, timestep, sample, generator=None, noise: Optional[torch.Tensor] = None,
             return_dict=False):
        if self._step_index is None:
            self._step_index = (self.timesteps == int(timestep)).nonzero()[0].item()
        t = int(timestep)
        prev_i = self._step_index + 1
        prev_t = int(self.timesteps[prev_i]) if prev_i < len(self.timesteps) else t
        alpha_prod_t = self.alphas_cumprod[t]
        alpha_prod_t_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        beta_prod_t = 1 - alpha_prod_t
        beta_prod_t_prev = 1 - alpha_prod_t_prev
        sigma_data = 0.5
        scaled_t = t * self.config["timestep_scaling"]
        c_skip = sigma_data ** 2 / (scaled_t ** 2 + sigma_data ** 2)
        c_out = scaled_t / (scaled_t ** 2 + sigma_data ** 2) ** 0.5
        x0 = (sample - beta_prod_t.sqrt() * model_output) / alpha_prod_t.sqrt()
        denoised = c_out * x0 + c_skip * sample
        if self._step_index != self.num_inference_steps - 1:
            if noise is None:
                noise = torch.randn(model_output.shape, generator=generator, dtype=denoised.dtype)
            prev_sample = alpha_prod_t_prev.sqrt() * denoised + beta_prod_t_prev.sqrt() * noise
        else:
            prev_sample = denoised
        self._step_index += 1
        return (prev_sample, denoised)


class PNDMOracle(_Base):
    """diffusers PNDMScheduler with ``skip_prk_steps=True`` (PLMS only) -- the checkpoint's own
    scheduler, which the reference's ``deep_cache`` / ``default`` methods run because they never swap
    it (``src/experiments/deep_cache.py:17-18``).  N steps -> N+1 UNet calls, the
    second timestep is duplicated (A.7: 981, 961, 961, 941, ...)."""

    def __init__(self, **config):
        config.setdefault("skip_prk_steps", True)
        super().__init__(**config)
        self.final_alpha_cumprod = (torch.tensor(1.0) if self.config.get("set_alpha_to_one", False)
                                    else self.alphas_cumprod[0])
        self.ets = []
        self.counter = 0
        self.cur_sample = None

    def set_timesteps(self, num_inference_steps: int, device=None):
        T = self.config["num_train_timesteps"]
        self.num_inference_steps = num_inference_steps
        assert self.config.get("timestep_spacing", "leading") == "leading" and self.config["skip_prk_steps"]
        step_ratio = T // num_inference_steps
        _ts = (np.arange(0, num_inference_steps) * step_ratio).round() + self.config.get("steps_offset", 0)
        plms = np.concatenate([_ts[:-1], _ts[-2:-1], _ts[-1:]])[::-1].copy()
        self.timesteps = torch.from_numpy(plms.astype(np.int64))
        self.ets, self.counter, self.cur_sample = [], 0, None

    def _get_prev_sample(self, sample, timestep, prev_timestep, model_output):
        alpha_prod_t = self.alphas_cumprod[timestep]
        alpha_prod_t_prev = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        beta_prod_t = 1 - alpha_prod_t
        beta_prod_t_prev = 1 - alpha_prod_t_prev
        sample_coeff = (alpha_prod_t_prev / alpha_prod_t) ** 0.5
        denom = alpha_prod_t * beta_prod_t_prev ** 0.5 + (alpha_prod_t * beta_prod_t * alpha_prod_t_prev) ** 0.5
        return sample_coeff * sample - (alpha_prod_t_prev - alpha_prod_t) * model_output / denom

    def step(self, model_output, timestep, sample, return_dict=False, **kwargs):
        timestep = int(timestep)
        ratio = self.config["num_train_timesteps"] // self.num_inference_steps
        prev_timestep = timestep - ratio
        if self.counter != 1:
            self.ets = self.ets[-3:]
            self.ets.append(model_output)
        else:
            prev_timestep = timestep
            timestep = timestep + ratio
        if len(self.ets) == 1 and self.counter == 0:
            self.cur_sample = sample
        elif len(self.ets) == 1 and self.counter == 1:
            model_output = (model_output + self.ets[-1]) / 2
            sample = self.cur_sample
            self.cur_sample = None
        elif len(self.ets) == 2:
            model_output = (3 * self.ets[-1] - self.ets[-2]) / 2
        elif len(self.ets) == 3:
            model_output = (23 * self.ets[-1] - 16 * self.ets[-2] + 5 * self.ets[-3]) / 12
        else:
            model_output = (1 / 24) * (55 * self.ets[-1] - 59 * self.ets[-2] + 37 * self.ets[-3] - 9 * self.ets[-4])
        prev_sample = self._get_prev_sample(sample, timestep, prev_timestep, model_output)
        self.counter += 1
        return (prev_sample,)
