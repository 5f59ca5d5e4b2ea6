"""The reference's three scheduler plugins, with ``step`` as ONE fused HIP launch.

Same registry keys, class names and call protocol as ``src/schedulers.py`` of the reference:

* ``"dpm_solver_scheduler"`` -> ``DPMSolverScheduler``  (``src/schedulers.py:12-187``)
* ``"ddim_scheduler"``       -> ``DDIMSchedulerMy``     (``src/schedulers.py:190-192``)
* ``"lcm_scheduler"``        -> ``LCMScheduler``        (``src/schedulers.py:195-197``)
* ``"pndm_scheduler"``       -> ``PNDMScheduler``       (the checkpoint's own scheduler)

Protocol kept: ``from_config(config, **overrides)``, ``.config``, ``.order``,
``.init_noise_sigma``, ``set_timesteps(n, device=)``, ``.timesteps``, ``scale_model_input``,
``step(model_output, timestep, sample, ..., return_dict=False) -> (prev_sample, x0_pred)``,
and for DPM ``convert_model_output`` / ``.model_outputs``.

MI355X-first design: the host keeps the fp32 alpha-bar / sigma tables exactly as diffusers builds
them, derives the step's scalar coefficients in fp64, and launches ``sd_sched_step`` once: CFG
combine + x0 + multistep update + history write in a single elementwise kernel with no host sync
(the reference does ~10 elementwise launches and an ``alphas_cumprod[t]`` device->host sync per
step).  ``step_fused`` is the loop's entry point (takes the raw 2B-batch UNet output and the
guidance scale); ``step`` is the drop-in signature and uses the same kernel without CFG.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Optional

import numpy as np
import torch

from . import _lib
from . import dist as sdist
from .registry import schedulers_registry

# runwayml/stable-diffusion-v1-5 scheduler/scheduler_config.json (PNDM) + the defaults PNDM stores;
# this is what ``from_config(self.model.scheduler.config)`` receives
# (src/experiments/base_experiment.py:66-72).
SD15_SCHEDULER_CONFIG = dict(
    num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
    trained_betas=None, skip_prk_steps=True, set_alpha_to_one=False, prediction_type="epsilon",
    timestep_spacing="leading", steps_offset=1, clip_sample=False,
)


class SchedulerConfig(dict):
    """dict with attribute access (diffusers FrozenDict behaviour the reference relies on)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None


def _alphas_cumprod_fp32(cfg) -> np.ndarray:
    if cfg.get("trained_betas") is not None:
        betas = torch.tensor(cfg["trained_betas"], dtype=torch.float32)
    elif cfg["beta_schedule"] == "scaled_linear":
        betas = torch.linspace(cfg["beta_start"] ** 0.5, cfg["beta_end"] ** 0.5, cfg["num_train_timesteps"],
                               dtype=torch.float32) ** 2
    elif cfg["beta_schedule"] == "linear":
        betas = torch.linspace(cfg["beta_start"], cfg["beta_end"], cfg["num_train_timesteps"], dtype=torch.float32)
    else:
        raise NotImplementedError(f"beta_schedule {cfg['beta_schedule']}")
    return torch.cumprod(1.0 - betas, dim=0).numpy()


class _FusedStepScheduler:
    order = 1
    init_noise_sigma = 1.0
    _own_defaults: dict = {}
    _accepted: tuple = ()

    def __init__(self, **kwargs):
        cfg = dict(self._own_defaults)
        cfg.update(kwargs)
        self.config = SchedulerConfig(cfg)
        self.alphas_cumprod = _alphas_cumprod_fp32(self.config)      # fp32 table, as upstream
        self.timesteps: Optional[torch.Tensor] = None
        self.num_inference_steps: Optional[int] = None
        self._timesteps_list: List[int] = []
        self._step_index: Optional[int] = None

    # diffusers ConfigMixin.from_config: keep the keys this class accepts, apply overrides
    @classmethod
    def from_config(cls, config, **overrides):
        merged = dict(config)
        merged.update(overrides)
        return cls(**{k: v for k, v in merged.items() if k in cls._accepted})

    @property
    def step_index(self):
        return self._step_index

    def scale_model_input(self, sample, timestep=None):
        return sample

    def _set(self, ts: np.ndarray, device=None):
        self._timesteps_list = [int(t) for t in ts]
        self.timesteps = torch.from_numpy(np.asarray(ts, dtype=np.int64)).to(device) if device is not None \
            else torch.from_numpy(np.asarray(ts, dtype=np.int64))
        self.num_inference_steps = len(ts)
        self._step_index = None

    def _index_of(self, timestep) -> int:
        t = int(timestep)
        idx = [i for i, v in enumerate(self._timesteps_list) if v == t]
        if not idx:
            raise ValueError(f"timestep {t} is not in the schedule")
        return idx[1] if len(idx) > 1 else idx[0]

    # --- the single fused launch ------------------------------------------------------------
    @staticmethod
    def _launch(eps, cfg, guidance, x, m1, m2, noise, coef, want_y2=True, want_m=False, m3=None):
        """coef = (px, pe, p1, p2, pn, yx, ye, mx, me[, p3]) -- see include/sd_hip.h::sd_sched_step."""
        lib = _lib.load()
        n = x.numel()
        prev = torch.empty_like(x)
        y2 = torch.empty_like(x) if want_y2 else None
        mo = torch.empty_like(x) if want_m else None
        c10 = [float(v) for v in coef] + [0.0] * (10 - len(coef))
        carr = (C.c_float * 10)(*c10)
        _lib.check(lib.sd_sched_step(_lib.current_stream(), eps.data_ptr(), int(cfg), float(guidance), x.data_ptr(),
                                     _lib.ptr(m1), _lib.ptr(m2), _lib.ptr(m3), _lib.ptr(noise), prev.data_ptr(),
                                     _lib.ptr(y2), _lib.ptr(mo), carr, n), "sd_sched_step")
        return prev, y2, mo

    @staticmethod
    def _prep(t: torch.Tensor) -> torch.Tensor:
        if not t.is_cuda:
            raise _lib.SdHipError("scheduler.step runs as a HIP kernel: tensors must live on the GPU")
        return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()

    def step(self, model_output, timestep, sample, eta: float = 0.0, generator=None, variance_noise=None,
             return_dict: bool = False, **kwargs):
        """Drop-in ``scheduler.step`` (CFG already combined by the caller, src/models.py:253)."""
        out_dtype = model_output.dtype
        kw = {"variance_noise": variance_noise} if variance_noise is not None else {}
        prev, x0 = self.step_fused(model_output, 0.0, sample, timestep, cfg=False, eta=eta, generator=generator, **kw)
        return (prev.to(out_dtype), x0.to(out_dtype))


@schedulers_registry.add_to_registry("ddim_scheduler")
class DDIMSchedulerMy(_FusedStepScheduler):
    """diffusers ``DDIMScheduler`` (eta = 0 path; the reference never passes another eta,
    ``src/models.py:43,185``).  A.6.1."""
    _own_defaults = dict(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                         trained_betas=None, clip_sample=True, set_alpha_to_one=True, steps_offset=0,
                         prediction_type="epsilon", thresholding=False, timestep_spacing="leading")
    _accepted = tuple(_own_defaults)

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        if self.config.prediction_type != "epsilon" or self.config.clip_sample or self.config.thresholding:
            raise NotImplementedError("only epsilon prediction without clipping is on the SD-1.5 hot path")
        self.final_alpha_cumprod = 1.0 if self.config.set_alpha_to_one else float(self.alphas_cumprod[0])

    def set_timesteps(self, num_inference_steps: int, device=None):
        T = self.config.num_train_timesteps
        if num_inference_steps > T:
            raise ValueError("num_inference_steps exceeds num_train_timesteps")
        sp = self.config.timestep_spacing
        if sp == "leading":
            ratio = T // num_inference_steps
            ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64) + self.config.steps_offset
        elif sp == "linspace":
            ts = np.linspace(0, T - 1, num_inference_steps).round()[::-1].copy().astype(np.int64)
        elif sp == "trailing":
            ts = np.round(np.arange(T, 0, -T / num_inference_steps)).astype(np.int64) - 1
        else:
            raise ValueError(sp)
        self._set(ts, device)

    def coefficients(self, timestep: int):
        """(c_x, c_e, d_x, d_e): prev = c_x*x + c_e*eps, x0 = d_x*x + d_e*eps (A.7 KATs)."""
        t = int(timestep)
        prev_t = t - self.config.num_train_timesteps // self.num_inference_steps
        a = float(self.alphas_cumprod[t])
        ap = float(self.alphas_cumprod[prev_t]) if prev_t >= 0 else self.final_alpha_cumprod
        dx, de = 1.0 / math.sqrt(a), -math.sqrt(1.0 - a) / math.sqrt(a)
        cx = math.sqrt(ap) * dx
        ce = math.sqrt(ap) * de + math.sqrt(1.0 - ap)
        return cx, ce, dx, de

    def step_fused(self, model_output, guidance_scale, sample, timestep, cfg=True, eta=0.0, generator=None):
        if eta != 0.0:
            raise NotImplementedError("DDIM eta != 0 is never used by the reference (src/models.py:43)")
        if self.num_inference_steps is None:
            raise ValueError("run set_timesteps first")
        cx, ce, dx, de = self.coefficients(timestep)
        prev, x0, _ = self._launch(self._prep(model_output), cfg, guidance_scale, self._prep(sample), None, None, None,
                                   (cx, ce, 0, 0, 0, dx, de, 0, 0))
        return prev, x0


@schedulers_registry.add_to_registry("dpm_solver_scheduler")
class DPMSolverScheduler(_FusedStepScheduler):
    """The reference's ``DPMSolverScheduler`` (``src/schedulers.py:12-187``): multistep
    DPM-Solver / DPM-Solver++ orders 1-3 whose ``step`` also returns the x0 prediction.

    Appendix-B quirk #2 of SURVEY.md: the reference unpacks two values from
    ``convert_model_output`` although its ``++`` branch returns one tensor; the intended behaviour
    implemented here is ``model_output := x0_pred`` for ``++`` and ``(epsilon, x0_pred)`` for
    ``dpmsolver``.  The SDE variants (``sde-dpmsolver`` / ``sde-dpmsolver++``, ``src/schedulers.py:134-147``)
    add a Gaussian term through the kernel's noise coefficient: drawn from ``generator`` per step, or taken
    from ``variance_noise``.  Thresholding is off the hot path.
    """
    _own_defaults = dict(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                         trained_betas=None, solver_order=2, prediction_type="epsilon", thresholding=False,
                         algorithm_type="dpmsolver++", solver_type="midpoint", lower_order_final=True,
                         euler_at_final=False, final_sigmas_type="zero", timestep_spacing="linspace", steps_offset=0)
    _accepted = tuple(_own_defaults)

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        c = self.config
        if c.algorithm_type not in ("dpmsolver", "dpmsolver++", "sde-dpmsolver", "sde-dpmsolver++"):
            raise NotImplementedError(f"{c.algorithm_type} is not implemented for {self.__class__}")
        if c.algorithm_type not in ("dpmsolver++", "sde-dpmsolver++") and c.final_sigmas_type == "zero":
            raise ValueError(f"`final_sigmas_type` {c.final_sigmas_type} is not supported for `algorithm_type` "
                             f"{c.algorithm_type}. Please choose `sigma_min` instead.")
        if c.solver_type != "midpoint" or c.prediction_type != "epsilon" or c.thresholding:
            raise NotImplementedError("only midpoint / epsilon / no thresholding is on the hot path")
        if c.solver_order not in (1, 2, 3):
            raise ValueError("solver_order must be 1, 2 or 3")
        self.model_outputs: List[Optional[torch.Tensor]] = [None] * c.solver_order
        self.lower_order_nums = 0
        self.sigmas: Optional[np.ndarray] = None

    def set_timesteps(self, num_inference_steps: Optional[int] = None, device=None, timesteps=None):
        """``timesteps``: custom schedule, as diffusers 0.32.1 accepts it and the two-scheduler pipeline uses it
        (``src/models.py:488-492``)."""
        c = self.config
        T = c.num_train_timesteps
        last = T
        if num_inference_steps is None and timesteps is None:
            raise ValueError("Must pass exactly one of `num_inference_steps` or `timesteps`.")
        if timesteps is not None:
            ts = np.array([int(t) for t in timesteps]).astype(np.int64)
        elif c.timestep_spacing == "linspace":
            ts = np.linspace(0, last - 1, num_inference_steps + 1).round()[::-1][:-1].copy().astype(np.int64)
        elif c.timestep_spacing == "leading":
            ratio = last // (num_inference_steps + 1)
            ts = (np.arange(0, num_inference_steps + 1) * ratio).round()[::-1][:-1].copy().astype(np.int64) + c.steps_offset
        elif c.timestep_spacing == "trailing":
            ts = np.arange(last, 0, -T / num_inference_steps).round().copy().astype(np.int64) - 1
        else:
            raise ValueError(c.timestep_spacing)
        ac = self.alphas_cumprod
        sig = np.array(((1 - ac) / ac) ** 0.5)
        sig = np.interp(ts, np.arange(0, len(sig)), sig)
        if c.final_sigmas_type == "sigma_min":
            sig_last = ((1 - ac[0]) / ac[0]) ** 0.5
        elif c.final_sigmas_type == "zero":
            sig_last = 0
        else:
            raise ValueError(c.final_sigmas_type)
        self.sigmas = np.concatenate([sig, [sig_last]]).astype(np.float32)
        self._set(ts, device)
        self.model_outputs = [None] * c.solver_order
        self.lower_order_nums = 0

    @staticmethod
    def _alpha_sigma_lambda(sigma: float):
        alpha = 1.0 / math.sqrt(sigma * sigma + 1.0)
        sh = sigma * alpha
        lam = math.log(alpha) - math.log(sh) if sh > 0 else math.inf
        return alpha, sh, lam

    def _convert_coefs(self, i: int):
        a0, s0, _ = self._alpha_sigma_lambda(float(self.sigmas[i]))
        yx, ye = 1.0 / a0, -s0 / a0                       # x0 = (x - sigma_t eps) / alpha_t
        if self.config.algorithm_type in ("dpmsolver++", "sde-dpmsolver++"):        # src/schedulers.py:35
            return yx, ye, yx, ye                         # history entry m := x0
        return yx, ye, 0.0, 1.0                           # history entry m := eps  (:64)

    def convert_model_output(self, model_output, *args, sample=None, **kwargs):
        """``src/schedulers.py:14-96``: returns (converted_output, x0_pred) at the current step."""
        if sample is None:
            if len(args) > 1:
                sample = args[1]
            else:
                raise ValueError("missing `sample` as a required keyward argument")
        i = self._step_index if self._step_index is not None else 0
        yx, ye, mx, me = self._convert_coefs(i)
        x, e = self._prep(sample), self._prep(model_output)
        _, x0, m = self._launch(e, False, 0.0, x, None, None, None, (1, 0, 0, 0, 0, yx, ye, mx, me),
                                want_y2=True, want_m=True)
        return m, x0

    def _update_coefs(self, i: int, order: int, mx: float, me: float):
        """prev = px*x + pe*eps + p1*m1 + p2*m2 + pn*z for the order-`order` multistep update (A.6.2); the SDE
        variants (z ~ N(0, I)) use the same linear form with their own kA/kB/kC and a noise coefficient."""
        alg = self.config.algorithm_type
        pp = alg in ("dpmsolver++", "sde-dpmsolver++")
        sde = alg.startswith("sde-")
        a_t, s_t, l_t = self._alpha_sigma_lambda(float(self.sigmas[i + 1]))
        a_0, s_0, l_0 = self._alpha_sigma_lambda(float(self.sigmas[i]))
        h = l_t - l_0
        pn = 0.0
        if pp and not sde:
            ratio = s_t / s_0
            em1 = math.expm1(-h) if math.isfinite(h) else -1.0       # e^{-h} - 1
            kA = -a_t * em1
        elif not sde:
            ratio = a_t / a_0
            em1 = math.expm1(h)
            kA = -s_t * em1
        elif pp:                                                     # sde-dpmsolver++
            e2 = math.expm1(-2.0 * h) if math.isfinite(h) else -1.0   # e^{-2h} - 1
            ratio = s_t / s_0 * (math.exp(-h) if math.isfinite(h) else 0.0)
            kA = -a_t * e2
            pn = s_t * math.sqrt(-e2)
        else:                                                        # sde-dpmsolver
            ratio = a_t / a_0
            kA = -2.0 * s_t * math.expm1(h)
            pn = s_t * math.sqrt(math.expm1(2.0 * h))
        c0, c1, c2 = kA, 0.0, 0.0
        if order >= 2:
            _, _, l_1 = self._alpha_sigma_lambda(float(self.sigmas[i - 1]))
            r0 = (l_0 - l_1) / h
        if order == 2:
            c0 = kA * (1.0 + 0.5 / r0)
            c1 = -0.5 * kA / r0
        elif order == 3:
            _, _, l_2 = self._alpha_sigma_lambda(float(self.sigmas[i - 2]))
            r1 = (l_1 - l_2) / h
            if sde and not pp:
                raise NotImplementedError("sde-dpmsolver has no third-order update (as upstream)")
            if sde:
                kB = a_t * (e2 / (2.0 * h) + 1.0)
                kC = a_t * ((-e2 - 2.0 * h) / (4.0 * h * h) - 0.5)
            elif pp:
                kB = a_t * (em1 / h + 1.0)
                kC = -a_t * ((em1 + h) / (h * h) - 0.5)
            else:
                kB = -s_t * (em1 / h - 1.0)
                kC = -s_t * ((em1 - h) / (h * h) - 0.5)
            rho = r0 / (r0 + r1)
            u = kB * (1.0 + rho) + kC / (r0 + r1)
            v = -kB * rho - kC / (r0 + r1)
            c0 = kA + u / r0
            c1 = -u / r0 + v / r1
            c2 = -v / r1
        return ratio + c0 * mx, c0 * me, c1, c2, pn

    def step_fused(self, model_output, guidance_scale, sample, timestep, cfg=True, eta=0.0, generator=None,
                   variance_noise: Optional[torch.Tensor] = None):
        c = self.config
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        if self._step_index is None:
            self._step_index = self._index_of(timestep)
        i, n = self._step_index, len(self._timesteps_list)
        lower_order_final = (i == n - 1) and (c.euler_at_final or (c.lower_order_final and n < 15)
                                              or c.final_sigmas_type == "zero")
        lower_order_second = (i == n - 2) and c.lower_order_final and n < 15
        if c.solver_order == 1 or self.lower_order_nums < 1 or lower_order_final:
            order = 1
        elif c.solver_order == 2 or self.lower_order_nums < 2 or lower_order_second:
            order = 2
        else:
            order = 3
        yx, ye, mx, me = self._convert_coefs(i)
        px, pe, p1, p2, pn = self._update_coefs(i, order, mx, me)
        hist = self.model_outputs                      # oldest ... newest
        m1 = hist[-1] if order >= 2 else None
        m2 = hist[-2] if order >= 3 else None
        x = self._prep(sample)
        z = None
        if c.algorithm_type.startswith("sde-"):        # src/schedulers.py:134-147
            if variance_noise is None:
                variance_noise = sdist.randn(x.shape, generator)
            z = self._prep(variance_noise.to(x.device))
        prev, x0, m0 = self._launch(self._prep(model_output), cfg, guidance_scale, x, m1, m2, z,
                                    (px, pe, p1, p2, pn, yx, ye, mx, me), want_y2=True, want_m=True)
        for k in range(c.solver_order - 1):
            self.model_outputs[k] = self.model_outputs[k + 1]
        self.model_outputs[-1] = m0
        if self.lower_order_nums < c.solver_order:
            self.lower_order_nums += 1
        self._step_index += 1
        return prev, x0


@schedulers_registry.add_to_registry("lcm_scheduler")
class LCMScheduler(_FusedStepScheduler):
    """diffusers ``LCMScheduler`` (A.6.3): consistency step with boundary-condition scalings and
    re-noising by a fresh Gaussian on every step but the last."""
    _own_defaults = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                         trained_betas=None, original_inference_steps=50, clip_sample=False, set_alpha_to_one=True,
                         steps_offset=0, prediction_type="epsilon", thresholding=False, timestep_spacing="leading",
                         timestep_scaling=10.0)
    _accepted = tuple(_own_defaults)

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        if self.config.prediction_type != "epsilon" or self.config.clip_sample or self.config.thresholding:
            raise NotImplementedError("only epsilon prediction without clipping is on the hot path")
        self.final_alpha_cumprod = 1.0 if self.config.set_alpha_to_one else float(self.alphas_cumprod[0])

    def set_timesteps(self, num_inference_steps: int, device=None, original_inference_steps=None):
        c = self.config
        original = original_inference_steps or c.original_inference_steps
        if num_inference_steps > original:
            raise ValueError("num_inference_steps cannot exceed original_inference_steps")
        k = c.num_train_timesteps // original
        origin = (np.asarray(list(range(1, int(original) + 1))) * k - 1)[::-1].copy()
        idx = np.floor(np.linspace(0, len(origin), num=num_inference_steps, endpoint=False)).astype(np.int64)
        self._set(origin[idx], device)

    def step_fused(self, model_output, guidance_scale, sample, timestep, cfg=True, eta=0.0, generator=None,
                   noise: Optional[torch.Tensor] = None):
        if self.num_inference_steps is None:
            raise ValueError("run set_timesteps first")
        if self._step_index is None:
            self._step_index = self._index_of(timestep)
        i = self._step_index
        t = int(timestep)
        prev_t = self._timesteps_list[i + 1] if i + 1 < len(self._timesteps_list) else t
        a = float(self.alphas_cumprod[t])
        ap = float(self.alphas_cumprod[prev_t]) if prev_t >= 0 else self.final_alpha_cumprod
        st = t * self.config.timestep_scaling
        c_skip = 0.25 / (st * st + 0.25)
        c_out = st / math.sqrt(st * st + 0.25)
        dx, de = 1.0 / math.sqrt(a), -math.sqrt(1.0 - a) / math.sqrt(a)
        yx, ye = c_out * dx + c_skip, c_out * de            # "denoised"
        x = self._prep(sample)
        last = i == self.num_inference_steps - 1
        if last:
            coef, z = (yx, ye, 0, 0, 0, yx, ye, 0, 0), None
        else:
            if noise is None:
                noise = sdist.randn(x.shape, generator)
            z = self._prep(noise.to(x.device))
            coef = (math.sqrt(ap) * yx, math.sqrt(ap) * ye, 0, 0, math.sqrt(1.0 - ap), yx, ye, 0, 0)
        prev, den, _ = self._launch(self._prep(model_output), cfg, guidance_scale, x, None, None, z, coef)
        self._step_index += 1
        return prev, den


@schedulers_registry.add_to_registry("pndm_scheduler")
class PNDMScheduler(_FusedStepScheduler):
    """diffusers ``PNDMScheduler`` with ``skip_prk_steps=True`` (PLMS) -- the scheduler the SD-1.5
    checkpoint ships and that the reference's ``deep_cache`` / ``default`` methods therefore run
    (``src/experiments/deep_cache.py:17-18``, ``default_sd.py:15-16``).
    N inference steps = N+1 UNet calls (second timestep duplicated).  Every PLMS update is a linear
    combination of the sample, the current and up to three earlier noise predictions, so it runs on
    the same fused kernel; ``step`` returns a 1-tuple like upstream."""
    _own_defaults = dict(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                         trained_betas=None, skip_prk_steps=False, set_alpha_to_one=False, prediction_type="epsilon",
                         timestep_spacing="leading", steps_offset=0)
    _accepted = tuple(_own_defaults)

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        if not self.config.skip_prk_steps or self.config.prediction_type != "epsilon" or \
                self.config.timestep_spacing != "leading":
            raise NotImplementedError("only the SD-1.5 PLMS configuration (skip_prk_steps, epsilon, leading) is built")
        self.final_alpha_cumprod = 1.0 if self.config.set_alpha_to_one else float(self.alphas_cumprod[0])
        self.ets: List[torch.Tensor] = []
        self.counter = 0
        self.cur_sample = None

    def set_timesteps(self, num_inference_steps: int, device=None):
        T = self.config.num_train_timesteps
        ratio = T // num_inference_steps
        base = (np.arange(0, num_inference_steps) * ratio).round() + self.config.steps_offset
        plms = np.concatenate([base[:-1], base[-2:-1], base[-1:]])[::-1].copy().astype(np.int64)
        self._set(plms, device)
        self.num_inference_steps = num_inference_steps        # N, not len(timesteps) = N + 1
        self.ets, self.counter, self.cur_sample = [], 0, None

    def _prev_coefs(self, t: int, prev_t: int):
        a_t = float(self.alphas_cumprod[t])
        a_p = float(self.alphas_cumprod[prev_t]) if prev_t >= 0 else self.final_alpha_cumprod
        sample_coeff = math.sqrt(a_p / a_t)
        denom = a_t * math.sqrt(1 - a_p) + math.sqrt(a_t * (1 - a_t) * a_p)
        return sample_coeff, -(a_p - a_t) / denom

    def step_fused(self, model_output, guidance_scale, sample, timestep, cfg=True, eta=0.0, generator=None):
        if self.num_inference_steps is None:
            raise ValueError("run set_timesteps first")
        t = int(timestep)
        ratio = self.config.num_train_timesteps // self.num_inference_steps
        prev_t = t - ratio
        x = self._prep(sample)
        e = self._prep(model_output)
        hist = self.ets
        if self.counter != 1:
            n_hist = min(len(hist), 3)
            w = {0: (1.0,), 1: (1.5, -0.5), 2: (23 / 12, -16 / 12, 5 / 12), 3: (55 / 24, -59 / 24, 37 / 24, -9 / 24)}[n_hist]
            sc, k = self._prev_coefs(t, prev_t)
            ms = [hist[-1 - i] if i < n_hist else None for i in range(3)]
            ws = [w[i + 1] if i < n_hist else 0.0 for i in range(3)]
            prev, _, e_out = self._launch(e, cfg, guidance_scale, x, ms[0], ms[1], None,
                                          (sc, k * w[0], k * ws[0], k * ws[1], 0, 0, 0, 0.0, 1.0, k * ws[2]),
                                          want_y2=False, want_m=True, m3=ms[2])
            self.ets = (hist + [e_out])[-4:]
            if self.counter == 0:
                self.cur_sample = x
        else:
            # second call (same timestep again): average with the first prediction, restart from cur_sample
            sc, k = self._prev_coefs(t + ratio, t)
            prev, _, _ = self._launch(e, cfg, guidance_scale, self.cur_sample, hist[-1], None, None,
                                      (sc, 0.5 * k, 0.5 * k, 0, 0, 0, 0, 0, 0, 0), want_y2=False, want_m=False)
            self.cur_sample = None
        self.counter += 1
        return (prev,)

    def step(self, model_output, timestep, sample, return_dict: bool = False, **kwargs):
        out_dtype = model_output.dtype
        (prev,) = self.step_fused(model_output, 0.0, sample, timestep, cfg=False)
        return (prev.to(out_dtype),)


class PNDMConfigStub:
    """The checkpoint's scheduler config holder: ``from_config(model.scheduler.config)``
    (``src/experiments/base_experiment.py:69-72``) reads the SD-1.5 PNDM config from it.  Call
    ``PNDMScheduler.from_config(stub.config)`` to actually step with PNDM."""
    order = 1
    init_noise_sigma = 1.0

    def __init__(self):
        self.config = SchedulerConfig(SD15_SCHEDULER_CONFIG)

    def set_timesteps(self, *a, **k):
        raise NotImplementedError("this is only the checkpoint's scheduler CONFIG; build a scheduler with "
                                  "schedulers_registry[...].from_config(model.scheduler.config)")
