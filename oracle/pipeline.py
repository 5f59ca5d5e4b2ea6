"""fp32 CPU restatement of the reference's denoising loop.

TEST INFRASTRUCTURE (see oracle/__init__.py) -- parity unpinned.

Follows ``StableDiffusionModel.call`` (``src/models.py:32-335``): CFG concat ``:154-155``,
timesteps ``:167-169``, loop ``:210-282`` (cat x2 ``:217``, scale_model_input ``:222``, UNet
``:227``, CFG combine ``:238-242``, scheduler.step ``:253-261``), loop-only timing
``:208,284-285``.  Text encoding and VAE decode sit outside the hot path so the
oracle starts from prompt embeddings and ends at latents (``output_type="latent"``).
"""
from __future__ import annotations

import time
from typing import List, Optional

import torch

from .unet import DeepCacheState, UNetConfig, unet_forward


@torch.no_grad()
def sample_loop(weights, cfg: UNetConfig, scheduler, prompt_embeds: torch.Tensor,
                negative_prompt_embeds: Optional[torch.Tensor], latents: torch.Tensor,
                num_inference_steps: int, guidance_scale: float = 7.5,
                deepcache: Optional[DeepCacheState] = None,
                lcm_noise: Optional[torch.Tensor] = None,
                max_steps: Optional[int] = None,
                forced_eps: Optional[List[torch.Tensor]] = None, fq=None):
    """Returns (final_latents, execution_time_s, x0_preds, trajectory).

    ``lcm_noise`` [N-1,B,4,H,W]: pre-drawn re-noising tensors for LCM so the
    trajectory does not depend on a device RNG.  ``max_steps`` truncates the loop (used by the
    bounded cpu_baseline sample).  ``trajectory`` holds the latents after each step and the
    CFG-combined noise prediction of each step, for teacher-forced comparisons.
    """
    do_cfg = guidance_scale > 1.0
    ctx = torch.cat([negative_prompt_embeds, prompt_embeds]) if do_cfg else prompt_embeds
    scheduler.set_timesteps(num_inference_steps)
    timesteps = scheduler.timesteps
    latents = latents.float() * scheduler.init_noise_sigma
    if deepcache is not None:
        deepcache.cached.clear()
        deepcache.start_timestep = None
    x0_preds = []
    traj = {"latents": [], "noise_pred": []}
    start = time.time()
    for i, t in enumerate(timesteps):
        if max_steps is not None and i >= max_steps:
            break
        latent_in = torch.cat([latents] * 2) if do_cfg else latents
        latent_in = scheduler.scale_model_input(latent_in, t)
        if deepcache is not None:
            # DeepCache's wrapped unet.forward: index of t in scheduler.timesteps (A.5)
            deepcache.cur_timestep = list(int(x) for x in timesteps).index(int(t))
        noise_pred = unet_forward(weights, cfg, latent_in, t, ctx, dc=deepcache, fq=fq)
        if do_cfg:
            u, c = noise_pred.chunk(2)
            noise_pred = u + guidance_scale * (c - u)
        traj["noise_pred"].append(noise_pred)
        kwargs = {}
        if lcm_noise is not None and i < len(timesteps) - 1:
            kwargs["noise"] = lcm_noise[i]
        step = scheduler.step(noise_pred, t, latents, return_dict=False, **kwargs)
        if len(step) == 1:
            latents = step[0]
        else:
            latents, x0 = step[0], step[1]
            x0_preds.append(x0[0].unsqueeze(0))
        traj["latents"].append(latents)
    exec_time = time.time() - start
    return latents, exec_time, x0_preds, traj


# ------------------------------------------------------------------------------------------------
# Variant pipelines: host-only control flow over the same UNet and schedulers.
# ------------------------------------------------------------------------------------------------
def switch_timestamp(timesteps_first, timesteps_second, num_step_switch: int, type_switch: str = "closest"):
    """``StableDiffusionModelTwoSchedulers.switch_timestamp`` (``src/models.py:704-730``): the first
    ``num_step_switch`` timesteps of the first schedule, then the second schedule from the entry closest to
    (``closest``), last not below (``left_closest``) or first not above (``right_closest``) the switch point."""
    first = [int(t) for t in timesteps_first][:num_step_switch]
    second = [int(t) for t in timesteps_second]
    pivot = first[-1]
    if type_switch == "closest":
        dist = [abs(t - pivot) for t in second]
        second = second[dist.index(min(dist)):]
    elif type_switch == "left_closest":
        idx = [i for i, t in enumerate(second) if t - pivot >= 0]
        second = second[idx[-1]:]
    elif type_switch == "right_closest":
        idx = [i for i, t in enumerate(second) if t - pivot <= 0]
        second = second[idx[0]:]
    return first, second


def _is_dpm(s) -> bool:
    return hasattr(s, "model_outputs") and hasattr(s, "convert_model_output")


def _push_history(sched, noise_pred, latents):
    """History hand-off of ``src/models.py:603-611`` / ``:1025-1033``: the other scheduler's
    ``model_outputs`` are shifted and its converted output of this noise prediction appended
    (``sample=latents`` is the latents AFTER the step, as written in the reference)."""
    out = sched.convert_model_output(noise_pred, sample=latents)
    model_output = out[0] if isinstance(out, tuple) else out
    for k in range(sched.config["solver_order"] - 1):
        sched.model_outputs[k] = sched.model_outputs[k + 1]
    sched.model_outputs[-1] = model_output


def _eps(weights, cfg, latents, t, ctx, do_cfg, guidance_scale, scheduler):
    latent_in = torch.cat([latents] * 2) if do_cfg else latents
    latent_in = scheduler.scale_model_input(latent_in, t)
    noise_pred = unet_forward(weights, cfg, latent_in, t, ctx)
    if do_cfg:
        u, c = noise_pred.chunk(2)
        noise_pred = u + guidance_scale * (c - u)
    return noise_pred


def _take(step, x0_preds):
    if len(step) == 2:
        x0_preds.append(step[1][0].unsqueeze(0))
    return step[0]


@torch.no_grad()
def sample_loop_two_schedulers(weights, cfg, scheduler_first, scheduler_second, prompt_embeds, negative_prompt_embeds,
                               latents, num_inference_steps_first: int, num_step_switch: int,
                               type_switch: str = "closest", guidance_scale: float = 7.5):
    """``StableDiffusionModelTwoSchedulers.call`` (``src/models.py:349-702``).  The second scheduler is
    given the FIRST scheduler's timesteps as a custom schedule (``:488-492``)."""
    do_cfg = guidance_scale > 1.0
    ctx = torch.cat([negative_prompt_embeds, prompt_embeds]) if do_cfg else prompt_embeds
    scheduler_first.set_timesteps(num_inference_steps_first)
    scheduler_second.set_timesteps(timesteps=[int(t) for t in scheduler_first.timesteps])
    first, second = switch_timestamp(scheduler_first.timesteps, scheduler_second.timesteps, num_step_switch, type_switch)
    latents = latents.float() * scheduler_first.init_noise_sigma
    x0_preds = []
    for i, t in enumerate(first + second):
        sched = scheduler_first if i < len(first) else scheduler_second
        noise_pred = _eps(weights, cfg, latents, t, ctx, do_cfg, guidance_scale, sched)
        latents = _take(sched.step(noise_pred, t, latents, return_dict=False), x0_preds)
        if i < len(first) and _is_dpm(scheduler_second):
            _push_history(scheduler_second, noise_pred, latents)
    return latents, x0_preds, first + second


def interleave_plan(timesteps_main, solver_order: int, interliving_steps):
    """``src/models.py:952-966``: main-schedule groups of ``solver_order`` steps listed in
    ``interliving_steps`` are replaced by ONE step of the inter scheduler at the group's first timestep."""
    ts = [int(t) for t in timesteps_main]
    keep, t_inter = [], []
    for i, t in enumerate(ts):
        if i // solver_order in interliving_steps:
            if i % solver_order != 0:
                continue
            t_inter.append(t)
        keep.append(t)
    return keep, t_inter


@torch.no_grad()
def sample_loop_interleaving(weights, cfg, scheduler_main, scheduler_inter, prompt_embeds, negative_prompt_embeds,
                             latents, num_inference_steps: int, interliving_steps, guidance_scale: float = 7.5):
    """``StableDiffusionModelInterlivingSchedulers.call`` (``src/models.py:744-1136``)."""
    do_cfg = guidance_scale > 1.0
    ctx = torch.cat([negative_prompt_embeds, prompt_embeds]) if do_cfg else prompt_embeds
    order = scheduler_main.config["solver_order"]
    scheduler_main.set_timesteps(num_inference_steps)
    scheduler_inter.set_timesteps(num_inference_steps // order)
    keep, t_inter = interleave_plan(scheduler_main.timesteps, order, list(interliving_steps))
    latents = latents.float() * scheduler_main.init_noise_sigma
    x0_preds = []
    for t in keep:
        if t in t_inter:
            noise_pred = _eps(weights, cfg, latents, t, ctx, do_cfg, guidance_scale, scheduler_inter)
            latents = _take(scheduler_inter.step(noise_pred, t, latents, return_dict=False), x0_preds)
            _push_history(scheduler_main, noise_pred, latents)
        else:
            noise_pred = _eps(weights, cfg, latents, t, ctx, do_cfg, guidance_scale, scheduler_main)
            latents = _take(scheduler_main.step(noise_pred, t, latents, return_dict=False), x0_preds)
            if _is_dpm(scheduler_inter):
                _push_history(scheduler_inter, noise_pred, latents)
    return latents, x0_preds, keep


@torch.no_grad()
def sample_loop_skip(weights, cfg, scheduler, prompt_embeds, negative_prompt_embeds, latents,
                     num_inference_steps: int, skip_timesteps, guidance_scale: float = 7.5):
    """``StableDiffusionModelSkipTimesteps.call`` (``src/models.py:1149-1467``): loop indices listed in
    ``skip_timesteps`` are skipped outright (``:1327-1330``); a multistep scheduler's internal step index
    is NOT advanced for them, exactly as in the reference."""
    do_cfg = guidance_scale > 1.0
    ctx = torch.cat([negative_prompt_embeds, prompt_embeds]) if do_cfg else prompt_embeds
    scheduler.set_timesteps(num_inference_steps)
    latents = latents.float() * scheduler.init_noise_sigma
    x0_preds, used = [], []
    for i, t in enumerate(scheduler.timesteps):
        if i in skip_timesteps:
            continue
        noise_pred = _eps(weights, cfg, latents, t, ctx, do_cfg, guidance_scale, scheduler)
        latents = _take(scheduler.step(noise_pred, t, latents, return_dict=False), x0_preds)
        used.append(int(t))
    return latents, x0_preds, used
