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
                forced_eps: Optional[List[torch.Tensor]] = None):
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
        noise_pred = unet_forward(weights, cfg, latent_in, t, ctx, dc=deepcache)
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
