"""``StableDiffusionModel`` -- the reference's pipeline plugin over the MI355X-native kernels.

Registered as ``models_registry["stable_diffusion_model"]`` like ``src/models.py:21`` of the
reference and callable the same way (``src/models.py:23-29,32-335``):

    images, execution_time, x0_preds = model(prompts, num_inference_steps=N, guidance_scale=7.5,
                                             generator=g, output_type="pt")

The denoising loop (``src/models.py:210-282``) is restated with three changes that are the
point of this framework: the UNet forward is libsdhip (hand-written gfx950 kernels), the CFG
duplication is fused into conv_in, and CFG-combine + ``scheduler.step`` is one fused launch.
Timing keeps the reference's definition -- wall-clock of the loop only (``:208,284-285``) -- but
brackets it with device synchronisation so the number is real.

``output_type="pt"`` decodes through the libsdhip AutoencoderKL decoder (``vae.py``, SURVEY 8f row 1;
outside the timed loop, as in the reference).  The three variant pipelines of ``src/models.py:338-1467``
(two schedulers, interleaved schedulers, skipped timesteps) are host-only control flow over
the same kernels and live at the bottom of this file.  Prompts are encoded by a pluggable ``text_encoder``:
the CLIP text tower on libsdhip + BPE tokenizer (``clip.py``) when the checkpoint directory
carries ``tokenizer/`` and ``text_encoder/``, otherwise a deterministic seeded stand-in (no CLIP weights
exist offline) -- ``weights_source`` / the reports say which.
"""
from __future__ import annotations

import hashlib
import os
import time
from dataclasses import dataclass
from typing import Callable, List, Optional, Union

import torch

from . import dist as sdist
from .registry import models_registry
from .schedulers import PNDMConfigStub
from .unet import CACHE_FULL_AND_STORE, CACHE_OFF, CACHE_SKIP, HipUNet2DConditionModel
from .vae import HipVaeDecoder, VaeConfig, load_vae_state_dict, make_synthetic_vae_state_dict
from .weights import UNetConfig, load_unet_config, load_unet_state_dict, make_synthetic_state_dict


@dataclass
class StableDiffusionPipelineOutput:
    images: torch.Tensor
    nsfw_content_detected: Optional[list] = None


class SyntheticTextEncoder:
    """Deterministic stand-in for CLIP ViT-L/14 text encoding: a prompt's [77,768] embedding is
    drawn ~N(0,1) from a generator seeded by the prompt's SHA-256.  Same prompt -> same
    embedding on every rank and box; documented as synthetic in every report."""

    def __init__(self, context_len: int = 77, dim: int = 768):
        self.context_len, self.dim = context_len, dim
        self._cache = {}

    def __call__(self, prompts: List[str]) -> torch.Tensor:
        out = []
        for p in prompts:
            if p not in self._cache:
                seed = int.from_bytes(hashlib.sha256(p.encode("utf-8")).digest()[:8], "little") & (2 ** 63 - 1)
                g = torch.Generator().manual_seed(seed)
                self._cache[p] = torch.randn((self.context_len, self.dim), generator=g)
            out.append(self._cache[p])
        return torch.stack(out)


class _VaeConfig:
    scaling_factor = 0.18215


def postprocess_images(image01: torch.Tensor, output_type: str):
    """The tail of ``src/models.py:312-321`` (diffusers ``VaeImageProcessor.postprocess`` on denormalised images):
    ``image01`` is ``[B, 3, H, W]`` in [0, 1].  "pt": as is; "np": float32 ``[B, H, W, 3]`` on the host; "pil": a list of
    ``PIL.Image`` (uint8, ``round(255 x)``)."""
    if output_type == "pt":
        return image01
    arr = image01.detach().to("cpu", torch.float32).permute(0, 2, 3, 1).contiguous().numpy()
    if output_type == "np":
        return arr
    if output_type == "pil":
        from PIL import Image
        return [Image.fromarray(a) for a in (arr * 255.0).round().astype("uint8")]
    raise NotImplementedError(f"output_type {output_type!r}")


@models_registry.add_to_registry("stable_diffusion_model")
class StableDiffusionModel:
    vae_scale_factor = 8

    def __init__(self, unet_config: Optional[UNetConfig] = None, state_dict=None, scheduler=None,
                 text_encoder: Optional[Callable] = None, vae_decoder: Optional[Callable] = None,
                 weights_seed: int = 1234, source: str = "synthetic", clip_dir: Optional[str] = None,
                 weight_dtype: Optional[str] = None):
        self.unet_config = unet_config or UNetConfig()
        # "bf16" | "fp8": operand type of the UNet's MFMA contractions (YAML model.weight_dtype, or SD_AMD_WEIGHT_DTYPE)
        self.weight_dtype = weight_dtype or os.environ.get("SD_AMD_WEIGHT_DTYPE", "bf16")
        self._state_dict = state_dict
        self._weights_seed = weights_seed
        self.weights_source = source
        self.unet: Optional[HipUNet2DConditionModel] = None
        self.scheduler = scheduler or PNDMConfigStub()
        self.text_encoder = text_encoder or SyntheticTextEncoder(self.unet_config.context_len,
                                                                 self.unet_config.cross_attention_dim)
        # a local checkpoint with tokenizer/ + text_encoder/: the CLIP text tower on libsdhip replaces the stand-in
        self._clip_dir = clip_dir if text_encoder is None else None
        self.vae_decoder = vae_decoder
        self.vae_config = _VaeConfig()
        self.device = torch.device("cpu")
        self._num_timesteps = 0
        self._guidance_scale = 7.5
        self._deepcache = None          # set by DeepCacheSDHelper.enable()
        self._lora = []
        self._fp8_calibrated = False    # fp8 handles: per-tensor activation scales, calibrated once on a FIXED seeded batch
        self.fp8_scales = {}            # {tensor name: scale} in use (reported by the harness next to the results)

    # -- loading ---------------------------------------------------------------------------
    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, timestamps=None, safety_checker=None,
                        requires_safety_checker=False, torch_dtype=None, **kwargs):
        """``from_pretrained`` of the harness (``src/experiments/base_experiment.py:57-63``).

        A local diffusers directory is loaded; a model NAME is a network fetch and cannot be
        resolved offline, in which case SD-1.5-shaped synthetic weights seeded
        by ``SD_AMD_WEIGHTS_SEED`` (default 1234) are used and ``weights_source`` says so."""
        path = os.environ.get("SD_AMD_MODEL_DIR") or str(pretrained_model_name_or_path)
        if os.path.isdir(path):
            has_clip = all(os.path.exists(os.path.join(path, *p)) for p in (("tokenizer", "vocab.json"),
                                                                            ("tokenizer", "merges.txt"),
                                                                            ("text_encoder", "model.safetensors")))
            return cls(unet_config=load_unet_config(path), state_dict=load_unet_state_dict(path), source=f"local:{path}",
                       clip_dir=path if has_clip else None, **kwargs)
        seed = int(os.environ.get("SD_AMD_WEIGHTS_SEED", "1234"))
        return cls(weights_seed=seed, source=f"synthetic(seed={seed}) for {pretrained_model_name_or_path}", **kwargs)

    def _ensure_unet(self):
        if self.unet is None:
            sd = self._state_dict or make_synthetic_state_dict(self.unet_config, self._weights_seed)
            for item in self._lora:
                if item[0] == "file":
                    from .weights import fuse_lora_state_dict
                    fuse_lora_state_dict(sd, item[1], item[2])
                    self.weights_source += " + LoRA(local file) fused"
                else:
                    _fuse_synthetic_lora(sd, *item)
                    self.weights_source += f" + SYNTHETIC low-rank stand-in for a hub LoRA (seed={item[0]}) fused"
            self.unet = HipUNet2DConditionModel(self.unet_config, sd, device="cuda:%d" % torch.cuda.current_device(),
                                                weight_dtype=self.weight_dtype)
            if self.unet.weight_dtype != "bf16":
                self.weights_source += f" [{self.unet.weight_dtype} weights + activations in the conv / FF / QKV contractions]"
            self._state_dict = None

    def _ensure_vae(self):
        """AutoencoderKL decoder on libsdhip; local weights if the model directory has
        them, SD-1.5-shaped synthetic weights otherwise."""
        if self.vae_decoder is None:
            self._ensure_unet()
            cfg = VaeConfig(sample_size=self.unet_config.sample_size)
            path = os.environ.get("SD_AMD_MODEL_DIR", "")
            try:
                sd = load_vae_state_dict(path) if path and os.path.isdir(path) else None
            except FileNotFoundError:
                sd = None
            sd = sd or make_synthetic_vae_state_dict(cfg)
            self.vae_decoder = HipVaeDecoder(cfg, sd, device=str(self.unet.device))
        return self.vae_decoder

    def to(self, device):
        """``model.to(device)`` (``base_experiment.py:64``; ``ddim.py:31,33``).  The UNet weights
        live in HBM for the life of the object (1.7 GB of 288 GB); ``to("cpu")`` is a no-op."""
        device = torch.device(device)
        if device.type == "cuda":
            self._ensure_unet()
            self.device = self.unet.device
            if self._clip_dir is not None:
                from .clip import ClipPromptEncoder
                self.text_encoder = ClipPromptEncoder.from_pretrained(self._clip_dir, device=str(self.device))
                self._clip_dir = None
        return self

    # LCM-LoRA hooks used by src/experiments/consistency_model.py:20-21
    def load_lora_weights(self, adapter_id, scale: float = 1.0, rank: int = 64):
        """A local LoRA file / directory (``pytorch_lora_weights.safetensors``) is read and fused for real
        (``weights.fuse_lora_state_dict``); a hub NAME is a network fetch, for which a seeded synthetic
        low-rank update of the same structure stands in."""
        if self.unet is not None:
            raise RuntimeError("load_lora_weights must be called before the model is moved to the GPU")
        path = str(adapter_id)
        if os.path.isdir(path):
            path = os.path.join(path, "pytorch_lora_weights.safetensors")
        if os.path.isfile(path):
            from safetensors.torch import load_file
            self._pending_lora = ("file", load_file(path), scale)
            return
        seed = int.from_bytes(hashlib.sha256(str(adapter_id).encode()).digest()[:4], "little")
        self._pending_lora = (seed, scale, rank)

    def fuse_lora(self):
        if getattr(self, "_pending_lora", None) is not None:
            self._lora.append(self._pending_lora)
            self._pending_lora = None

    # -- properties the harness reads --------------------------------------------------------
    @property
    def num_timesteps(self):
        return self._num_timesteps

    @property
    def guidance_scale(self):
        return self._guidance_scale

    @property
    def do_classifier_free_guidance(self):
        return self._guidance_scale > 1 and self.unet_config.time_cond_proj_dim is None

    # -- helpers of the loop -----------------------------------------------------------------
    def encode_prompt(self, prompt, device, do_cfg, negative_prompt=None, prompt_embeds=None,
                      negative_prompt_embeds=None):
        if prompt_embeds is None:
            prompts = [prompt] if isinstance(prompt, str) else list(prompt)
            prompt_embeds = self.text_encoder(prompts)
        if do_cfg and negative_prompt_embeds is None:
            n = prompt_embeds.shape[0]
            neg = negative_prompt if negative_prompt is not None else [""] * n
            neg = [neg] * n if isinstance(neg, str) else list(neg)
            negative_prompt_embeds = self.text_encoder(neg)
        to = lambda t: None if t is None else t.to(device, torch.float32)
        return to(prompt_embeds), to(negative_prompt_embeds)

    def prepare_latents(self, batch_size, num_channels, height, width, device, generator, latents=None):
        shape = (batch_size, num_channels, height // self.vae_scale_factor, width // self.vae_scale_factor)
        if latents is None:
            latents = sdist.randn(shape, generator)      # (the global batch's draw, sliced, under a sharded harness call)
        latents = latents.to(device, torch.float32)
        return (latents * self.scheduler.init_noise_sigma).contiguous()

    def __call__(self, *args, return_execution_time=True, **kwargs):
        result, execution_time, x0_preds = self.call(*args, **kwargs)
        if return_execution_time:
            return result, execution_time, x0_preds
        return result, x0_preds

    # -- pieces shared by the four pipelines ------------------------------------------------------
    def _begin(self, prompt, height, width, guidance_scale, negative_prompt, num_images_per_prompt, prompt_embeds,
               negative_prompt_embeds, guidance_rescale=0.0, timesteps=None, sigmas=None):
        """Steps 0-3 of the reference's ``call`` (``src/models.py:110-160``): argument checks, batch size,
        prompt encoding, CFG concat; uploads the prompt K/V projections.  Returns (device, batch, do_cfg, ctx)."""
        if guidance_rescale != 0.0:
            raise NotImplementedError("guidance_rescale is never used by the reference (src/models.py:53)")
        if num_images_per_prompt != 1 or timesteps is not None or sigmas is not None:
            raise NotImplementedError("custom timesteps / num_images_per_prompt are outside the reference's use")
        self._ensure_unet()
        if not self._fp8_calibrated:
            self.calibrate_fp8()        # fp8 handles only; shared by the four pipelines (all enter through _begin)
        device = self.unet.device
        cfgu = self.unet_config
        height = height or cfgu.sample_size * self.vae_scale_factor
        width = width or cfgu.sample_size * self.vae_scale_factor
        if height != cfgu.sample_size * 8 or width != cfgu.sample_size * 8:
            raise ValueError("resolution is fixed by the UNet sample_size")
        self._guidance_scale = guidance_scale
        if prompt is not None and isinstance(prompt, str):
            batch_size = 1
        elif prompt is not None:
            batch_size = len(prompt)
        else:
            batch_size = prompt_embeds.shape[0]
        do_cfg = self.do_classifier_free_guidance
        prompt_embeds, negative_prompt_embeds = self.encode_prompt(
            prompt, device, do_cfg, negative_prompt, prompt_embeds, negative_prompt_embeds)
        ctx = torch.cat([negative_prompt_embeds, prompt_embeds]) if do_cfg else prompt_embeds   # :154-155
        return device, batch_size, do_cfg, ctx

    def _eps_buffer(self, unet_batch, device):
        c = self.unet_config
        return torch.empty((unet_batch, c.out_channels, c.sample_size, c.sample_size), dtype=torch.float32, device=device)

    def _finish(self, latents, x0_preds, output_type, return_dict, execution_time):
        """``src/models.py:287-335``: decode (outside the timed loop), post-process, 3-tuple."""
        if output_type == "latent":
            image = latents
            image_x0 = x0_preds
        elif output_type in ("pt", "np", "pil"):
            vae = self._ensure_vae()
            inv = 1.0 / self.vae_config.scaling_factor
            image = postprocess_images((vae.decode(latents, inv) / 2 + 0.5).clamp(0, 1), output_type)   # :288,:312
            # the reference decodes EVERY stored x0 prediction as well (:296-302)
            image_x0 = [postprocess_images((vae.decode(x, inv) / 2 + 0.5).clamp(0, 1), output_type) for x in x0_preds]
        else:
            raise NotImplementedError(f"output_type {output_type!r}: 'latent', 'pt', 'np' and 'pil' are built")
        if not return_dict:
            return (image, None), execution_time, image_x0
        return StableDiffusionPipelineOutput(images=image, nsfw_content_detected=None), execution_time, image_x0

    # fixed calibration inputs: the SAME on every rank, for every prompt order, shard size and schedule
    FP8_CALIBRATION_SEED = 20240229
    FP8_CALIBRATION_PROMPTS = ("", "a photograph of an astronaut riding a horse")
    FP8_CALIBRATION_TIMESTEPS = (999.0, 499.0, 1.0)

    def calibrate_fp8(self, scales=None, margin: float = 2.0):
        """fp8 handles (``weight_dtype="fp8"``): fix the per-tensor e4m3 activation scales, ONCE and deterministically.
        ``scales`` given (e.g. saved from an earlier run: ``model.fp8_scales``): they are loaded through
        ``sd_unet_set_fp8_scale``.  Otherwise they come from the amax observed on a fixed calibration batch -- two latents
        drawn from a generator seeded with ``FP8_CALIBRATION_SEED``, the text encoder's embeddings of two fixed prompts, no
        CFG, at three fixed timesteps (``sd_unet_calibrate_fp8``, margin 2) -- so every rank of a sharded run, whatever its
        shard, and every world size end up with the SAME scales (dist.py's invariant: an image does not depend on the
        world size; round 3 calibrated on the first call's own inputs, which differ per rank).  Runs outside any timed
        region; ``SD_AMD_FP8_CALIBRATE=0`` keeps the static defaults (clipping at |x| > 56 / 224)."""
        self._ensure_unet()
        if self.unet.weight_dtype != "fp8_e4m3":
            self._fp8_calibrated = True
            return {}
        if scales is not None:
            self.unet.set_fp8_scales(scales)
            how = "loaded"
        elif os.environ.get("SD_AMD_FP8_CALIBRATE", "1") == "0":
            self._fp8_calibrated = True
            return {}
        else:
            cfgu = self.unet_config
            g = torch.Generator().manual_seed(self.FP8_CALIBRATION_SEED)
            lat = torch.randn((2, cfgu.in_channels, cfgu.sample_size, cfgu.sample_size), generator=g)
            ctx = self.text_encoder(list(self.FP8_CALIBRATION_PROMPTS)).to(self.unet.device, torch.float32)
            branch = self.unet.cache_branch_id
            self.unet.set_deepcache(-1)             # the calibration pass runs the plan without DeepCache
            self.unet.set_context(ctx)
            self.unet.calibrate_fp8(lat, 2, list(self.FP8_CALIBRATION_TIMESTEPS), margin=margin)
            self.unet.set_deepcache(branch)         # (the caller sets its own context next)
            how = "calibrated on the fixed seeded batch"
        self.fp8_scales = dict(self.unet.fp8_scales())
        self._fp8_calibrated = True
        if "e4m3 activation scales" not in self.weights_source:
            self.weights_source += f" (per-tensor e4m3 activation scales {how})"
        return self.fp8_scales

    # -- the sampling loop (src/models.py:32-335) ----------------------------------------------
    @torch.no_grad()
    def call(self, prompt: Union[str, List[str]] = None, height: Optional[int] = None, width: Optional[int] = None,
             num_inference_steps: int = 50, timesteps=None, sigmas=None, guidance_scale: float = 7.5,
             negative_prompt=None, num_images_per_prompt: int = 1, eta: float = 0.0, generator=None,
             latents: Optional[torch.Tensor] = None, prompt_embeds: Optional[torch.Tensor] = None,
             negative_prompt_embeds: Optional[torch.Tensor] = None, output_type: str = "pil",
             return_dict: bool = True, guidance_rescale: float = 0.0, step_noise: Optional[torch.Tensor] = None,
             collect_x0: bool = True, **kwargs):
        device, batch_size, do_cfg, ctx = self._begin(prompt, height, width, guidance_scale, negative_prompt,
                                                      num_images_per_prompt, prompt_embeds, negative_prompt_embeds,
                                                      guidance_rescale, timesteps, sigmas)
        cfgu = self.unet_config
        unet_batch = ctx.shape[0]

        self.scheduler.set_timesteps(num_inference_steps, device=device)                        # :167-169
        ts_host = list(self.scheduler._timesteps_list)
        latents = self.prepare_latents(batch_size, cfgu.in_channels, cfgu.sample_size * 8, cfgu.sample_size * 8,
                                       device, generator, latents)

        dc = self._deepcache
        self.unet.set_deepcache(dc.cache_branch_id if dc is not None else -1)
        self.unet.set_context(ctx)
        eps = self._eps_buffer(unet_batch, device)
        self._num_timesteps = len(ts_host)
        x0_preds = []
        is_lcm = hasattr(self.scheduler, "config") and "timestep_scaling" in self.scheduler.config

        torch.cuda.synchronize(device)
        start_time = time.time()                                                                   # :208
        for i, t in enumerate(ts_host):                                                            # :211
            mode = CACHE_OFF
            if dc is not None:
                # DeepCache: index of t in scheduler.timesteps, first step always full (A.5)
                cur = ts_host.index(t)      # first match, as DeepCache's list.index (duplicate PNDM timestep quirk)
                mode = CACHE_FULL_AND_STORE if (cur - 0) % dc.cache_interval == 0 else CACHE_SKIP
            self.unet.forward_latents(latents, unet_batch, float(t), out=eps, cache_mode=mode)     # :217-235
            kw = {}
            if is_lcm and step_noise is not None and i < len(ts_host) - 1:
                kw["noise"] = step_noise[i]
            step = self.scheduler.step_fused(eps, guidance_scale, latents, t, cfg=do_cfg,          # :238-261
                                             eta=eta, generator=generator, **kw)
            if len(step) == 1:                                                                      # :257-261
                latents = step[0]
            else:
                latents, x0 = step[0], step[1]
                if collect_x0:
                    x0_preds.append(x0[0:1])
        torch.cuda.synchronize(device)
        execution_time = time.time() - start_time                                                 # :284-285
        return self._finish(latents, x0_preds, output_type, return_dict, execution_time)


# --------------------------------------------------------------------------------------------------
# Variant pipelines: host-only control flow over the same kernels.
# --------------------------------------------------------------------------------------------------
def _is_dpm(s) -> bool:
    return hasattr(s, "model_outputs") and hasattr(s, "convert_model_output")


def _push_history(sched, noise_pred, latents):
    """History hand-off (``src/models.py:603-611``, ``:1025-1033``, ``:1045-1053``): shift the other
    scheduler's ``model_outputs`` and append its conversion of this noise prediction.  ``sample=latents``
    is the latents AFTER the step, as the reference writes it."""
    out = sched.convert_model_output(noise_pred, sample=latents)
    model_output = out[0] if isinstance(out, tuple) else out
    for k in range(sched.config.solver_order - 1):
        sched.model_outputs[k] = sched.model_outputs[k + 1]
    sched.model_outputs[-1] = model_output


class _VariantBase(StableDiffusionModel):
    """Shared step of the variant loops: UNet forward, then CFG combine + ``scheduler.step`` in the one fused
    launch, returning the new latents and the CFG-combined noise prediction the hand-off needs."""

    def _step_with(self, sched, eps, latents, t, unet_batch, do_cfg, guidance_scale, eta, generator, x0_preds):
        self.unet.forward_latents(latents, unet_batch, float(t), out=eps, cache_mode=CACHE_OFF)
        step = sched.step_fused(eps, guidance_scale, latents, t, cfg=do_cfg, eta=eta, generator=generator)
        if len(step) == 2:
            x0_preds.append(step[1][0:1])
        return step[0]

    @staticmethod
    def _combined(eps, do_cfg, guidance_scale):
        if not do_cfg:
            return eps
        u, c = eps.chunk(2)
        return u + guidance_scale * (c - u)                                                         # :238-242


@models_registry.add_to_registry("stable_diffusion_model_two_schedulers")
class StableDiffusionModelTwoSchedulers(_VariantBase):
    """``src/models.py:338-730``: ``scheduler_first`` for the first ``num_step_switch`` steps, then
    ``scheduler_second`` from the timestep ``switch_timestamp`` selects.  The second scheduler is handed the
    FIRST scheduler's timesteps as a custom schedule (``:488-492``; ``num_inference_steps_second`` is
    accepted and, as in the reference, not used)."""
    scheduler_first = None
    scheduler_second = None

    @staticmethod
    def switch_timestamp(timesteps_first, timesteps_second, num_step_switch, type_switch="closest"):
        """``src/models.py:704-730``."""
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

    @torch.no_grad()
    def call(self, prompt=None, height=None, width=None, num_inference_steps_first: int = 50,
             num_inference_steps_second: int = 50, num_step_switch: int = 10, type_switch: str = "closest",
             timesteps=None, sigmas=None, guidance_scale: float = 7.5, negative_prompt=None,
             num_images_per_prompt: int = 1, eta: float = 0.0, generator=None, latents=None, prompt_embeds=None,
             negative_prompt_embeds=None, output_type: str = "pil", return_dict: bool = True,
             guidance_rescale: float = 0.0, **kwargs):
        if self.scheduler_first is None or self.scheduler_second is None:
            raise ValueError("scheduler_first / scheduler_second must be set (two_schedulers.py:44-62)")
        device, batch_size, do_cfg, ctx = self._begin(prompt, height, width, guidance_scale, negative_prompt,
                                                      num_images_per_prompt, prompt_embeds, negative_prompt_embeds,
                                                      guidance_rescale, timesteps, sigmas)
        c = self.unet_config
        unet_batch = ctx.shape[0]
        self.scheduler_first.set_timesteps(num_inference_steps_first, device=device)               # :484-487
        self.scheduler_second.set_timesteps(device=device, timesteps=self.scheduler_first._timesteps_list)  # :488-492
        first, second = self.switch_timestamp(self.scheduler_first._timesteps_list,
                                              self.scheduler_second._timesteps_list, num_step_switch, type_switch)
        self.scheduler = self.scheduler_first                     # prepare_latents reads init_noise_sigma
        latents = self.prepare_latents(batch_size, c.in_channels, c.sample_size * 8, c.sample_size * 8, device,
                                       generator, latents)
        self.unet.set_deepcache(-1)
        self.unet.set_context(ctx)
        eps = self._eps_buffer(unet_batch, device)
        self._num_timesteps = len(first) + len(second)                                              # :545
        x0_preds = []
        hand_off = _is_dpm(self.scheduler_second)
        torch.cuda.synchronize(device)
        start_time = time.time()
        for i, t in enumerate(first + second):                                                      # :550
            in_first = i < len(first)
            sched = self.scheduler_first if in_first else self.scheduler_second
            latents = self._step_with(sched, eps, latents, t, unet_batch, do_cfg, guidance_scale, eta, generator,
                                      x0_preds)
            if in_first and hand_off:                                                               # :603-611
                _push_history(self.scheduler_second, self._combined(eps, do_cfg, guidance_scale), latents)
        torch.cuda.synchronize(device)
        return self._finish(latents, x0_preds, output_type, return_dict, time.time() - start_time)


@models_registry.add_to_registry("stable_diffusion_model_interliving_schedulers")
class StableDiffusionModelInterlivingSchedulers(_VariantBase):
    """``src/models.py:733-1136``: groups of ``solver_order`` steps of ``scheduler_main`` listed in
    ``interliving_steps`` are replaced by one step of ``scheduler_inter`` at the group's first timestep; each
    scheduler's multistep history is fed the other's noise predictions."""
    scheduler_main = None
    scheduler_inter = None

    @staticmethod
    def interleave_plan(timesteps_main, solver_order, interliving_steps):
        """``src/models.py:952-966``: returns (timesteps that run, those of them the inter scheduler takes)."""
        keep, t_inter = [], []
        for i, t in enumerate(int(x) for x in timesteps_main):
            if i // solver_order in interliving_steps:
                if i % solver_order != 0:
                    continue
                t_inter.append(t)
            keep.append(t)
        return keep, t_inter

    @torch.no_grad()
    def call(self, prompt=None, height=None, width=None, num_inference_steps: int = 50, interliving_steps=None,
             timesteps=None, sigmas=None, guidance_scale: float = 7.5, negative_prompt=None,
             num_images_per_prompt: int = 1, eta: float = 0.0, generator=None, latents=None, prompt_embeds=None,
             negative_prompt_embeds=None, output_type: str = "pil", return_dict: bool = True,
             guidance_rescale: float = 0.0, **kwargs):
        if self.scheduler_main is None or self.scheduler_inter is None:
            raise ValueError("scheduler_main / scheduler_inter must be set (interliving_exp.py:41-62)")
        interliving_steps = list(interliving_steps or [])
        device, batch_size, do_cfg, ctx = self._begin(prompt, height, width, guidance_scale, negative_prompt,
                                                      num_images_per_prompt, prompt_embeds, negative_prompt_embeds,
                                                      guidance_rescale, timesteps, sigmas)
        c = self.unet_config
        unet_batch = ctx.shape[0]
        order = self.scheduler_main.config.solver_order
        self.scheduler_main.set_timesteps(num_inference_steps, device=device)                       # :880-886
        self.scheduler_inter.set_timesteps(num_inference_steps // order, device=device)             # :888-894
        keep, t_inter = self.interleave_plan(self.scheduler_main._timesteps_list, order, interliving_steps)
        self.scheduler = self.scheduler_main
        latents = self.prepare_latents(batch_size, c.in_channels, c.sample_size * 8, c.sample_size * 8, device,
                                       generator, latents)
        self.unet.set_deepcache(-1)
        self.unet.set_context(ctx)
        eps = self._eps_buffer(unet_batch, device)
        self._num_timesteps = len(self.scheduler_main._timesteps_list) - len(interliving_steps)     # :946
        x0_preds = []
        torch.cuda.synchronize(device)
        start_time = time.time()
        for t in keep:
            if t in t_inter:                                                                        # :1008-1034
                latents = self._step_with(self.scheduler_inter, eps, latents, t, unet_batch, do_cfg, guidance_scale,
                                          eta, generator, x0_preds)
                _push_history(self.scheduler_main, self._combined(eps, do_cfg, guidance_scale), latents)
            else:                                                                                   # :1035-1054
                latents = self._step_with(self.scheduler_main, eps, latents, t, unet_batch, do_cfg, guidance_scale,
                                          eta, generator, x0_preds)
                if _is_dpm(self.scheduler_inter):
                    _push_history(self.scheduler_inter, self._combined(eps, do_cfg, guidance_scale), latents)
        torch.cuda.synchronize(device)
        return self._finish(latents, x0_preds, output_type, return_dict, time.time() - start_time)


@models_registry.add_to_registry("stable_diffusion_model_skip_timesteps")
class StableDiffusionModelSkipTimesteps(_VariantBase):
    """``src/models.py:1138-1467``: the plain loop with the loop indices in ``skip_timesteps`` skipped
    (``:1327-1330``).  A multistep scheduler's internal step index is not advanced for a skipped step -- the
    reference's behaviour, kept."""

    @torch.no_grad()
    def call(self, prompt=None, height=None, width=None, num_inference_steps: int = 50, skip_timesteps=None,
             timesteps=None, sigmas=None, guidance_scale: float = 7.5, negative_prompt=None,
             num_images_per_prompt: int = 1, eta: float = 0.0, generator=None, latents=None, prompt_embeds=None,
             negative_prompt_embeds=None, output_type: str = "pil", return_dict: bool = True,
             guidance_rescale: float = 0.0, **kwargs):
        skip = set(int(i) for i in (skip_timesteps or []))
        device, batch_size, do_cfg, ctx = self._begin(prompt, height, width, guidance_scale, negative_prompt,
                                                      num_images_per_prompt, prompt_embeds, negative_prompt_embeds,
                                                      guidance_rescale, timesteps, sigmas)
        c = self.unet_config
        unet_batch = ctx.shape[0]
        self.scheduler.set_timesteps(num_inference_steps, device=device)
        ts_host = list(self.scheduler._timesteps_list)
        latents = self.prepare_latents(batch_size, c.in_channels, c.sample_size * 8, c.sample_size * 8, device,
                                       generator, latents)
        self.unet.set_deepcache(-1)
        self.unet.set_context(ctx)
        eps = self._eps_buffer(unet_batch, device)
        self._num_timesteps = len(ts_host)                                                          # :1322
        x0_preds = []
        torch.cuda.synchronize(device)
        start_time = time.time()
        for i, t in enumerate(ts_host):
            if i in skip:                                                                           # :1327-1330
                continue
            latents = self._step_with(self.scheduler, eps, latents, t, unet_batch, do_cfg, guidance_scale, eta,
                                      generator, x0_preds)
        torch.cuda.synchronize(device)
        return self._finish(latents, x0_preds, output_type, return_dict, time.time() - start_time)


def _fuse_synthetic_lora(sd, seed: int, scale: float, rank: int):
    """``fuse_lora``: W += scale * B @ A on the attention projections (A.6.3).  The LCM-LoRA
    adapter is a network fetch (src/experiments/consistency_model.py:20), so a seeded synthetic
    low-rank update of the same structure is fused instead."""
    g = torch.Generator().manual_seed(seed)
    for name in list(sd.keys()):
        if any(k in name for k in (".to_q.weight", ".to_k.weight", ".to_v.weight", ".to_out.0.weight")):
            w = sd[name]
            o, i = w.shape
            a = torch.randn((rank, i), generator=g) / (i ** 0.5)
            b = torch.randn((o, rank), generator=g) * (0.02 / rank ** 0.5)
            sd[name] = (w + scale * (b @ a)).to(torch.bfloat16).float()
