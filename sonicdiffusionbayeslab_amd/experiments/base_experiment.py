"""Harness base class with the reference's hook order and benchmark semantics
(``src/experiments/base_experiment.py:18-163``): 7 ``setup_*`` hooks in fixed order, one shared
``torch.Generator`` that is never reseeded, scheduler swap through the registry +
``from_config``, ``generate()`` looping prompt batches through ``self.model(...)`` and feeding
``time_metric``.  Quality metrics that need fetched models and the wandb logger are out of the
hot-path scope; results go to stdout as JSON lines.

Multi-GPU (new relative to the single-device reference): launched under
``torch.distributed.run`` (one process per GPU), every prompt batch is split contiguously over the ranks,
each rank samples its shard on its own weight replica, and ONE all-gather per batch returns the results to
every rank.  Every Gaussian a pipeline call consumes (initial latents, LCM re-noising, SDE-DPM-Solver noise, eta > 0, the
variant pipelines' draws) is drawn for the GLOBAL batch from the shared seeded CPU generator, in the single-process order,
and sliced per rank (``dist.randn`` / ``dist.shard_draws``), so the inputs of every image are the same for every world
size -- also for the stochastic samplers, ragged shards and ranks without a prompt."""
from __future__ import annotations

import contextlib
import json
import os
from abc import ABC, abstractmethod
from collections import defaultdict

import torch

from .. import dist as sdist
from ..dataset import PromptDataset
from ..registry import metrics_registry, models_registry, schedulers_registry


class BaseMethod(ABC):
    def __init__(self, config):
        self.config = config
        # one process per GPU under torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE); world 1 otherwise.
        # SD_DIST_BACKEND=gloo is the CPU-side rehearsal backend of the tests (ranks may then share one GPU)
        backend = os.environ.get("SD_DIST_BACKEND", "nccl" if torch.cuda.is_available() else "gloo")
        self.rank, local_rank, self.world = sdist.init_process_group(backend)
        ngpu = torch.cuda.device_count() if torch.cuda.is_available() else 0
        self.device = f"cuda:{local_rank % ngpu}" if ngpu else "cpu"
        if ngpu:
            torch.cuda.set_device(self.device)
        self.setup_exp_params()
        self.setup_generator()
        self.setup_model()
        self.setup_scheduler()
        self.setup_dataset()
        self.setup_metrics()
        self.setup_loggers()

    @abstractmethod
    def run_experiment(self):
        pass

    def setup_exp_params(self):
        pass

    def setup_generator(self):
        # the reference seeds a device generator (:51-53); a CPU generator makes the initial latents
        # identical for every world size and box
        self.generator = torch.Generator(device="cpu")
        self.generator.manual_seed(self.config.experiment.seed)

    def setup_model(self):
        model_name = self.config.model.model_name
        extra = {}
        if self.config.model.get("weight_dtype", None) is not None:      # key of this build: "bf16" | "fp8"
            extra["weight_dtype"] = self.config.model.weight_dtype
        self.model = models_registry[model_name].from_pretrained(
            self.config.model.pretrained_model,
            timestamps=self.config.model.get("timestamps", None),
            safety_checker=None,
            requires_safety_checker=False,
            torch_dtype=torch.float16,
            **extra,
        )
        self.model.to(self.device)

    def setup_scheduler(self, **kwargs):
        scheduler_name = self.config.scheduler.scheduler_name
        self.model.scheduler = schedulers_registry[scheduler_name].from_config(self.model.scheduler.config, **kwargs)

    def setup_dataset(self):
        self.test_dataset = PromptDataset(self.config.dataset.img_dataset, self.config.dataset.prompts)

    def setup_metrics(self):
        """``src/experiments/base_experiment.py:93-113``.  ``time_metric`` always; ``clip_score`` (``:96-98``) when
        ``quality_metrics.clip_score.model_name_or_path`` is a LOCAL checkpoint directory -- a hub name is a network
        fetch and the metric is then reported as not computable.  FID / ImageReward are not built."""
        self.metric_dict = defaultdict(list)
        self.time_metric = metrics_registry["time_metric"]()
        self.clip_score_gen_metric = None
        qm = self.config.get("quality_metrics", None)
        path = qm.get("clip_score", {}).get("model_name_or_path", None) if qm else None
        if path and os.path.isdir(str(path)):
            self.clip_score_gen_metric = metrics_registry["clip_score"](model_name_or_path=str(path))
        self.clip_score_source = str(path) if self.clip_score_gen_metric is not None else (
            f"not computable offline ({path!r} is not a local directory)" if path else "not configured")

    def setup_loggers(self):
        self.logger = None

    def generate(self, test_dataloader, steps=None, batch_size: int = 1, guidance_scale: float = 7.5, **call_kwargs):
        """One pass over the prompt batches through ``self.model(...)`` (``base_experiment.py:122-163``).
        ``steps`` becomes ``num_inference_steps``; pipelines with other step arguments (the variant
        pipelines) receive theirs through ``call_kwargs``."""
        if steps is not None:
            call_kwargs["num_inference_steps"] = steps
        limit = self.config.inference.get("batch_count", None)
        out_type = self.config.inference.get("output_type", "pt")        # the reference hard-codes "pt" (:145)
        images, x0_preds = [], []
        self.last_prompts = []                                           # prompt of every returned image, in order
        for idx, batch in enumerate(test_dataloader):
            if limit is not None and idx >= limit:
                break
            prompts = list(batch["prompt"])
            self.last_prompts.extend(prompts)
            sharded = sdist.active()
            lo, hi = sdist.shard_range(len(prompts), self.rank, self.world) if sharded else (0, len(prompts))
            local_prompts = prompts[lo:hi]
            if len(local_prompts) > 0:
                # every Gaussian of the call (initial latents, per-step noise of the stochastic samplers) is drawn for the
                # GLOBAL batch from the shared generator and sliced: dist.randn inside dist.shard_draws
                with (sdist.shard_draws(len(prompts), lo, hi) if sharded else contextlib.nullcontext()):
                    result, seconds, x0_preds = self.model(local_prompts, guidance_scale=guidance_scale,
                                                           generator=self.generator, output_type=out_type, **call_kwargs)
                local = result.images
            else:                       # more ranks than prompts in a ragged last batch
                local, seconds = self._empty_result(out_type), 0.0
            if sharded:                 # the ONE data-path collective; the slowest rank's loop time rides along (and, when a
                                        # rank had no prompt and therefore drew nothing, rank 0's generator state)
                local, seconds = sdist.gather_latents(local, self.world, len(prompts), seconds=seconds,
                                                      generator=self.generator)
            host = local.cpu()
            images.extend(host[i] for i in range(host.shape[0]))
            self.time_metric.update(seconds, batch_size)                 # configured size, as :161
        return images, x0_preds

    def _empty_result(self, out_type: str) -> torch.Tensor:
        ucfg = self.model.unet_config
        if out_type == "latent":
            return torch.empty((0, ucfg.in_channels, ucfg.sample_size, ucfg.sample_size), device=self.device)
        return torch.empty((0, 3, ucfg.sample_size * 8, ucfg.sample_size * 8), device=self.device)

    def sweep(self, points, call_kwargs, label, extra=None, guidance_scale: float = 7.5):
        """The loop every method's ``run_experiment`` is: for each sweep point move the model to the device,
        generate, move it back, report (``src/experiments/ddim.py:26-57`` and its siblings).
        ``call_kwargs(point)`` -> keyword arguments of the pipeline call, ``label(point)`` -> run name,
        ``extra(point)`` -> additional logged values."""
        batch_size = self.config.inference.get("batch_size", 1)
        self.metric_dict = defaultdict(list)
        for point in points:
            self.model.to(self.device)
            if hasattr(self.model, "calibrate_fp8") and not getattr(self.model, "_fp8_calibrated", True):
                self.model.calibrate_fp8()      # fp8 handles: an explicit set-up step on EVERY rank (also one whose shards are empty)
            images, _ = self.generate(self.test_dataset.batches(batch_size), None, batch_size,
                                      guidance_scale=guidance_scale, **call_kwargs(point))
            self.model.to("cpu")
            self.validate(f"{self.config.experiment_name}, {label(point)}",
                          additional_values=extra(point) if extra else None, n_images=len(images),
                          images=images, prompts=self.last_prompts)

    def clip_score(self, images, prompts, batch_size: int = 32):
        """``validate`` of the reference for the parity metric (``:198-201``): generated images -> uint8 by
        ``(img * 255).to(uint8)`` (truncation), CLIPScore over (image, prompt) pairs.  None when no local CLIP
        checkpoint was configured or the run produced latents."""
        m = self.clip_score_gen_metric
        if m is None or not images or images[0].dim() != 3 or images[0].shape[0] != 3:
            return None
        m.reset()
        for s in range(0, len(images), batch_size):
            gen = (torch.stack(images[s:s + batch_size]) * 255).to(torch.uint8).cpu()
            m.update(gen, prompts[s:s + batch_size])
        return float(m.compute())

    def validate(self, name_images, additional_values=None, n_images=0, images=None, prompts=None):
        """The hot-path metric, seconds / image over the loop (``time_metric``), and -- with a local CLIP checkpoint and
        decoded images -- the reference's parity metric ``clip_score``."""
        if additional_values:
            for k, v in additional_values.items():
                self.metric_dict[k].append(v)
        cs = self.clip_score(images, prompts) if (images is not None and self.rank == 0) else None
        if cs is not None:
            self.metric_dict["clip_score"].append(cs)
        t = float(self.time_metric.compute())
        self.metric_dict["nfe"].append(self.model.num_timesteps)
        self.metric_dict["time_metric"].append(t)
        self.time_metric.reset()
        if self.rank != 0:
            return
        print(json.dumps({"experiment": self.config.experiment_name, "run": name_images, "nfe": self.model.num_timesteps,
                          "images": n_images, "time_metric_s_per_image": t,
                          "images_per_s": (1.0 / t if t > 0 else None), "weights": self.model.weights_source,
                          "clip_score": cs, "clip_score_model": self.clip_score_source, "n_gpus": self.world,
                          "fp8_activation_scales": self._fp8_scale_report()}), flush=True)

    def _fp8_scale_report(self):
        """The e4m3 activation scales the run used (fp8 handles; None otherwise): count, range and a digest, so that two
        runs -- or two world sizes -- can be checked to have quantised identically."""
        sc = getattr(self.model, "fp8_scales", None)
        if not sc:
            return None
        import hashlib
        digest = hashlib.sha256(json.dumps(sorted(sc.items())).encode()).hexdigest()[:16]
        return {"tensors": len(sc), "min": min(sc.values()), "max": max(sc.values()), "sha256_16": digest}
