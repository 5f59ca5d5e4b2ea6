"""``methods_registry["consistency_model"]`` (``src/experiments/consistency_model.py:9-52``):
LCM-LoRA fused into the UNet, LCM scheduler, guidance_scale 0 (no CFG)."""
from collections import defaultdict

from ..registry import methods_registry
from .base_experiment import BaseMethod


@methods_registry.add_to_registry("consistency_model")
class ConsistencyModelMethod(BaseMethod):
    def setup_exp_params(self):
        self.num_inference_steps = self.config.experiment_params.num_inference_steps
        self.guidance_scale = self.config.experiment_params.guidance_scale
        self.batch_size = self.config.inference.get("batch_size", 1)

    def setup_model(self):
        # load_lora_weights + fuse_lora (:20-21) happen on the host copy before upload
        from ..registry import models_registry
        import torch
        self.model = models_registry[self.config.model.model_name].from_pretrained(
            self.config.model.pretrained_model, safety_checker=None, requires_safety_checker=False,
            torch_dtype=torch.float16)
        self.model.load_lora_weights(self.config.experiment_params.adapter_id)
        self.model.fuse_lora()
        self.model.to(self.device)

    def run_experiment(self):
        self.metric_dict = defaultdict(list)
        for steps in self.num_inference_steps:
            self.model.to(self.device)
            gen_images, _ = self.generate(self.test_dataset.batches(self.batch_size), steps, self.batch_size,
                                          guidance_scale=self.guidance_scale)
            self.model.to("cpu")
            self.validate(f"{self.config.experiment_name}, Inference steps: {steps}", n_images=len(gen_images))
