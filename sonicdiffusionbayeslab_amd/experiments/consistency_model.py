"""``methods_registry["consistency_model"]`` (``src/experiments/consistency_model.py:9-52``):
LCM-LoRA fused into the UNet, LCM scheduler, guidance_scale 0 (no CFG)."""
from ..registry import methods_registry
from .base_experiment import BaseMethod


@methods_registry.add_to_registry("consistency_model")
class ConsistencyModelMethod(BaseMethod):
    def setup_exp_params(self):
        ep = self.config.experiment_params
        self.num_inference_steps, self.guidance_scale, self.adapter_id = ep.num_inference_steps, ep.guidance_scale, ep.adapter_id

    def setup_model(self):
        # the adapter is fused into the HOST copy of the weights, i.e. before the upload super().setup_model()
        # would trigger (load_lora_weights + fuse_lora, :20-21)
        self.device_after_fuse, self.device = self.device, "cpu"
        super().setup_model()
        self.model.load_lora_weights(self.adapter_id)
        self.model.fuse_lora()
        self.device = self.device_after_fuse
        self.model.to(self.device)

    def run_experiment(self):
        self.sweep(self.num_inference_steps, lambda n: {"num_inference_steps": n}, lambda n: f"Inference steps: {n}",
                   guidance_scale=self.guidance_scale)
