"""``methods_registry["default"]`` (``src/experiments/default_sd.py:10-100``): the checkpoint's own
scheduler (PNDM/PLMS, never swapped: ``:15-16``) swept over ``num_inference_steps``."""
from ..registry import methods_registry, schedulers_registry
from .base_experiment import BaseMethod


@methods_registry.add_to_registry("default")
class DefaultStableDiffusion(BaseMethod):
    def setup_exp_params(self):
        self.num_inference_steps = self.config.experiment_params.num_inference_steps

    def setup_scheduler(self):
        self.model.scheduler = schedulers_registry["pndm_scheduler"].from_config(self.model.scheduler.config)

    def run_experiment(self):
        self.sweep(self.num_inference_steps, lambda n: {"num_inference_steps": n}, lambda n: f"Inference steps: {n}")
