"""``methods_registry["ddim"]`` (``src/experiments/ddim.py:11-57``): sweep over num_inference_steps."""
from ..registry import methods_registry
from .base_experiment import BaseMethod


@methods_registry.add_to_registry("ddim")
class DDIMMethod(BaseMethod):
    def setup_exp_params(self):
        self.num_inference_steps = self.config.experiment_params.num_inference_steps

    def run_experiment(self):
        self.sweep(self.num_inference_steps, lambda n: {"num_inference_steps": n}, lambda n: f"Inference steps: {n}")
