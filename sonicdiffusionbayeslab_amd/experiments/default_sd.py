"""``methods_registry["default"]`` (``src/experiments/default_sd.py:10-100``): the checkpoint's own
scheduler (PNDM/PLMS, never swapped: ``:15-16``) swept over ``num_inference_steps``."""
from collections import defaultdict

from ..registry import methods_registry, schedulers_registry
from .base_experiment import BaseMethod


@methods_registry.add_to_registry("default")
class DefaultStableDiffusion(BaseMethod):
    def setup_exp_params(self):
        self.num_inference_steps = self.config.experiment_params.num_inference_steps

    def setup_scheduler(self):
        self.model.scheduler = schedulers_registry["pndm_scheduler"].from_config(self.model.scheduler.config)

    def run_experiment(self):
        batch_size = self.config.inference.get("batch_size", 1)
        self.metric_dict = defaultdict(list)
        for steps in self.num_inference_steps:
            self.model.to(self.device)
            gen_images, _ = self.generate(self.test_dataset.batches(batch_size), steps, batch_size)
            self.model.to("cpu")
            self.validate(f"{self.config.experiment_name}, Inference steps: {steps}", n_images=len(gen_images))
