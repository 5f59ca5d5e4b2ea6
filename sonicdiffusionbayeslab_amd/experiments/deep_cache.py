"""``methods_registry["deep_cache"]`` (``src/experiments/deep_cache.py:10-58``).

The reference does not swap the scheduler here (``:17-18``) and so runs the checkpoint's PNDM:
that is the default (``pndm_scheduler``); a ``scheduler.scheduler_name`` key in the YAML selects
another plugin (BASELINE config 4 quotes DeepCache on DDIM 50 steps, see configs/deep_cache_config.yaml)."""
from collections import defaultdict

from ..deepcache import DeepCacheSDHelper
from ..registry import methods_registry, schedulers_registry
from .base_experiment import BaseMethod


@methods_registry.add_to_registry("deep_cache")
class DeepCacheMethod(BaseMethod):
    def setup_exp_params(self):
        self.cache_interval = self.config.experiment_params.cache_interval
        self.cache_branch_id = self.config.experiment_params.get("cache_branch_id", 0)
        self.num_inference_steps = self.config.experiment_params.num_inference_steps

    def setup_scheduler(self):
        name = self.config.get("scheduler", {}).get("scheduler_name", "pndm_scheduler")
        self.model.scheduler = schedulers_registry[name].from_config(self.model.scheduler.config)

    def run_experiment(self):
        batch_size = self.config.inference.get("batch_size", 1)
        for cache_interval in self.cache_interval:
            helper = DeepCacheSDHelper(pipe=self.model)
            helper.set_params(cache_interval=cache_interval, cache_branch_id=self.cache_branch_id)
            helper.enable()
            self.metric_dict = defaultdict(list)
            for steps in self.num_inference_steps:
                self.model.to(self.device)
                gen_images, _ = self.generate(self.test_dataset.batches(batch_size), steps, batch_size)
                self.model.to("cpu")
                self.validate(f"{self.config.experiment_name}, Inference steps: {steps}, Cache interval: {cache_interval}",
                              additional_values={"Cache interval": cache_interval}, n_images=len(gen_images))
            helper.disable()
