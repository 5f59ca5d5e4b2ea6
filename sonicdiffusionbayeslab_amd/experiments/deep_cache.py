"""``methods_registry["deep_cache"]`` (``src/experiments/deep_cache.py:10-58``).

The reference does not swap the scheduler here (``:17-18``) and so runs the checkpoint's PNDM:
that is the default (``pndm_scheduler``); a ``scheduler.scheduler_name`` key in the YAML selects
another plugin (BASELINE config 4 quotes DeepCache on DDIM 50 steps, see configs/deep_cache_config.yaml)."""
from ..deepcache import DeepCacheSDHelper
from ..registry import methods_registry, schedulers_registry
from .base_experiment import BaseMethod


@methods_registry.add_to_registry("deep_cache")
class DeepCacheMethod(BaseMethod):
    def setup_exp_params(self):
        ep = self.config.experiment_params
        self.cache_interval, self.num_inference_steps = ep.cache_interval, ep.num_inference_steps
        self.cache_branch_id = ep.get("cache_branch_id", 0)

    def setup_scheduler(self):
        name = self.config.get("scheduler", {}).get("scheduler_name", "pndm_scheduler")
        self.model.scheduler = schedulers_registry[name].from_config(self.model.scheduler.config)

    def run_experiment(self):
        # one helper per interval, enabled across the whole sweep of step counts (:23-58)
        for interval in self.cache_interval:
            helper = DeepCacheSDHelper(pipe=self.model)
            helper.set_params(cache_interval=interval, cache_branch_id=self.cache_branch_id)
            helper.enable()
            try:
                self.sweep(self.num_inference_steps, lambda n: {"num_inference_steps": n},
                           lambda n: f"Inference steps: {n}, Cache interval: {interval}",
                           extra=lambda n: {"Cache interval": interval})
            finally:
                helper.disable()
