"""``methods_registry["interliving_schedulers"]`` (``src/experiments/interliving_exp.py:10-171``): the main
(multistep) scheduler with groups of its steps replaced by single steps of the inter scheduler."""
from ..registry import methods_registry, schedulers_registry
from .base_experiment import BaseMethod


@methods_registry.add_to_registry("interliving_schedulers")
class InterlivingSchedulerMethod(BaseMethod):
    _ROLES = ("main", "inter")

    def setup_exp_params(self):
        ep = self.config.experiment_params
        self.num_inference_steps_first = ep.num_inference_steps_first
        self.interliving_steps = ep.interliving_steps
        for role in self._ROLES:
            for key in ("order_solver", "algorithm_type", "final_sigmas_type"):
                setattr(self, f"{role}_{key}", ep.get(f"{role}_{key}", ""))

    def setup_scheduler(self):
        # interliving_exp.py:45-62: a copy of the checkpoint scheduler's config with three keys overwritten;
        # "" (= key absent from the YAML) keeps the scheduler's default
        for role in self._ROLES:
            cfg = dict(self.model.scheduler.config)
            for cfg_key, attr in (("solver_order", "order_solver"), ("algorithm_type", "algorithm_type"),
                                  ("final_sigmas_type", "final_sigmas_type")):
                value = getattr(self, f"{role}_{attr}")
                if value not in ("", None):
                    cfg[cfg_key] = value
            cls = schedulers_registry[self.config.scheduler[f"scheduler_{role}"]]
            setattr(self.model, f"scheduler_{role}", cls.from_config(cfg))

    def run_experiment(self):
        tag = lambda steps: " ".join(map(str, steps))
        self.sweep(list(zip(self.num_inference_steps_first, self.interliving_steps)),
                   lambda p: {"num_inference_steps": p[0], "interliving_steps": list(p[1])},
                   lambda p: f"Step main: {p[0]}, Inter steps:{tag(p[1])}",
                   extra=lambda p: {"num_inference_steps": p[0], "num_inter_steps": tag(p[1])})
