"""``methods_registry["skip_steps"]`` (``src/experiments/skip_steps_exp.py:10-135``): the plain loop with some
loop indices skipped, to study which steps matter."""
from ..registry import methods_registry
from .base_experiment import BaseMethod


@methods_registry.add_to_registry("skip_steps")
class SkipStepsMethod(BaseMethod):
    def setup_exp_params(self):
        ep = self.config.experiment_params
        self.skip_steps, self.num_inference_steps = ep.skip_steps, ep.num_inference_steps
        self.solver_order, self.algorithm_type, self.final_sigmas_type = ep.solver_order, ep.algorithm_type, ep.final_sigmas_type

    def setup_scheduler(self, **kwargs):
        return super().setup_scheduler(solver_order=self.solver_order, algorithm_type=self.algorithm_type,
                                       final_sigmas_type=self.final_sigmas_type)

    def run_experiment(self):
        tag = lambda steps: " ".join(map(str, steps))
        self.sweep(list(zip(self.num_inference_steps, self.skip_steps)),
                   lambda p: {"num_inference_steps": p[0], "skip_timesteps": list(p[1])},
                   lambda p: f"Step main: {p[0]}, Skip steps:{tag(p[1])}",
                   extra=lambda p: {"num_inference_steps": p[0], "skip_steps": tag(p[1])})
