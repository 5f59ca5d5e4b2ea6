"""``methods_registry["dpm_solver"]`` (``src/experiments/dpm_solver.py:9-69``).

Appendix-B quirk #1: the reference reads ``experiment_params.algorithm_type`` and
``.final_sigmas_type`` which ``configs/dpm_solver_config.yaml`` does not define; the intended
defaults (BASELINE config 3: DPM-Solver++, final sigma zero) are applied when absent."""
from ..registry import methods_registry
from .base_experiment import BaseMethod


@methods_registry.add_to_registry("dpm_solver")
class DPMSolverMethod(BaseMethod):
    def setup_exp_params(self):
        ep = self.config.experiment_params
        self.num_inference_steps = ep.num_inference_steps
        self.solver_order = ep.solver_order
        self.algorithm_type = ep.get("algorithm_type", "dpmsolver++")
        self.final_sigmas_type = ep.get("final_sigmas_type", "zero" if self.algorithm_type.endswith("++") else "sigma_min")

    def setup_scheduler(self, **kwargs):
        return super().setup_scheduler(solver_order=self.solver_order, algorithm_type=self.algorithm_type,
                                       final_sigmas_type=self.final_sigmas_type)

    def run_experiment(self):
        self.sweep(self.num_inference_steps, lambda n: {"num_inference_steps": n},
                   lambda n: f"Solver order: {self.solver_order}, Inference steps: {n}")
