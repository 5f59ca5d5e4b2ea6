"""``methods_registry["two_schedulers"]`` (``src/experiments/two_schedulers.py:10-173``): a sweep over
(steps of the first scheduler, steps of the second, switch step) with the two-scheduler pipeline."""
from ..registry import methods_registry, schedulers_registry
from .base_experiment import BaseMethod


def _given(**kw):
    """The reference passes "" for parameters the YAML leaves out (``.get(name, "")``) and misspells
    ``solver_order`` as ``sovler_order`` (two_schedulers.py:50,58), so the order never reaches ``from_config``
    and an absent ``algorithm_type`` arrives as "" (which diffusers rejects); unset / empty values keep the
    scheduler's default here."""
    return {k: v for k, v in kw.items() if v not in ("", None)}


@methods_registry.add_to_registry("two_schedulers")
class TwoSchedulerMethod(BaseMethod):
    _SIDES = ("first", "second")

    def setup_exp_params(self):
        ep = self.config.experiment_params
        self.num_inference_steps_first = ep.num_inference_steps_first
        self.num_inference_steps_second = ep.num_inference_steps_second
        self.num_step_switch = ep.num_step_switch
        self.type_switch = ep.type_switch
        for side in self._SIDES:
            for key in ("order_solver", "algorithm_type", "final_sigmas_type"):
                setattr(self, f"{side}_{key}", ep.get(f"{side}_{key}", ""))

    def setup_scheduler(self):
        for side in self._SIDES:
            cls = schedulers_registry[self.config.scheduler[f"scheduler_{side}"]]
            sched = cls.from_config(self.model.scheduler.config,
                                    **_given(algorithm_type=getattr(self, f"{side}_algorithm_type"),
                                             final_sigmas_type=getattr(self, f"{side}_final_sigmas_type")))
            setattr(self.model, f"scheduler_{side}", sched)

    def run_experiment(self):
        points = list(zip(self.num_inference_steps_first, self.num_inference_steps_second, self.num_step_switch))
        self.sweep(points,
                   lambda p: {"num_inference_steps_first": p[0], "num_inference_steps_second": p[1],
                              "num_step_switch": p[2], "type_switch": self.type_switch},
                   lambda p: f"Step first: {p[0]}, Step second: {p[1]}, Switch: {p[2]}",
                   extra=lambda p: {"num_inference_steps_first": p[0], "num_inference_steps_second": p[1],
                                    "switch_step": p[2]})
