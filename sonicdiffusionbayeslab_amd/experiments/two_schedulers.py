"""``methods_registry["two_schedulers"]`` (``src/experiments/two_schedulers.py:10-173``): a sweep over
(steps of the first scheduler, steps of the second, switch step) with the two-scheduler pipeline."""
from collections import defaultdict

from ..registry import methods_registry, schedulers_registry
from .base_experiment import BaseMethod


def _clean(**kw):
    """The reference passes "" for parameters the YAML leaves out (``.get(name, "")``) and misspells
    ``solver_order`` as ``sovler_order`` (two_schedulers.py:50,58), so the order never reaches
    ``from_config``; unset / empty values keep the scheduler's default here as well."""
    return {k: v for k, v in kw.items() if v not in ("", None)}


@methods_registry.add_to_registry("two_schedulers")
class TwoSchedulerMethod(BaseMethod):
    def setup_exp_params(self):
        ep = self.config.experiment_params
        self.num_inference_steps_first = ep.num_inference_steps_first
        self.num_inference_steps_second = ep.num_inference_steps_second
        self.num_step_switch = ep.num_step_switch
        self.first_order_solver = ep.get("first_order_solver", "")
        self.second_order_solver = ep.get("second_order_solver", "")
        self.first_algorithm_type = ep.get("first_algorithm_type", "")
        self.second_algorithm_type = ep.get("second_algorithm_type", "")
        self.first_final_sigmas_type = ep.get("first_final_sigmas_type", "")
        self.second_final_sigmas_type = ep.get("second_final_sigmas_type", "")
        self.type_switch = ep.type_switch
        self.batch_size = self.config.inference.get("batch_size", 1)

    def setup_scheduler(self):
        base = self.model.scheduler.config
        self.model.scheduler_first = schedulers_registry[self.config.scheduler.scheduler_first].from_config(
            base, **_clean(algorithm_type=self.first_algorithm_type, final_sigmas_type=self.first_final_sigmas_type))
        self.model.scheduler_second = schedulers_registry[self.config.scheduler.scheduler_second].from_config(
            base, **_clean(algorithm_type=self.second_algorithm_type, final_sigmas_type=self.second_final_sigmas_type))

    def generate(self, test_dataloader, num_inference_steps_first, num_inference_steps_second, num_step_switch,
                 batch_size=1, guidance_scale=7.5):
        gen_images_list, x0_preds = [], []
        for idx, batch in enumerate(test_dataloader):
            bc = self.config.inference.get("batch_count", None)
            if bc is not None and idx >= bc:
                break
            imgs, inference_time, x0_preds = self.model(
                batch["prompt"], guidance_scale=guidance_scale, generator=self.generator,
                num_inference_steps_first=num_inference_steps_first,
                num_inference_steps_second=num_inference_steps_second, num_step_switch=num_step_switch,
                type_switch=self.type_switch, output_type=self.config.inference.get("output_type", "latent"))
            imgs = imgs.images.cpu()
            gen_images_list.extend(imgs[i] for i in range(imgs.shape[0]))
            self.time_metric.update(inference_time, batch_size)
        return gen_images_list, x0_preds

    def run_experiment(self):
        self.metric_dict = defaultdict(list)
        for n1, n2, sw in zip(self.num_inference_steps_first, self.num_inference_steps_second, self.num_step_switch):
            self.model.to(self.device)
            gen_images, _ = self.generate(self.test_dataset.batches(self.batch_size), n1, n2, sw, self.batch_size)
            self.model.to("cpu")
            self.validate(f"{self.config.experiment_name}, Step first: {n1}, Step second: {n2}, Switch: {sw}",
                          additional_values={"num_inference_steps_first": n1, "num_inference_steps_second": n2,
                                             "switch_step": sw}, n_images=len(gen_images))
