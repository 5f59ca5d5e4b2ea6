"""``methods_registry["interliving_schedulers"]`` (``src/experiments/interliving_exp.py:10-171``): the main
(multistep) scheduler with groups of its steps replaced by single steps of the inter scheduler."""
from collections import defaultdict

from ..registry import methods_registry, schedulers_registry
from .base_experiment import BaseMethod


@methods_registry.add_to_registry("interliving_schedulers")
class InterlivingSchedulerMethod(BaseMethod):
    def setup_exp_params(self):
        ep = self.config.experiment_params
        self.num_inference_steps_first = ep.num_inference_steps_first
        self.interliving_steps = ep.interliving_steps
        self.main_order_solver = ep.get("main_order_solver", "")
        self.inter_order_solver = ep.get("inter_order_solver", "")
        self.main_algorithm_type = ep.get("main_algorithm_type", "")
        self.inter_algorithm_type = ep.get("inter_algorithm_type", "")
        self.main_final_sigmas_type = ep.get("main_final_sigmas_type", "")
        self.inter_final_sigmas_type = ep.get("inter_final_sigmas_type", "")
        self.batch_size = self.config.inference.get("batch_size", 1)

    def _from_base(self, name, order, algorithm_type, final_sigmas_type):
        # interliving_exp.py:45-62: a copy of the checkpoint scheduler's config with three keys overwritten;
        # "" (= key absent from the YAML) keeps the scheduler's default
        cfg = dict(self.model.scheduler.config)
        for k, v in (("solver_order", order), ("algorithm_type", algorithm_type), ("final_sigmas_type", final_sigmas_type)):
            if v not in ("", None):
                cfg[k] = v
        return schedulers_registry[name].from_config(cfg)

    def setup_scheduler(self):
        self.model.scheduler_main = self._from_base(self.config.scheduler.scheduler_main, self.main_order_solver,
                                                    self.main_algorithm_type, self.main_final_sigmas_type)
        self.model.scheduler_inter = self._from_base(self.config.scheduler.scheduler_inter, self.inter_order_solver,
                                                     self.inter_algorithm_type, self.inter_final_sigmas_type)

    def generate(self, test_dataloader, num_inference_steps, interliving_steps, batch_size=1, guidance_scale=7.5):
        gen_images_list, x0_preds = [], []
        for idx, batch in enumerate(test_dataloader):
            bc = self.config.inference.get("batch_count", None)
            if bc is not None and idx >= bc:
                break
            imgs, inference_time, x0_preds = self.model(
                batch["prompt"], guidance_scale=guidance_scale, generator=self.generator,
                num_inference_steps=num_inference_steps, interliving_steps=list(interliving_steps),
                output_type=self.config.inference.get("output_type", "latent"))
            imgs = imgs.images.cpu()
            gen_images_list.extend(imgs[i] for i in range(imgs.shape[0]))
            self.time_metric.update(inference_time, batch_size)
        return gen_images_list, x0_preds

    def run_experiment(self):
        self.metric_dict = defaultdict(list)
        for steps, inter in zip(self.num_inference_steps_first, self.interliving_steps):
            self.model.to(self.device)
            gen_images, _ = self.generate(self.test_dataset.batches(self.batch_size), steps, inter, self.batch_size)
            self.model.to("cpu")
            tag = " ".join(map(str, inter))
            self.validate(f"{self.config.experiment_name}, Step main: {steps}, Inter steps:{tag}",
                          additional_values={"num_inference_steps": steps, "num_inter_steps": tag},
                          n_images=len(gen_images))
