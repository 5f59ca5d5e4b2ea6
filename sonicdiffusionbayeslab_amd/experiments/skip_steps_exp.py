"""``methods_registry["skip_steps"]`` (``src/experiments/skip_steps_exp.py:10-135``): the plain loop with some
loop indices skipped, to study which steps matter."""
from collections import defaultdict

from ..registry import methods_registry
from .base_experiment import BaseMethod


@methods_registry.add_to_registry("skip_steps")
class SkipStepsMethod(BaseMethod):
    def setup_exp_params(self):
        ep = self.config.experiment_params
        self.skip_steps = ep.skip_steps
        self.num_inference_steps = ep.num_inference_steps
        self.solver_order = ep.solver_order
        self.algorithm_type = ep.algorithm_type
        self.final_sigmas_type = ep.final_sigmas_type
        self.batch_size = self.config.inference.get("batch_size", 1)

    def setup_scheduler(self, **kwargs):
        return super().setup_scheduler(solver_order=self.solver_order, algorithm_type=self.algorithm_type,
                                       final_sigmas_type=self.final_sigmas_type)

    def generate(self, test_dataloader, num_inference_steps, skip_steps, batch_size=1, guidance_scale=7.5):
        gen_images_list, x0_preds = [], []
        for idx, batch in enumerate(test_dataloader):
            bc = self.config.inference.get("batch_count", None)
            if bc is not None and idx >= bc:
                break
            imgs, inference_time, x0_preds = self.model(
                batch["prompt"], guidance_scale=guidance_scale, generator=self.generator,
                num_inference_steps=num_inference_steps, skip_timesteps=list(skip_steps),
                output_type=self.config.inference.get("output_type", "latent"))
            imgs = imgs.images.cpu()
            gen_images_list.extend(imgs[i] for i in range(imgs.shape[0]))
            self.time_metric.update(inference_time, batch_size)
        return gen_images_list, x0_preds

    def run_experiment(self):
        self.metric_dict = defaultdict(list)
        for steps, skip in zip(self.num_inference_steps, self.skip_steps):
            self.model.to(self.device)
            gen_images, _ = self.generate(self.test_dataset.batches(self.batch_size), steps, skip, self.batch_size)
            self.model.to("cpu")
            tag = " ".join(map(str, skip))
            self.validate(f"{self.config.experiment_name}, Step main: {steps}, Skip steps:{tag}",
                          additional_values={"num_inference_steps": steps, "skip_steps": tag},
                          n_images=len(gen_images))
