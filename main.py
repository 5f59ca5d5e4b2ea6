"""The reference's command line (``main.py:10-24``): ``python main.py --config ddim_config.yaml`` reads
``./configs/<file>``, seeds the host RNGs and runs ``methods_registry[experiment.method](config).run_experiment()``."""
import argparse
import random

import torch

import sonicdiffusionbayeslab_amd  # noqa: F401  (importing the package registers the plugins)
from sonicdiffusionbayeslab_amd.config import load_named_config
from sonicdiffusionbayeslab_amd.registry import methods_registry


def setup_seed(seed: int) -> None:
    """``src/utils/model_utils.py:15-17``."""
    random.seed(seed)
    torch.random.manual_seed(seed)


def run(config_file: str) -> None:
    cfg = load_named_config(config_file)
    setup_seed(cfg.experiment.get("seed", 29))
    method_cls = methods_registry[cfg.experiment.method]
    method_cls(cfg).run_experiment()


main = run      # the reference's name for the same function


if __name__ == "__main__":
    cli = argparse.ArgumentParser(description="Sonic Diffusion sampling experiments on the MI355X-native hot path")
    cli.add_argument("--config", type=str, default="config.yaml", help="file name inside ./configs")
    run(cli.parse_args().config)
