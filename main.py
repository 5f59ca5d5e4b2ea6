"""Entry point with the reference's CLI (``main.py:10-24``): ``python main.py --config ddim_config.yaml``
loads ``./configs/<file>``, seeds, and dispatches ``methods_registry[method](config).run_experiment()``."""
import argparse
import random

import torch

import sonicdiffusionbayeslab_amd  # noqa: F401  (registers the plugins)
from sonicdiffusionbayeslab_amd.config import load_named_config
from sonicdiffusionbayeslab_amd.registry import methods_registry


def setup_seed(seed):
    """``src/utils/model_utils.py:15-17``."""
    random.seed(seed)
    torch.random.manual_seed(seed)


def main(config_file):
    config = load_named_config(config_file)
    setup_seed(config.experiment.get("seed", 29))
    methods_registry[config.experiment.method](config).run_experiment()


if __name__ == "__main__":
    parser = argparse.ArgumentParser(description="Sonic Diffusion (MI355X-native hot path)")
    parser.add_argument("--config", type=str, default="config.yaml", help="Path to the config file")
    args = parser.parse_args()
    main(args.config)
