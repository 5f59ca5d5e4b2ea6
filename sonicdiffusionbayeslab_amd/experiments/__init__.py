from .base_experiment import BaseMethod  # noqa: F401
from .consistency_model import ConsistencyModelMethod  # noqa: F401
from .ddim import DDIMMethod  # noqa: F401
from .default_sd import DefaultStableDiffusion  # noqa: F401
from .deep_cache import DeepCacheMethod  # noqa: F401
from .dpm_solver import DPMSolverMethod  # noqa: F401
