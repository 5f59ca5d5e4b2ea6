"""``DeepCacheSDHelper`` with the call pattern of the reference
(``src/experiments/deep_cache.py:3,24-29,58``):

    helper = DeepCacheSDHelper(pipe=model)
    helper.set_params(cache_interval=N, cache_branch_id=0)
    helper.enable()   ...   helper.disable()

DeepCache 0.1.1 monkey-patches every UNet sub-module's ``forward`` and decides per call whether
to return a cached output.  Here the same rule is evaluated ONCE, when the
execution plan is built inside libsdhip: ops of skipped modules are dropped from the skip-step
plan and the tensors they leave behind for still-running ops are pinned in the workspace.  For
the reference's ``cache_branch_id: 0`` a skip step therefore runs only time-embedding, conv_in,
``up_blocks[3].resnets[2]`` (on ``cat[cached up3.attn[1] output, fresh conv_in output]``),
``up_blocks[3].attentions[2]``, ``conv_norm_out`` and ``conv_out``.
"""
from __future__ import annotations


class DeepCacheSDHelper:
    def __init__(self, pipe=None):
        self.pipe = pipe
        self.cache_interval = 1
        self.cache_branch_id = 0
        self.skip_mode = "uniform"

    def set_params(self, cache_interval: int = 1, cache_branch_id: int = 0, skip_mode: str = "uniform"):
        if skip_mode != "uniform":
            raise NotImplementedError("only skip_mode='uniform' is used by the reference")
        if cache_interval < 1:
            raise ValueError("cache_interval must be >= 1")
        if cache_branch_id < 0:
            raise ValueError("cache_branch_id must be >= 0")
        self.cache_interval = int(cache_interval)
        self.cache_branch_id = int(cache_branch_id)
        self.skip_mode = skip_mode

    def enable(self, pipe=None):
        if pipe is not None:
            self.pipe = pipe
        if self.pipe is None:
            raise ValueError("DeepCacheSDHelper needs a pipeline")
        self.pipe._deepcache = self

    def disable(self):
        if self.pipe is not None and getattr(self.pipe, "_deepcache", None) is self:
            self.pipe._deepcache = None
