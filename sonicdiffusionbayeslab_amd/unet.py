"""``UNet2DConditionModel``-shaped host object over the libsdhip UNet handle.

Replaces ``pipe.unet`` of the reference pipeline: the call
``self.unet(latent_model_input, t, encoder_hidden_states=prompt_embeds, ...)[0]``
(``src/models.py:227-235``) works unchanged, and the sampling loop additionally uses
``forward_latents`` which fuses the CFG duplication ``torch.cat([latents]*2)``
(``src/models.py:217``) into conv_in.  All arithmetic runs in hand-written gfx950 kernels; torch
only owns the device buffers.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch

from . import _lib
from .weights import UNetConfig, param_shapes

CACHE_OFF, CACHE_FULL_AND_STORE, CACHE_SKIP = 0, 1, 2


def _c_config(cfg: UNetConfig, weight_dtype: str = "bf16", fp8_act_scales=(0.0, 0.0)) -> _lib.SdUnetConfig:
    c = _lib.SdUnetConfig()
    if weight_dtype not in _lib.DTYPES:
        raise ValueError(f"weight_dtype {weight_dtype!r}: one of {sorted(_lib.DTYPES)}")
    c.weight_dtype = _lib.DTYPES[weight_dtype]
    c.fp8_act_scale_norm, c.fp8_act_scale_ff = float(fp8_act_scales[0]), float(fp8_act_scales[1])
    c.sample_size, c.in_channels, c.out_channels = cfg.sample_size, cfg.in_channels, cfg.out_channels
    c.num_levels = len(cfg.block_out_channels)
    for i, v in enumerate(cfg.block_out_channels):
        c.block_out_channels[i] = v
        c.attn_levels[i] = int(cfg.attn_levels[i])
    c.layers_per_block = cfg.layers_per_block
    c.cross_attention_dim, c.num_heads = cfg.cross_attention_dim, cfg.num_heads
    c.norm_num_groups, c.norm_eps, c.context_len = cfg.norm_num_groups, cfg.norm_eps, cfg.context_len
    return c


def load_params(lib, handle, config: UNetConfig, state_dict: Dict[str, torch.Tensor]) -> None:
    """Every parameter ``param_shapes(config)`` names, checked for presence and shape, into the handle
    (``sd_unet_load_param`` copies; host-only, runs without a GPU).  Tensors of any float dtype are accepted --
    diffusers checkpoints are fp16 on disk."""
    for name, shape in param_shapes(config):
        if name not in state_dict:
            raise KeyError(f"state_dict lacks UNet parameter {name!r}")
        t = state_dict[name].detach().to("cpu", torch.float32).contiguous()
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"{name}: expected shape {shape}, got {tuple(t.shape)}")
        _lib.check(lib.sd_unet_load_param(handle, name.encode(), t.data_ptr(), t.numel()), f"sd_unet_load_param({name})")


class HipUNet2DConditionModel:
    """SD-1.5 UNet running on libsdhip.  ``config`` mirrors the diffusers attributes the
    reference loop reads (``in_channels``, ``sample_size``, ``time_cond_proj_dim``)."""

    def __init__(self, config: UNetConfig, state_dict: Dict[str, torch.Tensor], device: str = "cuda:0",
                 weight_dtype: str = "bf16", fp8_act_scales=(0.0, 0.0)):
        """``weight_dtype="fp8"``: OCP e4m3 weights (per-output-channel scales) and e4m3 activations with static
        per-tensor scales for the resnet 3x3 convs, proj_in, the self-attention QKV projection and the feed-forward
        GEMMs (include/sd_hip.h::sd_unet_config.weight_dtype; BASELINE configs[4])."""
        if not torch.cuda.is_available():
            raise _lib.SdHipError("HipUNet2DConditionModel needs an MI355X (no CPU fallback exists)")
        self.config = config
        self.device = torch.device(device)
        self.dtype = torch.float32          # dtype of latents / eps crossing the ABI
        self._lib = _lib.load()
        self._handle = C.c_void_p()
        torch.cuda.set_device(self.device)
        self.weight_dtype = "fp8_e4m3" if _lib.DTYPES.get(weight_dtype) == _lib.DTYPE_FP8_E4M3 else "bf16"
        ccfg = _c_config(config, weight_dtype, fp8_act_scales)
        _lib.check(self._lib.sd_unet_create(C.byref(ccfg), C.byref(self._handle)), "sd_unet_create")
        load_params(self._lib, self._handle, config, state_dict)
        _lib.check(self._lib.sd_unet_finalize(self._handle), "sd_unet_finalize")
        self._ws: Optional[torch.Tensor] = None
        self._ws_key = None
        self._ctx_key = None
        self._ctx_keepalive = None
        self.cache_branch_id = -1

    def __del__(self):
        try:
            if getattr(self, "_handle", None):
                self._lib.sd_unet_destroy(self._handle)
                self._handle = None
        except Exception:
            pass

    # -- workspace / context ---------------------------------------------------------------
    def _workspace(self, unet_batch: int) -> torch.Tensor:
        key = (unet_batch, self.cache_branch_id)
        if self._ws is None or self._ws_key != key:
            n = self._lib.sd_unet_workspace_bytes(self._handle, unet_batch, self.cache_branch_id)
            if n < 0:
                _lib.check(-1, "sd_unet_workspace_bytes")
            self._ws = None
            self._ws = torch.empty(n + 256, dtype=torch.uint8, device=self.device)
            self._ws_key = key
            self._ctx_key = None
        return self._ws

    def _ws_ptr(self, ws: torch.Tensor) -> int:
        return (ws.data_ptr() + 255) // 256 * 256

    def set_deepcache(self, cache_branch_id: int) -> None:
        """-1 disables the DeepCache plan; >= 0 reserves that branch's cached tensors."""
        if cache_branch_id != self.cache_branch_id:
            self.cache_branch_id = cache_branch_id
            self._ws_key = None

    def set_context(self, encoder_hidden_states: torch.Tensor) -> None:
        """Project K/V of the prompt for all cross-attention layers (once per sampling run)."""
        ehs = encoder_hidden_states.to(self.device, torch.float32).contiguous()
        ub = ehs.shape[0]
        if ehs.shape[1] != self.config.context_len or ehs.shape[2] != self.config.cross_attention_dim:
            raise ValueError(f"encoder_hidden_states must be [N,{self.config.context_len},"
                             f"{self.config.cross_attention_dim}], got {tuple(ehs.shape)}")
        ws = self._workspace(ub)
        _lib.check(self._lib.sd_unet_set_context(self._handle, _lib.current_stream(), ehs.data_ptr(), ub,
                                                 self.cache_branch_id, self._ws_ptr(ws), ws.numel() - 256),
                   "sd_unet_set_context")
        self._ctx_keepalive = ehs
        self._ctx_key = (encoder_hidden_states.data_ptr(), encoder_hidden_states._version, ub)

    # -- forward ---------------------------------------------------------------------------
    def forward_latents(self, latents: torch.Tensor, unet_batch: int, timestep: float,
                        out: Optional[torch.Tensor] = None, cache_mode: int = CACHE_OFF) -> torch.Tensor:
        """eps [unet_batch,4,H,W] fp32 for fp32 NCHW ``latents`` [B,4,H,W]; ``unet_batch`` is B or
        a multiple of it (CFG: 2B, the duplication is fused).  ``set_context`` must have run."""
        if self._ctx_key is None or self._ctx_key[2] != unet_batch:
            raise _lib.SdHipError("set_context(encoder_hidden_states) must be called for this batch first")
        if latents.dtype != torch.float32 or not latents.is_contiguous() or latents.device != self.device:
            latents = latents.to(self.device, torch.float32).contiguous()
        b, c, h, w = latents.shape
        if c != self.config.in_channels or h != self.config.sample_size or w != self.config.sample_size:
            raise ValueError(f"latents must be [B,{self.config.in_channels},{self.config.sample_size},"
                             f"{self.config.sample_size}], got {tuple(latents.shape)}")
        if out is None:
            out = torch.empty((unet_batch, self.config.out_channels, h, w), dtype=torch.float32, device=self.device)
        ws = self._workspace(unet_batch)
        _lib.check(self._lib.sd_unet_forward(self._handle, _lib.current_stream(), latents.data_ptr(), b, unet_batch,
                                             float(timestep), out.data_ptr(), self._ws_ptr(ws), ws.numel() - 256,
                                             cache_mode, self.cache_branch_id), "sd_unet_forward")
        return out

    # -- fp8 activation-scale calibration (include/sd_hip.h::sd_unet_calibrate_fp8) ---------------------------------
    def calibrate_fp8(self, latents: torch.Tensor, unet_batch: int, timesteps, margin: float = 2.0) -> Dict[str, float]:
        """Per-tensor e4m3 activation scales from the amax observed on ``latents`` at each of ``timesteps`` (the context of
        ``set_context`` is used; ``unet_batch`` as for ``forward_latents``).  Returns ``fp8_scales()``."""
        if self.weight_dtype != "fp8_e4m3":
            raise _lib.SdHipError("calibrate_fp8: the handle was not created with weight_dtype='fp8'")
        if self.cache_branch_id != -1:
            # sd_unet_calibrate_fp8 runs the plan WITHOUT DeepCache; a workspace sized and a context laid out for a cache
            # branch's plan would be read at another plan's offsets
            raise _lib.SdHipError("calibrate_fp8: call set_deepcache(-1) and set_context(...) first (the calibration pass runs "
                                  "the plan without DeepCache)")
        if self._ctx_key is None or self._ctx_key[2] != unet_batch:
            raise _lib.SdHipError("set_context(encoder_hidden_states) must be called for this batch first")
        latents = latents.to(self.device, torch.float32).contiguous()
        ws = self._workspace(unet_batch)
        for t in timesteps:
            _lib.check(self._lib.sd_unet_calibrate_fp8(self._handle, _lib.current_stream(), latents.data_ptr(), latents.shape[0],
                                                       unet_batch, float(t), float(margin), self._ws_ptr(ws), ws.numel() - 256),
                       "sd_unet_calibrate_fp8")
        return self.fp8_scales()

    def fp8_scales(self, with_amax: bool = False) -> Dict[str, float]:
        """{tensor name: scale} of every e4m3 activation tensor known so far (``with_amax``: (scale, observed amax))."""
        out = {}
        name = C.create_string_buffer(256)
        sc, am = C.c_float(), C.c_float()
        for i in range(self._lib.sd_unet_fp8_scale_count(self._handle)):
            _lib.check(self._lib.sd_unet_fp8_scale_info(self._handle, i, name, 256, C.byref(sc), C.byref(am)), "sd_unet_fp8_scale_info")
            out[name.value.decode()] = (sc.value, am.value) if with_amax else sc.value
        return out

    def set_fp8_scales(self, scales: Dict[str, float]) -> None:
        # the e4m3 activation tensors get their names when a plan is built: make sure one exists (batch 1, no DeepCache)
        if self._lib.sd_unet_workspace_bytes(self._handle, 1, -1) < 0:
            _lib.check(-1, "sd_unet_workspace_bytes")
        for k, v in scales.items():
            _lib.check(self._lib.sd_unet_set_fp8_scale(self._handle, k.encode(), float(v)), f"sd_unet_set_fp8_scale({k})")

    KIND_NAMES = {0: "sinusoid", 1: "gemv", 2: "conv_in", 3: "groupnorm", 4: "conv3x3", 5: "gemm", 6: "layernorm",
                  7: "attention", 8: "conv_out", 16: "conv3x3_fp8", 17: "gemm_fp8", 18: "xattn_fused",
                  19: "replicate", 20: "conv3x3_gemm", 21: "conv3x3_halo_subpix"}

    def forward_profiled(self, latents: torch.Tensor, unet_batch: int, timestep: float, cache_mode: int = CACHE_OFF):
        """One forward with a hipEvent pair around every launch (measurement only, synchronises).
        Returns {kind: dict(ms, launches, flops, bytes)}."""
        latents = latents.to(self.device, torch.float32).contiguous()
        out = torch.empty((unet_batch, self.config.out_channels, latents.shape[2], latents.shape[3]),
                          dtype=torch.float32, device=self.device)
        ws = self._workspace(unet_batch)
        ms, fl, by = (C.c_double * 32)(), (C.c_double * 32)(), (C.c_double * 32)()
        ln = (C.c_longlong * 32)()
        _lib.check(self._lib.sd_unet_forward_profiled(
            self._handle, _lib.current_stream(), latents.data_ptr(), latents.shape[0], unet_batch, float(timestep),
            out.data_ptr(), self._ws_ptr(ws), ws.numel() - 256, cache_mode, self.cache_branch_id, ms, ln, fl, by),
            "sd_unet_forward_profiled")
        return {n: dict(ms=ms[i], launches=ln[i], flops=fl[i], bytes=by[i]) for i, n in self.KIND_NAMES.items()
                if ln[i] or i < 9}

    def __call__(self, sample: torch.Tensor, timestep, encoder_hidden_states: torch.Tensor = None,
                 timestep_cond=None, cross_attention_kwargs=None, added_cond_kwargs=None, return_dict: bool = False,
                 **kwargs):
        """diffusers-style call of the reference loop (``src/models.py:227-235``)."""
        if timestep_cond is not None or added_cond_kwargs is not None:
            raise NotImplementedError("timestep_cond / added_cond_kwargs are not part of the SD-1.5 hot path")
        key = (encoder_hidden_states.data_ptr(), encoder_hidden_states._version, sample.shape[0])
        if self._ctx_key != key:
            self.set_context(encoder_hidden_states)
        t = float(timestep.item()) if torch.is_tensor(timestep) else float(timestep)
        eps = self.forward_latents(sample, sample.shape[0], t)
        return (eps.to(sample.dtype),)

    def debug_tensor(self, name: str, unet_batch: int, numel: int) -> torch.Tensor:
        out = torch.empty(numel, dtype=torch.float32)
        ws = self._workspace(unet_batch)
        _lib.check(self._lib.sd_unet_debug_tensor(self._handle, _lib.current_stream(), name.encode(), out.data_ptr(),
                                                  numel, self._ws_ptr(ws), unet_batch, self.cache_branch_id),
                   "sd_unet_debug_tensor")
        return out
