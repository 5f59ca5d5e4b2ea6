"""fp8-e4m3 emulation for the CPU oracle: the rounding points of the product's ``SD_DTYPE_FP8_E4M3`` plan
(include/sd_hip.h::sd_unet_config.weight_dtype; BASELINE configs[4] "fp8 MFMA weights"), applied to the fp32
restatement so that the HIP path can be compared against the reference arithmetic ON THE SAME ROUNDED OPERANDS.

TEST INFRASTRUCTURE (see oracle/__init__.py) -- parity unpinned; the reference itself has no fp8 path
(``configs/consistency_model_config.yaml:1-34``, ``src/experiments/consistency_model.py:9-52`` run fp16): what is
emulated here is this build's own quantisation scheme, so that its error is separated from kernel error.

Scheme (identical in csrc/unet.hip):
  * weights of the resnet 3x3 convs, proj_in, attn1 to_q/to_k/to_v, ff.net.0.proj and ff.net.2: OCP e4m3fn, one
    fp32 scale per OUTPUT channel (row): scale = amax / 448, w_q = rne(w / scale), saturating;
  * the activations entering those contractions -- GroupNorm(+SiLU) and LayerNorm outputs (scale ``s_norm``) and the
    GEGLU product (scale ``s_ff``) -- e4m3 of ``clamp(x * s, +-448)`` with a static per-tensor scale (the defaults, or
    the per-tensor scales of the product's calibration, ``scales``);
  * everything else (attention, cross-attention, to_out, proj_out, shortcuts, down/upsampler convs, conv_in/out)
    is not quantised.
"""
from __future__ import annotations

from typing import Dict

import torch

E4M3_MAX = 448.0


def e4m3_round(x: torch.Tensor) -> torch.Tensor:
    """Nearest-even rounding to the OCP e4m3fn grid with saturation at +-448 (fp32 in, fp32 out)."""
    return x.float().clamp(-E4M3_MAX, E4M3_MAX).to(torch.float8_e4m3fn).to(torch.float32)


def quantize_rows(w: torch.Tensor):
    """Per-output-channel (dim 0) scale and e4m3 codes: w ~= q * scale[:, None...]."""
    flat = w.float().reshape(w.shape[0], -1)
    amax = flat.abs().amax(dim=1)
    scale = torch.where(amax > 0, amax / E4M3_MAX, torch.ones_like(amax))
    q = e4m3_round(flat * (1.0 / scale)[:, None])     # the host packer multiplies by the reciprocal, as here
    return q.reshape(w.shape), scale


QUANTISED_SUFFIXES = (".conv1.weight", ".conv2.weight", ".proj_in.weight", ".attn1.to_q.weight", ".attn1.to_k.weight",
                      ".attn1.to_v.weight", ".ff.net.0.proj.weight", ".ff.net.2.weight")


class Fp8Emulation:
    def __init__(self, weights: Dict[str, torch.Tensor], s_norm: float = 8.0, s_ff: float = 2.0, scales=None):
        """``scales``: {tensor name: scale} as ``HipUNet2DConditionModel.fp8_scales()`` returns them (the product's
        calibrated per-tensor scales; names = the module that writes the tensor: "<resnet>.norm1", "<attn>.norm",
        "<block>.norm1|norm3", "<block>.ff.net.0"); tensors without an entry use the defaults ``s_norm`` / ``s_ff``."""
        self.s_norm, self.s_ff = float(s_norm), float(s_ff)
        self.scales = dict(scales or {})
        self.wq: Dict[str, torch.Tensor] = {}
        for name, w in weights.items():
            if name.endswith(QUANTISED_SUFFIXES) and ("resnets." in name or "attentions." in name):
                q, scale = quantize_rows(w)
                self.wq[name] = q * scale.reshape((-1,) + (1,) * (w.dim() - 1))

    def w(self, weights, name):
        return self.wq.get(name, weights[name])

    def act_norm(self, x, name=None):
        s = float(self.scales.get(name, self.s_norm))
        return e4m3_round(x * s) / s

    def act_ff(self, x, name=None):
        s = float(self.scales.get(name, self.s_ff))
        return e4m3_round(x * s) / s


class Fp8AmaxRecorder(Fp8Emulation):
    """Oracle-side calibration (tests/golden/make_loop_golden.py): a forward with e4m3 WEIGHTS whose activation rounding
    points only RECORD the largest |value| of the tensor they would quantise (the activations pass unrounded, as the
    product's calibration pass sees them under its non-saturating probe scale).  ``calibrated_scales(margin)`` applies the product's
    rule (csrc/unet.hip::sd_unet_calibrate_fp8): the largest power of two <= 448 / (margin x amax)."""

    def __init__(self, weights):
        super().__init__(weights)
        self.amax: Dict[str, float] = {}

    def _see(self, x, name):
        self.amax[name] = max(self.amax.get(name, 0.0), float(x.abs().max()))
        return x

    def act_norm(self, x, name=None):
        return self._see(x, name)

    def act_ff(self, x, name=None):
        return self._see(x, name)

    def calibrated_scales(self, margin: float = 2.0) -> Dict[str, float]:
        import math
        return {k: 2.0 ** math.floor(math.log2(E4M3_MAX / (margin * a))) for k, a in self.amax.items() if a > 0}
