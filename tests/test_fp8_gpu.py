"""GPU parity of the fp8-e4m3 operand path (include/sd_hip.h::SD_DTYPE_FP8_E4M3; BASELINE configs[4]: LCM 4 steps,
"fp8 MFMA weights"; reference side configs/consistency_model_config.yaml:1-34, src/experiments/consistency_model.py:9-52
-- the reference runs fp16, the fp8 scheme is this build's).

Operator level (through the C ABI): operands are rounded to the e4m3 grid on the CPU (torch.float8_e4m3fn, the same
OCP format), so the comparison isolates kernel arithmetic -- the products of e4m3 values are exact in fp32, only the
accumulation order and the bf16 output rounding differ: tolerance rel-L2 <= 6e-3 like the bf16 operator tests.
Producers (GroupNorm / LayerNorm / GEGLU writing e4m3): a value that lands within fp32 noise of a rounding boundary may
round the other way, i.e. differ by one e4m3 step (6 % of its magnitude): tolerance 2e-2 on the dequantised tensor.

Model level: the HIP forward / LCM loop vs the fp32 oracle with the SAME rounding points (oracle/fp8.py) AND vs the
unquantised fp32 oracle.  e4m3 keeps 3 mantissa bits: every fp8 contraction adds ~5 % relative noise to its output and
the emulating oracle itself sits ~1e-1 (rel-L2, one forward of the seeded synthetic UNet) from the unquantised one.
Upstream bf16-vs-fp32 differences (4e-3) flip the e4m3 rounding decision of a few per cent of the activations at every
quantisation point (each flip = one 6-12 % step on that element) and the flips compound over ~60 quantisation
points, so two correct implementations of the same scheme agree only to about the scheme's own error (measured:
8.7e-2 between HIP and the emulation, noise correlation 0.64).  The test therefore asserts what is meaningful:
  * the HIP forward is no further from the UNQUANTISED oracle than 1.25 x the emulated scheme is (the kernels add no
    error beyond the number format), cosine >= 0.99;
  * HIP vs the emulation <= 1.5e-1 (sanity bound of the same order as the scheme's error);
  * free-running 4-step LCM latents vs the emulation <= 1.2e-1, cosine >= 0.99.
Bit-level agreement of the quantisation itself is pinned elsewhere: operator tests above (kernels), 
tests/test_host_cpu.py::test_finalize_packs_weights_on_the_host (weight codes and scales, bit for bit)."""
import dataclasses
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from sonicdiffusionbayeslab_amd import _lib
from tests.util import cosine, oracle_cfg, rel_l2, synth_inputs

TOL = 6e-3
PROD_TOL = 2e-2
FWD_TOL = 1.5e-1
FWD_EXCESS = 1.25        # HIP-vs-unquantised error allowed relative to the emulated scheme's own error
LOOP_TOL = 1.2e-1

_KEEP = []


def stream():
    return torch.cuda.current_stream().cuda_stream


def P(t):
    if t is None:
        return None
    if t.device.type != "cuda":
        t = t.cuda()
    _KEEP.append(t)
    return t.data_ptr()


@pytest.fixture(autouse=True)
def _drop_keep():
    yield
    torch.cuda.synchronize()
    _KEEP.clear()


def pad128(c):
    return (c + 127) // 128 * 128


def q8(x):
    """fp32 -> (e4m3 codes as uint8, dequantised fp32): nearest-even, saturating (OCP e4m3fn)."""
    q = x.float().clamp(-448, 448).to(torch.float8_e4m3fn)
    return q.view(torch.uint8), q.float()


def quant_w(w):
    """per-output-channel weight quantisation -> (codes uint8 [N, ...], dequantised, scale [N])"""
    from oracle.fp8 import quantize_rows
    q, scale = quantize_rows(w)
    codes = q.clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
    return codes, q, scale


def padk(codes, Kp):
    out = torch.zeros(codes.shape[0], Kp, dtype=torch.uint8)
    out[:, : codes.shape[1]] = codes
    return out


@pytest.mark.parametrize("M,N,K,bias,res", [
    (256, 320, 320, True, False),       # K = 320 pads to 384
    (300, 320, 640, True, True),        # M tail
    (130, 640, 1280, False, True),
    (64, 1280, 5120, True, True),       # ff.net.2 at the 8x8 level: split-K
    (1000, 960, 320, False, False),     # fused QKV shape
])
def test_gemm_fp8(sdlib, M, N, K, bias, res):
    g = torch.Generator().manual_seed(M + N + K)
    xs = 8.0
    xc, xq = q8(torch.randn(M, K, generator=g) * xs)            # activations as their producer would write them
    wc, wq, wsc = quant_w(torch.randn(N, K, generator=g) / math.sqrt(K))
    b = torch.randn(N, generator=g) if bias else None
    r = torch.randn(M, N, generator=g).bfloat16().float() if res else None
    ref = (xq / xs) @ (wq * wsc[:, None]).t()
    if bias: ref = ref + b
    if res: ref = ref + r
    Kp = pad128(K)
    out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
    _lib.check(sdlib.sd_op_gemm_fp8(stream(), P(padk(xc, Kp)), Kp, P(padk(wc, Kp)), P(wsc), xs, P(b),
                                    P(r.bfloat16()) if res else None, N, P(out), N, M, N, Kp, 0, 0, 1.0))
    torch.cuda.synchronize()
    assert rel_l2(out, ref) < TOL


@pytest.mark.parametrize("M,C,out_fp8", [(200, 320, 0), (200, 320, 1), (512, 640, 1), (96, 1280, 0)])
def test_gemm_geglu_fp8(sdlib, M, C, out_fp8):
    g = torch.Generator().manual_seed(3 + C)
    xs, os_ = 8.0, 2.0
    xc, xq = q8(torch.randn(M, C, generator=g) * xs)
    w = torch.randn(8 * C, C, generator=g) / math.sqrt(C)
    b = torch.randn(8 * C, generator=g)
    wc, wq, wsc = quant_w(w)
    proj = (xq / xs) @ (wq * wsc[:, None]).t() + b
    a, gate = proj.chunk(2, dim=-1)
    ref = a * F.gelu(gate)
    H = 4 * C
    idx = []
    for r in range(2 * H):
        grp, within = divmod(r, 32)
        idx.append(grp * 16 + within if within < 16 else H + grp * 16 + within - 16)
    idx = torch.tensor(idx)
    Kp = pad128(C)
    wp, sp, bp = padk(wc[idx].contiguous(), Kp), wsc[idx].contiguous(), b[idx].contiguous()
    if out_fp8:
        out = torch.full((M, H), 0x7f, device="cuda", dtype=torch.uint8)
        _lib.check(sdlib.sd_op_gemm_fp8(stream(), P(padk(xc, Kp)), Kp, P(wp), P(sp), xs, P(bp), None, 0, P(out), H, M,
                                        8 * C, Kp, 1, 1, os_))
        torch.cuda.synchronize()
        got = out.cpu().view(torch.float8_e4m3fn).float() / os_
        assert torch.isfinite(got).all()
        assert rel_l2(got, q8(ref * os_)[1] / os_) < PROD_TOL
        assert rel_l2(got, ref) < 4e-2                  # e4m3 itself: 2^-4 relative steps
    else:
        out = torch.full((M, H), float("nan"), device="cuda", dtype=torch.bfloat16)
        _lib.check(sdlib.sd_op_gemm_fp8(stream(), P(padk(xc, Kp)), Kp, P(wp), P(sp), xs, P(bp), None, 0, P(out), H, M,
                                        8 * C, Kp, 1, 0, 1.0))
        torch.cuda.synchronize()
        assert rel_l2(out, ref) < TOL


@pytest.mark.parametrize("B,H,Cin,Cout,stride,up,extras", [
    (2, 16, 320, 320, 1, 0, True),      # halo kernel, Cin 320 -> 384 (a half-empty last slice)
    (1, 16, 640, 320, 1, 0, False),
    (1, 64, 128, 320, 1, 0, True),      # halo kernel: 4 rows of 64 pixels per tile, one slice
    (2, 32, 960, 640, 1, 0, True),      # up-path concat width, 7.5 slices
    (1, 16, 2560, 1280, 1, 0, True),    # split-K over 128-channel slices
    (5, 8, 1280, 192, 1, 0, True),      # 4 whole 8x8 images per tile + an M tail tile, Cout tail
    (3, 4, 256, 320, 1, 0, True),       # 4x4 images: implicit-GEMM kernel (halo kernel needs width >= 8)
    (2, 16, 256, 320, 2, 0, False),     # stride 2: implicit-GEMM kernel
    (2, 16, 128, 320, 1, 1, True),      # halo kernel with the fused nearest-2x upsample
])
def test_conv3x3_fp8(sdlib, B, H, Cin, Cout, stride, up, extras):
    g = torch.Generator().manual_seed(B * 100 + H + Cin)
    xs = 8.0
    xc, xq = q8(torch.randn(B, Cin, H, H, generator=g) * xs)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    wc, wq, wsc = quant_w(w)
    b = torch.randn(Cout, generator=g)
    xin = xq / xs
    if up:
        xin = F.interpolate(xin, scale_factor=2.0, mode="nearest")
    ref = F.conv2d(xin, wq * wsc[:, None, None, None], b, stride=stride, padding=1)
    Ho = ref.shape[-1]
    b2 = r = None
    if extras:
        b2 = torch.randn(Cout, generator=g)
        r = torch.randn(B, Cout, Ho, Ho, generator=g).bfloat16().float()
        ref = ref + b2[None, :, None, None] + r
    Cp = pad128(Cin)
    xd = torch.zeros(B, H, H, Cp, dtype=torch.uint8)
    xd[..., :Cin] = xc.permute(0, 2, 3, 1)
    wpad = torch.zeros(Cout, Cp, 3, 3, dtype=torch.uint8)
    wpad[:, :Cin] = wc
    wd = wpad.permute(0, 2, 3, 1).reshape(Cout, 9, Cp // 128, 128).permute(0, 2, 1, 3).contiguous()
    rd = r.permute(0, 2, 3, 1).contiguous().bfloat16() if extras else None
    out = torch.full((B, Ho, Ho, Cout), float("nan"), device="cuda", dtype=torch.bfloat16)
    _lib.check(sdlib.sd_op_conv3x3_fp8(stream(), P(xd), P(wd), P(wsc), xs, P(b), P(b2) if extras else None, P(rd),
                                       P(out), B, H, H, Cp, Cout, stride, up))
    torch.cuda.synchronize()
    assert rel_l2(out.permute(0, 3, 1, 2), ref) < TOL


@pytest.mark.parametrize("B,HW,C1,C2,silu", [(2, 256, 320, 0, 1), (2, 64, 640, 320, 1), (1, 1024, 1280, 0, 0),
                                               (3, 16, 1280, 640, 1), (2, 64, 320, 0, 1)])
def test_groupnorm_fp8_out(sdlib, B, HW, C1, C2, silu):
    g = torch.Generator().manual_seed(HW + C1)
    C, Cp, s = C1 + C2, pad128(C1 + C2), 8.0
    x = (torch.randn(B, HW, C, generator=g) * 2 + 0.5).bfloat16().float()
    gm, bt = torch.randn(C, generator=g) * 0.3 + 1, torch.randn(C, generator=g) * 0.2
    ref = F.group_norm(x.permute(0, 2, 1), 32, gm, bt, 1e-5)
    if silu:
        ref = F.silu(ref)
    ref = ref.permute(0, 2, 1)
    x1 = x[..., :C1].contiguous().bfloat16()
    x2 = x[..., C1:].contiguous().bfloat16() if C2 else None
    out = torch.full((B, HW, Cp), 0x7f, device="cuda", dtype=torch.uint8)
    _lib.check(sdlib.sd_op_groupnorm_fp8(stream(), P(x1), C1, P(x2), C2, P(gm), P(bt), P(out), B, HW, 32, 1e-5, silu, Cp, s))
    torch.cuda.synchronize()
    o = out.cpu()
    assert (o[..., C:] == 0).all(), "K-tail padding must be zero"
    got = o[..., :C].contiguous().view(torch.float8_e4m3fn).float() / s
    assert rel_l2(got, q8(ref * s)[1] / s) < PROD_TOL


@pytest.mark.parametrize("rows,C", [(300, 320), (129, 640), (64, 1280), (50, 768)])
def test_layernorm_fp8_out_and_quantize(sdlib, rows, C):
    g = torch.Generator().manual_seed(rows + C)
    Cp, s = pad128(C), 8.0
    x = (torch.randn(rows, C, generator=g) * 1.5 + 0.2).bfloat16().float()
    gm, bt = torch.randn(C, generator=g) * 0.3 + 1, torch.randn(C, generator=g) * 0.2
    ref = F.layer_norm(x, (C,), gm, bt, 1e-5)
    out = torch.full((rows, Cp), 0x7f, device="cuda", dtype=torch.uint8)
    _lib.check(sdlib.sd_op_layernorm_fp8(stream(), P(x.bfloat16()), P(gm), P(bt), P(out), rows, C, Cp, 1e-5, s))
    torch.cuda.synchronize()
    o = out.cpu()
    assert (o[:, C:] == 0).all()
    assert rel_l2(o[:, :C].contiguous().view(torch.float8_e4m3fn).float() / s, q8(ref * s)[1] / s) < PROD_TOL
    # plain conversion: bit-exact against torch's e4m3 rounding (same format, same nearest-even rule, saturation)
    big = x * 100.0                                   # exercises the +-448 saturation
    out2 = torch.full((rows, Cp), 0x7f, device="cuda", dtype=torch.uint8)
    _lib.check(sdlib.sd_op_quantize_fp8(stream(), P(big.bfloat16()), P(out2), rows, C, Cp, 1.0))
    torch.cuda.synchronize()
    want = q8(big.bfloat16().float())[0]
    got = out2.cpu()[:, :C]
    same = (got == want) | ((got & 0x7f) == 0) & ((want & 0x7f) == 0)      # +0 / -0
    assert same.all() and (out2.cpu()[:, C:] == 0).all()


# ---------------------------------------------------------------------------------------------------------------
# model level
# ---------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def small_fp8():
    from sonicdiffusionbayeslab_amd.unet import HipUNet2DConditionModel
    from sonicdiffusionbayeslab_amd.weights import UNetConfig, make_synthetic_state_dict
    cfg = UNetConfig(sample_size=16)
    sd = make_synthetic_state_dict(cfg, seed=1234)
    net = HipUNet2DConditionModel(cfg, sd, weight_dtype="fp8")
    return cfg, sd, net


@pytest.mark.parametrize("t", [981.0, 21.0])
def test_unet_forward_fp8_matches_emulating_oracle(small_fp8, t):
    from oracle.fp8 import Fp8Emulation
    from oracle.unet import unet_forward
    cfg, sd, net = small_fp8
    assert net.weight_dtype == "fp8_e4m3"
    lat, pe, ne = synth_inputs(cfg, 1)
    ctx = torch.cat([ne, pe])
    with torch.no_grad():
        ref_q = unet_forward(sd, oracle_cfg(cfg), torch.cat([lat, lat]), t, ctx, fq=Fp8Emulation(sd))
        ref = unet_forward(sd, oracle_cfg(cfg), torch.cat([lat, lat]), t, ctx)
    net.set_context(ctx.cuda())
    eps = net.forward_latents(lat.cuda(), 2, t)
    torch.cuda.synchronize()
    e_q, e_f = rel_l2(eps, ref_q), rel_l2(eps, ref)
    print(f"fp8 forward t={t}: vs emulating oracle {e_q:.3e} (cos {cosine(eps, ref_q):.5f}); vs unquantised oracle "
          f"{e_f:.3e}; oracle fp8-vs-fp32 {rel_l2(ref_q, ref):.3e}")
    assert torch.isfinite(eps).all()
    assert e_q < FWD_TOL and cosine(eps, ref) > 0.99
    assert e_f < FWD_EXCESS * rel_l2(ref_q, ref) + 1e-2


def test_lcm_loop_fp8(small_fp8):
    """BASELINE configs[4] at reduced size: LCM sampling without CFG, pre-drawn noise, fp8 weights, through the pipeline with its
    own calibration.  Two steps here (the emulating oracle costs ~15 s per step on the GPU box's host; round 4: 114 s of the
    suite's 600): the FULL 4-step loop at 64x64 runs from the committed fixture in seconds
    (tests/test_benchshapes_gpu.py::test_full_length_loops_at_64x64_against_the_oracle_fixtures[lcm4_fp8])."""
    from oracle.fp8 import Fp8Emulation
    from oracle.pipeline import sample_loop
    from oracle.schedulers import LCMOracle
    from sonicdiffusionbayeslab_amd.models import StableDiffusionModel
    from sonicdiffusionbayeslab_amd.registry import schedulers_registry
    from sonicdiffusionbayeslab_amd.schedulers import PNDMConfigStub
    cfg, sd, _ = small_fp8
    model = StableDiffusionModel(unet_config=cfg, state_dict=dict(sd), weight_dtype="fp8").to("cuda:0")
    model.scheduler = schedulers_registry["lcm_scheduler"].from_config(PNDMConfigStub().config)
    lat, pe, _ = synth_inputs(cfg, 2, seed=17)
    g = torch.Generator().manual_seed(8)
    noise = torch.randn(1, 2, 4, 16, 16, generator=g)
    out, secs, _ = model(prompt_embeds=pe, latents=lat, num_inference_steps=2, guidance_scale=0.0,
                         output_type="latent", step_noise=noise.cuda())
    assert "fp8_e4m3" in model.weights_source
    scales = model.unet.fp8_scales(with_amax=True)       # the pipeline calibrated them on its fixed seeded batch before the loop
    assert scales and all(a > 0 for _, a in scales.values()) and "calibrated" in model.weights_source
    ref_q, _, _, _ = sample_loop(sd, oracle_cfg(cfg), LCMOracle(), pe, None, lat, 2, 0.0, lcm_noise=noise,
                                 fq=Fp8Emulation(sd, scales={k: v[0] for k, v in scales.items()}))
    ref, _, _, _ = sample_loop(sd, oracle_cfg(cfg), LCMOracle(), pe, None, lat, 2, 0.0, lcm_noise=noise)
    e_q, e_f = rel_l2(out.images, ref_q), rel_l2(out.images, ref)
    print(f"LCM 2 steps fp8: vs emulating oracle {e_q:.3e} cos {cosine(out.images, ref_q):.5f}; vs unquantised oracle "
          f"{e_f:.3e}; loop {secs * 1e3:.1f} ms")
    assert e_q < LOOP_TOL and cosine(out.images, ref_q) > 0.99


def test_fp8_calibration_positions_the_range(small_fp8):
    """``sd_unet_calibrate_fp8`` (include/sd_hip.h): with the static default scale 8 an e4m3 norm output clips at
    |x| > 56.  A checkpoint whose GroupNorm / LayerNorm gains are 30x larger than the synthetic ones (real SD-1.5 has such
    layers) drives the norm outputs far beyond that: the default-scale forward saturates and lands far from the
    unquantised oracle; after calibration on the same inputs every tensor's scale * amax stays below 448 and the forward is
    back at the number format's own error.  The emulating oracle is run with the calibrated per-tensor scales."""
    from oracle.fp8 import Fp8Emulation
    from oracle.unet import unet_forward
    from sonicdiffusionbayeslab_amd.unet import HipUNet2DConditionModel
    cfg, sd0, _ = small_fp8
    sd = dict(sd0)
    hot = [k for k in sd if k.endswith(("resnets.0.norm2.weight", "transformer_blocks.0.norm3.weight")) and "down_blocks.1" in k]
    assert len(hot) >= 2
    for k in hot:
        sd[k] = (sd[k] * 30.0).to(torch.bfloat16).float()
    net = HipUNet2DConditionModel(cfg, sd, weight_dtype="fp8")
    lat, pe, ne = synth_inputs(cfg, 1)
    ctx = torch.cat([ne, pe])
    t = 499.0
    with torch.no_grad():
        ref = unet_forward(sd, oracle_cfg(cfg), torch.cat([lat, lat]), t, ctx)
    net.set_context(ctx.cuda())
    before = net.forward_latents(lat.cuda(), 2, t).clone()
    defaults = net.fp8_scales(with_amax=True)
    assert all(a == 0.0 for _, a in defaults.values()) and set(s for s, _ in defaults.values()) == {8.0, 2.0}
    scales = net.calibrate_fp8(lat.cuda(), 2, [t], margin=2.0)
    full = net.fp8_scales(with_amax=True)
    assert set(scales) == set(defaults) and len(scales) >= 60
    for name, (s, amax) in full.items():
        assert amax > 0 and s * amax <= 448.0 / 2.0 * 1.0001 and s * amax > 448.0 / 8.0, (name, s, amax)   # margin 2, power-of-two floor
        assert s == 2.0 ** round(math.log2(s))
    hot_names = [k[: -len(".weight")] for k in hot]
    assert all(full[n][1] > 56.0 and full[n][0] < 8.0 for n in hot_names), [full[n] for n in hot_names]
    net.set_context(ctx.cuda())
    after = net.forward_latents(lat.cuda(), 2, t).clone()
    with torch.no_grad():
        ref_q = unet_forward(sd, oracle_cfg(cfg), torch.cat([lat, lat]), t, ctx, fq=Fp8Emulation(sd, scales=scales))
    e_before, e_after, scheme = rel_l2(before, ref), rel_l2(after, ref), rel_l2(ref_q, ref)
    print(f"fp8 calibration: vs unquantised oracle {e_before:.3e} with the default scales (hot layers saturate) -> {e_after:.3e} "
          f"calibrated; emulated calibrated scheme {scheme:.3e}; HIP vs emulation {rel_l2(after, ref_q):.3e}")
    assert e_after < 0.5 * e_before
    # HIP vs the emulating oracle under 30x gains: both quantise the same way, but a last-bit difference in a GroupNorm variance
    # flips e4m3 ties in the hot layers and the comparison moves by percent: 1.462e-1 / 1.482e-1 / 1.513e-1 on three builds of
    # round 5 that differ only in how hipcc contracts the variance sum (packed / scalar fp32, helper inlining) -- the scheme's own
    # distance from the unquantised oracle is 1.63e-1.  Gate at that distance, not at the ordinary forward's FWD_TOL.
    assert e_after < FWD_EXCESS * scheme + 1e-2 and rel_l2(after, ref_q) < max(FWD_TOL, scheme)
    # a saved calibration restores bit-identical behaviour on a fresh handle
    net2 = HipUNet2DConditionModel(cfg, sd, weight_dtype="fp8")
    net2.set_context(ctx.cuda())
    net2.forward_latents(lat.cuda(), 2, t)                                  # builds the plan: the tensor names exist
    net2.set_fp8_scales(scales)
    net2.set_context(ctx.cuda())
    assert torch.equal(net2.forward_latents(lat.cuda(), 2, t), after)
