"""Parity at the shapes ``bench.py`` really runs (BASELINE configs[1..4]), which the batch-2 tests of
tests/test_fullsize_gpu.py do not reach: the 64- vs 128-row GEMM tiles, the split-K factors, the ``rep = 2`` CFG plan and
the arena all depend on M = UNet batch x HW (csrc/unet.hip, csrc/gemm_conv.hip heuristics).

Tolerances (stated here, asserted below; "oracle" = oracle/ = this build's fp32 CPU restatement, PARITY UNPINNED --
DESIGN.md 2):
  (a) batch consistency at 64x64, bf16: a sample evaluated inside UNet batch 16 (CFG plan, rep = 2: configs[1]),
      64 (configs[2]: DPM-Solver++ batch 32 with CFG) and 32 without CFG (configs[4] per-GPU share) vs the same sample in a
      UNet-batch-2 forward.  At 64x64 the batch size changes the accumulation order of nearly EVERY contraction (UNet batch 2
      runs all convs and most GEMMs with split-K, batch >= 16 none at the upper levels; 64- vs 128-row tiles), so the fp32
      sums differ in their last bits ahead of every bf16 rounding of the ~130 activation tensors: the two runs are two
      independent draws of the bf16 rounding noise and differ by about as much as either differs from the fp32 oracle
      (measured 1.1e-2 against 1.0e-2; at 16x16, where both batch sizes pick the same kernels, 2e-3 holds:
      tests/test_unet_gpu.py).  Asserted: (i) per sample rel-L2 <= BATCH_TOL = 2e-2 and cosine >= 0.9995 -- a wrong tile,
      split or arena overlap shows as O(1), not as noise; (ii) for the probe sample whose oracle forward is computed (one
      CFG pair, ~5 s), the error of every batch size vs the ORACLE stays <= UNET_TOL = 2e-2 and within 1.5 x the batch-2
      forward's own error: a larger batch adds no error of its own.
  (b) fp8-e4m3 at 64x64 (configs[4]): one UNet-batch-2 forward vs oracle.fp8.Fp8Emulation with the assertions of
      tests/test_fp8_gpu.py::test_unet_forward_fp8_matches_emulating_oracle (FWD_TOL 1.5e-1 vs the emulation, no further
      from the unquantised oracle than 1.25 x the emulated scheme is, cosine >= 0.99), plus batch 32 vs batch 2:
      rel-L2 <= FP8_BATCH_TOL = 8e-2 (measured 4.3e-2 .. 5.0e-2) -- a bf16-level difference upstream flips e4m3 rounding decisions downstream (one
      flip = a 6-12 % step on that element), so fp8 batch consistency is of the order of the scheme's own noise, not 2e-3.
  (c) free-running drift (src/models.py:210-282): 50 DDIM steps with CFG 7.5 at 16x16 and the first 10 of 50 at 64x64,
      batch 1, HIP loop vs the oracle loop from the same latents; max-abs / rel-L2 / cosine of the final latents are
      printed, asserted: rel-L2 <= DRIFT_TOL_50 = 1.5e-1 and cosine >= 0.99 after 50 steps (per-step error ~1e-2 with CFG
      7.5, compounding over the trajectory), rel-L2 <= DRIFT_TOL_10 = 6e-2 and cosine >= 0.995 after 10 steps at 64x64.
      The full 50 of 50 steps at 64x64 are an opt-in test (SD_LONG_PARITY=1, ~5 min of oracle time): last run rel-L2
      1.02e-2 / 1.03e-2 / 1.03e-2 / 1.03e-2 / 1.03e-2 after 10 / 20 / 30 / 40 / 50 steps, cosine 0.99995 throughout.
      Same switch: the full loops of configs[2..4] at 64x64 through StableDiffusionModel.__call__ (DPM-Solver++ 20 steps
      1.6e-2, LCM 4 steps 5.4e-3, DeepCache N = 3 over 50 steps 1.7e-2; asserted <= DRIFT_TOL_50, cosine >= 0.99).
  (d) VAE decoder 64 -> 512 (src/models.py:287-302), batch 1, vs oracle/vae.py: rel-L2 <= 2e-2, cosine >= 0.999 -- the
      128^2 .. 512^2 levels run the implicit-GEMM conv kernel the 16 -> 128 test never reaches.
Reference call sites: src/models.py:210-282,287-302; configs/consistency_model_config.yaml:1-34."""
import dataclasses
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.util import cosine, oracle_cfg, rel_l2, synth_inputs

BATCH_TOL = 2e-2
UNET_TOL = 2e-2
FP8_BATCH_TOL = 8e-2
FWD_TOL, FWD_EXCESS = 1.5e-1, 1.25
DRIFT_TOL_50, DRIFT_TOL_10 = 1.5e-1, 6e-2


def _net(sample_size, weight_dtype="bf16"):
    from sonicdiffusionbayeslab_amd.unet import HipUNet2DConditionModel
    from sonicdiffusionbayeslab_amd.weights import UNetConfig, make_synthetic_state_dict
    cfg = UNetConfig(sample_size=sample_size)
    sd = make_synthetic_state_dict(cfg, seed=1234)
    return cfg, sd, HipUNet2DConditionModel(cfg, sd, weight_dtype=weight_dtype)


def _pairs_reference(net, lat, pe, ne, t, idx):
    """Samples ``idx`` of a CFG batch evaluated one CFG pair at a time (UNet batch 2): returns (uncond, cond) rows."""
    un, co = [], []
    for i in idx:
        net.set_context(torch.cat([ne[i:i + 1], pe[i:i + 1]]).cuda())
        e = net.forward_latents(lat[i:i + 1].cuda(), 2, t).clone()
        un.append(e[0:1]); co.append(e[1:2])
    return torch.cat(un), torch.cat(co)


@pytest.fixture(scope="module")
def full_bf16():
    return _net(64)


def test_batch_consistency_at_bench_batches_bf16(full_bf16):
    """(a) UNet batch 16 with CFG (the headline: 8 images, rep = 2 plan), 64 with CFG (configs[2]) and 32 without CFG."""
    from oracle.unet import unet_forward
    cfg, sd, net = full_bf16
    t = 501.0
    lat, pe, ne = synth_inputs(cfg, 32, seed=5)
    probe = [0, 3, 7]
    ref_un, ref_co = _pairs_reference(net, lat, pe, ne, t, probe)
    with torch.no_grad():                                 # fp32 oracle for probe sample 0: [uncond, cond]
        orc = unet_forward(sd, oracle_cfg(cfg), torch.cat([lat[:1], lat[:1]]), t, torch.cat([ne[:1], pe[:1]]))
    e2 = max(rel_l2(ref_un[0:1], orc[0:1]), rel_l2(ref_co[0:1], orc[1:2]))
    print(f"64x64 bf16 UNet batch 2 sample 0 vs the fp32 oracle: {e2:.3e}")
    assert e2 < UNET_TOL
    worst, worst_orc = 0.0, e2
    for B in (8, 32):                                     # latent batch; UNet batch 2 B with the CFG-deduplicated prefix
        net.set_context(torch.cat([ne[:B], pe[:B]]).cuda())
        eps = net.forward_latents(lat[:B].cuda(), 2 * B, t).clone()
        assert torch.isfinite(eps).all()
        for k, i in enumerate(probe):
            eu, ec = rel_l2(eps[i:i + 1], ref_un[k:k + 1]), rel_l2(eps[B + i:B + i + 1], ref_co[k:k + 1])
            cs = min(cosine(eps[i:i + 1], ref_un[k:k + 1]), cosine(eps[B + i:B + i + 1], ref_co[k:k + 1]))
            worst = max(worst, eu, ec)
            print(f"64x64 bf16 UNet batch {2 * B} (CFG) sample {i}: uncond {eu:.3e} cond {ec:.3e} (cos {cs:.5f}) vs its batch-2 forward")
            assert eu < BATCH_TOL and ec < BATCH_TOL and cs > 0.9995
        eo = max(rel_l2(eps[0:1], orc[0:1]), rel_l2(eps[B:B + 1], orc[1:2]))
        worst_orc = max(worst_orc, eo)
        print(f"64x64 bf16 UNet batch {2 * B} (CFG) sample 0 vs the fp32 oracle: {eo:.3e} (its batch-2 forward: {e2:.3e})")
        assert eo < UNET_TOL and eo < 1.5 * e2
    # no CFG (LCM, configs[4] per-GPU share): UNet batch 32 = latent batch 32, conditional prompts only
    net.set_context(pe[:32].cuda())
    eps = net.forward_latents(lat[:32].cuda(), 32, t).clone()
    for k, i in enumerate(probe):
        e, cs = rel_l2(eps[i:i + 1], ref_co[k:k + 1]), cosine(eps[i:i + 1], ref_co[k:k + 1])
        worst = max(worst, e)
        print(f"64x64 bf16 UNet batch 32 (no CFG) sample {i}: {e:.3e} (cos {cs:.5f}) vs its batch-2 forward")
        assert e < BATCH_TOL and cs > 0.9995
    eo = rel_l2(eps[0:1], orc[1:2])
    print(f"64x64 bf16 UNet batch 32 (no CFG) sample 0 vs the fp32 oracle: {eo:.3e}")
    assert eo < UNET_TOL and eo < 1.5 * e2
    print(f"64x64 bf16 batch consistency: worst per-sample rel-L2 between batch sizes {worst:.3e} (tolerance {BATCH_TOL:.0e}), "
          f"worst vs the oracle {max(worst_orc, eo):.3e}")


def test_fp8_at_64x64_vs_emulating_oracle_and_batch_32(full_bf16):
    """(b) configs[4]'s operand path at the benchmark resolution."""
    from oracle.fp8 import Fp8Emulation
    from oracle.unet import unet_forward
    cfg, sd, _ = full_bf16
    _, _, net = _net(64, "fp8")
    assert net.weight_dtype == "fp8_e4m3"
    t = 499.0                                             # an LCM timestep (999, 759, 499, 259)
    lat, pe, ne = synth_inputs(cfg, 32, seed=9)
    ctx = torch.cat([ne[:1], pe[:1]])
    with torch.no_grad():
        fq = Fp8Emulation(sd)
        ref_q = unet_forward(sd, oracle_cfg(cfg), torch.cat([lat[:1], lat[:1]]), t, ctx, fq=fq)
        ref = unet_forward(sd, oracle_cfg(cfg), torch.cat([lat[:1], lat[:1]]), t, ctx)
    net.set_context(ctx.cuda())
    eps2 = net.forward_latents(lat[:1].cuda(), 2, t).clone()
    e_q, e_f, scheme = rel_l2(eps2, ref_q), rel_l2(eps2, ref), rel_l2(ref_q, ref)
    print(f"64x64 fp8 forward t={t}: vs emulating oracle {e_q:.3e} (cos {cosine(eps2, ref_q):.5f}); vs unquantised oracle "
          f"{e_f:.3e} (cos {cosine(eps2, ref):.5f}); emulated scheme vs unquantised {scheme:.3e}")
    assert torch.isfinite(eps2).all()
    assert e_q < FWD_TOL and cosine(eps2, ref) > 0.99
    assert e_f < FWD_EXCESS * scheme + 1e-2
    # batch 32 without CFG (the bench's configs[4] run) vs the same samples in batch-2 forwards
    net.set_context(pe[:32].cuda())
    eps32 = net.forward_latents(lat[:32].cuda(), 32, t).clone()
    assert torch.isfinite(eps32).all()
    worst = 0.0
    for i in (0, 13, 31):
        net.set_context(torch.cat([pe[i:i + 1], pe[i:i + 1]]).cuda())
        one = net.forward_latents(lat[i:i + 1].cuda(), 2, t)[0:1].clone()
        e = rel_l2(eps32[i:i + 1], one)
        worst = max(worst, e)
        print(f"64x64 fp8 UNet batch 32 sample {i}: {e:.3e} vs its batch-2 forward")
    assert worst < FP8_BATCH_TOL


def _free_running(cfg, sd, n_total, n_run, seed, net=None):
    """HIP loop vs oracle loop for the first ``n_run`` of ``n_total`` DDIM steps, CFG 7.5, batch 1, same initial latents."""
    from oracle.pipeline import sample_loop
    from oracle.schedulers import DDIMOracle
    from sonicdiffusionbayeslab_amd.registry import schedulers_registry
    from sonicdiffusionbayeslab_amd.schedulers import PNDMConfigStub
    from sonicdiffusionbayeslab_amd.unet import HipUNet2DConditionModel
    net = net or HipUNet2DConditionModel(cfg, sd)
    lat, pe, ne = synth_inputs(cfg, 1, seed=seed)
    ref, _, _, _ = sample_loop(sd, oracle_cfg(cfg), DDIMOracle(), pe, ne, lat, n_total, 7.5, max_steps=n_run)
    s = schedulers_registry["ddim_scheduler"].from_config(PNDMConfigStub().config)
    s.set_timesteps(n_total, device="cuda")
    net.set_context(torch.cat([ne, pe]).cuda())
    x = lat.cuda()
    for t in s._timesteps_list[:n_run]:
        eps = net.forward_latents(x, 2, float(t))
        x, _ = s.step_fused(eps, 7.5, x, t, cfg=True)
    got = x.float().cpu()
    return (got - ref).abs().max().item(), rel_l2(got, ref), cosine(got, ref)


def test_free_running_50_ddim_steps_at_16x16():
    """(c) the headline's step count, reduced resolution (the oracle's 100 sample-forwards take seconds at 16x16)."""
    from sonicdiffusionbayeslab_amd.weights import UNetConfig, make_synthetic_state_dict
    cfg = UNetConfig(sample_size=16)
    sd = make_synthetic_state_dict(cfg, seed=1234)
    ma, rl, cs = _free_running(cfg, sd, 50, 50, seed=29)
    print(f"free-running DDIM 50/50 steps, CFG 7.5, 16x16, batch 1: max-abs {ma:.3e} rel-L2 {rl:.3e} cosine {cs:.5f}")
    assert rl < DRIFT_TOL_50 and cs > 0.99


def test_free_running_10_of_50_ddim_steps_at_64x64(full_bf16):
    """(c) the benchmark's resolution and schedule, first 10 steps (20 oracle sample-forwards at 64x64)."""
    cfg, sd, net = full_bf16
    ma, rl, cs = _free_running(cfg, sd, 50, 10, seed=29, net=net)
    print(f"free-running DDIM 10/50 steps, CFG 7.5, 64x64, batch 1: max-abs {ma:.3e} rel-L2 {rl:.3e} cosine {cs:.5f}")
    assert rl < DRIFT_TOL_10 and cs > 0.995


@pytest.mark.skipif(os.environ.get("SD_LONG_PARITY") != "1", reason="~5 min of fp32 CPU oracle; run with SD_LONG_PARITY=1")
def test_free_running_50_of_50_ddim_steps_at_64x64(full_bf16):
    """(c) at full length: the headline's loop -- 50 of 50 DDIM steps, CFG 7.5, 64x64 latents, batch 1 -- against the oracle's
    loop from the same latents (100 oracle sample-forwards at 64x64), the drift printed every 10 steps.  Opt-in because of
    its CPU time; the figures of the last run are in profiles/round3_notes.md."""
    from oracle.pipeline import sample_loop
    from oracle.schedulers import DDIMOracle
    from sonicdiffusionbayeslab_amd.registry import schedulers_registry
    from sonicdiffusionbayeslab_amd.schedulers import PNDMConfigStub
    cfg, sd, net = full_bf16
    lat, pe, ne = synth_inputs(cfg, 1, seed=29)
    _, _, _, traj = sample_loop(sd, oracle_cfg(cfg), DDIMOracle(), pe, ne, lat, 50, 7.5)
    s = schedulers_registry["ddim_scheduler"].from_config(PNDMConfigStub().config)
    s.set_timesteps(50, device="cuda")
    net.set_context(torch.cat([ne, pe]).cuda())
    x = lat.cuda()
    for i, t in enumerate(s._timesteps_list):
        eps = net.forward_latents(x, 2, float(t))
        x, _ = s.step_fused(eps, 7.5, x, t, cfg=True)
        if (i + 1) % 10 == 0:
            got, ref = x.float().cpu(), traj["latents"][i]
            print(f"free-running DDIM {i + 1}/50 steps, CFG 7.5, 64x64, batch 1: max-abs {(got - ref).abs().max().item():.3e} "
                  f"rel-L2 {rel_l2(got, ref):.3e} cosine {cosine(got, ref):.5f}")
    got, ref = x.float().cpu(), traj["latents"][-1]
    assert rel_l2(got, ref) < DRIFT_TOL_50 and cosine(got, ref) > 0.99


@pytest.mark.skipif(os.environ.get("SD_LONG_PARITY") != "1", reason="~5 min of fp32 CPU oracle; run with SD_LONG_PARITY=1")
@pytest.mark.parametrize("which", ["dpm_solver_pp_20", "lcm_4", "lcm_4_fp8", "deepcache_n3_50"])
def test_other_configs_full_loops_at_64x64(full_bf16, which):
    """BASELINE configs[2..4] at the benchmark's resolution and FULL length, through the product pipeline
    (StableDiffusionModel.__call__ -> libsdhip) against the oracle's loop from the same latents: DPM-Solver++ (order 2)
    20 steps with CFG 7.5; LCM 4 steps without CFG (pre-drawn re-noising tensors); DeepCache N = 3, branch 0, over 50 DDIM
    steps with CFG 7.5.  Opt-in; the figures of the last run are in profiles/round3_notes.md."""
    from oracle.pipeline import sample_loop
    from oracle.schedulers import DDIMOracle, DPMSolverOracle, LCMOracle
    from oracle.unet import DeepCacheState
    from sonicdiffusionbayeslab_amd.deepcache import DeepCacheSDHelper
    from sonicdiffusionbayeslab_amd.models import StableDiffusionModel
    from sonicdiffusionbayeslab_amd.registry import schedulers_registry
    from sonicdiffusionbayeslab_amd.schedulers import PNDMConfigStub
    cfg, sd, _ = full_bf16
    stub = PNDMConfigStub().config
    if which == "lcm_4_fp8":
        # configs[4] with fp8-e4m3 weights + activations: against the oracle EMULATING the scheme with the per-tensor
        # scales the pipeline calibrated on this call (and, printed, against the unquantised oracle)
        from oracle.fp8 import Fp8Emulation
        model = StableDiffusionModel(unet_config=cfg, state_dict=dict(sd), weight_dtype="fp8").to("cuda:0")
        model.scheduler = schedulers_registry["lcm_scheduler"].from_config(stub)
        lat, pe, ne = synth_inputs(cfg, 2, seed=33)
        noise = torch.randn(3, 2, 4, cfg.sample_size, cfg.sample_size, generator=torch.Generator().manual_seed(8))
        out, _, _ = model(prompt_embeds=pe, latents=lat, num_inference_steps=4, guidance_scale=0.0, output_type="latent",
                          step_noise=noise.cuda())
        scales = {k: v for k, v in model.unet.fp8_scales().items()}
        ref = sample_loop(sd, oracle_cfg(cfg), LCMOracle(), pe, None, lat, 4, 0.0, lcm_noise=noise, fq=Fp8Emulation(sd, scales=scales))[0]
        plain = sample_loop(sd, oracle_cfg(cfg), LCMOracle(), pe, None, lat, 4, 0.0, lcm_noise=noise)[0]
        got = out.images.float().cpu()
        err, cs = rel_l2(got, ref), cosine(got, ref)
        print(f"{which} at 64x64, full loop: vs emulating oracle rel-L2 {err:.3e} cosine {cs:.5f}; vs unquantised oracle "
              f"{rel_l2(got, plain):.3e}; emulated scheme vs unquantised {rel_l2(ref, plain):.3e}")
        assert err < DRIFT_TOL_50 and cs > 0.99
        return
    model = StableDiffusionModel(unet_config=cfg, state_dict=dict(sd)).to("cuda:0")
    if which == "dpm_solver_pp_20":
        kw = dict(solver_order=2, algorithm_type="dpmsolver++", final_sigmas_type="zero")
        model.scheduler = schedulers_registry["dpm_solver_scheduler"].from_config(stub, **kw)
        lat, pe, ne = synth_inputs(cfg, 1, seed=31)
        out, _, _ = model(prompt_embeds=pe, negative_prompt_embeds=ne, latents=lat, num_inference_steps=20, guidance_scale=7.5,
                          output_type="latent")
        ref = sample_loop(sd, oracle_cfg(cfg), DPMSolverOracle(**kw), pe, ne, lat, 20, 7.5)[0]
    elif which == "lcm_4":
        model.scheduler = schedulers_registry["lcm_scheduler"].from_config(stub)
        lat, pe, ne = synth_inputs(cfg, 2, seed=33)
        noise = torch.randn(3, 2, 4, cfg.sample_size, cfg.sample_size, generator=torch.Generator().manual_seed(8))
        out, _, _ = model(prompt_embeds=pe, latents=lat, num_inference_steps=4, guidance_scale=0.0, output_type="latent",
                          step_noise=noise.cuda())
        ref = sample_loop(sd, oracle_cfg(cfg), LCMOracle(), pe, None, lat, 4, 0.0, lcm_noise=noise)[0]
    else:
        model.scheduler = schedulers_registry["ddim_scheduler"].from_config(stub)
        lat, pe, ne = synth_inputs(cfg, 1, seed=41)
        helper = DeepCacheSDHelper(pipe=model)
        helper.set_params(cache_interval=3, cache_branch_id=0)
        helper.enable()
        try:
            out, _, _ = model(prompt_embeds=pe, negative_prompt_embeds=ne, latents=lat, num_inference_steps=50,
                              guidance_scale=7.5, output_type="latent")
        finally:
            helper.disable()
        dc = DeepCacheState(cache_interval=3, cache_branch_id=0, enabled=True)
        ref = sample_loop(sd, oracle_cfg(cfg), DDIMOracle(), pe, ne, lat, 50, 7.5, deepcache=dc)[0]
    got = out.images.float().cpu()
    err, cs = rel_l2(got, ref), cosine(got, ref)
    print(f"{which} at 64x64, full loop: max-abs {(got - ref).abs().max().item():.3e} rel-L2 {err:.3e} cosine {cs:.5f}")
    assert err < DRIFT_TOL_50 and cs > 0.99


def test_vae_decode_64_to_512_matches_oracle():
    """(d) the decode the end-to-end path really runs: [1,4,64,64] -> [1,3,512,512]."""
    from oracle.vae import VaeConfig as OC, vae_decode
    from sonicdiffusionbayeslab_amd.vae import HipVaeDecoder, VaeConfig, make_synthetic_vae_state_dict
    cfg = VaeConfig(sample_size=64)
    sd = make_synthetic_vae_state_dict(cfg)
    dec = HipVaeDecoder(cfg, sd)
    lat = torch.randn(1, 4, 64, 64, generator=torch.Generator().manual_seed(4))
    inv = 1.0 / cfg.scaling_factor
    with torch.no_grad():
        ref = vae_decode(sd, OC(**dataclasses.asdict(cfg)), lat * inv)
    got = dec.decode(lat.cuda(), inv)
    torch.cuda.synchronize()
    err, cs = rel_l2(got, ref), cosine(got, ref)
    print(f"VAE decode 64x64 -> 512x512, batch 1: rel-L2 {err:.3e} cosine {cs:.5f} max-abs {(got.cpu() - ref).abs().max():.3e}")
    assert got.shape == (1, 3, 512, 512) and torch.isfinite(got).all()
    assert err < 2e-2 and cs > 0.999


def test_deepcache_plan_at_the_bench_batch(full_bf16):
    """configs[3] per-GPU share (DeepCache N = 3, branch 0, batch 16 with CFG = UNet batch 32): the full-and-store step and
    the skip step that re-uses its cache, each sample against the same pair of steps run at UNet batch 2 (the batch the
    oracle parity of tests/test_fullsize_gpu.py covers).  Same noise-floor tolerance as (a); no oracle time."""
    from sonicdiffusionbayeslab_amd.unet import CACHE_FULL_AND_STORE, CACHE_SKIP
    cfg, sd, net = full_bf16
    lat, pe, ne = synth_inputs(cfg, 16, seed=15)
    lat2 = lat + 0.05 * synth_inputs(cfg, 16, seed=16)[0]            # the latents a step later
    net.set_deepcache(0)
    try:
        net.set_context(torch.cat([ne, pe]).cuda())
        full = net.forward_latents(lat.cuda(), 32, 981.0, cache_mode=CACHE_FULL_AND_STORE).clone()
        skip = net.forward_latents(lat2.cuda(), 32, 961.0, cache_mode=CACHE_SKIP).clone()
        assert torch.isfinite(full).all() and torch.isfinite(skip).all()
        for i in (0, 9, 15):
            net.set_context(torch.cat([ne[i:i + 1], pe[i:i + 1]]).cuda())
            f2 = net.forward_latents(lat[i:i + 1].cuda(), 2, 981.0, cache_mode=CACHE_FULL_AND_STORE).clone()
            s2 = net.forward_latents(lat2[i:i + 1].cuda(), 2, 961.0, cache_mode=CACHE_SKIP).clone()
            ef = max(rel_l2(full[i:i + 1], f2[0:1]), rel_l2(full[16 + i:17 + i], f2[1:2]))
            es = max(rel_l2(skip[i:i + 1], s2[0:1]), rel_l2(skip[16 + i:17 + i], s2[1:2]))
            print(f"64x64 DeepCache(3,0) UNet batch 32 sample {i}: full step {ef:.3e}, skip step {es:.3e} vs its batch-2 pair of steps")
            assert ef < BATCH_TOL and es < BATCH_TOL
    finally:
        net.set_deepcache(-1)
