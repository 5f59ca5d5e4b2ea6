"""GPU parity of the sampling hot path against the CPU oracle on identical seeded weights,
prompt embeddings and initial latents (SURVEY.md §8d synthetic inputs), through the product's
plugin surface (registries -> pipeline -> libsdhip).

Tolerances (bf16 storage / fp32 accumulate vs the fp32 oracle):
  * one fused scheduler step:            rel-L2 <= 1e-5  (fp32 kernel, coefficient re-association)
  * teacher-forced step (same inputs):   rel-L2 <= 2e-2 on the noise prediction
  * free-running N-step latents:         rel-L2 <= 6e-2, cosine >= 0.998  (errors compound over steps)
"""
import dataclasses

import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.util import cosine, oracle_cfg, rel_l2, synth_inputs

FREE_TOL, FREE_COS = 6e-2, 0.998


@pytest.fixture(scope="module")
def env():
    from sonicdiffusionbayeslab_amd.models import StableDiffusionModel
    from sonicdiffusionbayeslab_amd.weights import UNetConfig, make_synthetic_state_dict
    cfg = UNetConfig(sample_size=16)
    sd = make_synthetic_state_dict(cfg, seed=1234)
    model = StableDiffusionModel(unet_config=cfg, state_dict=dict(sd)).to("cuda:0")
    return cfg, sd, model


def _sched(model, name, **kw):
    from sonicdiffusionbayeslab_amd.registry import schedulers_registry
    from sonicdiffusionbayeslab_amd.schedulers import PNDMConfigStub
    model.scheduler = schedulers_registry[name].from_config(PNDMConfigStub().config, **kw)
    return model.scheduler


# ---------------------------------------------------------------- fused scheduler.step kernel
@pytest.mark.parametrize("kind,kw,n", [
    ("ddim", {}, 50), ("ddim", {}, 3),
    ("dpm", dict(solver_order=2, algorithm_type="dpmsolver++", final_sigmas_type="zero"), 20),
    ("dpm", dict(solver_order=3, algorithm_type="dpmsolver++", final_sigmas_type="zero"), 7),
    ("dpm", dict(solver_order=2, algorithm_type="dpmsolver", final_sigmas_type="sigma_min"), 10),
    ("dpm", dict(solver_order=2, algorithm_type="sde-dpmsolver++", final_sigmas_type="zero"), 10),   # src/schedulers.py:134-147
    ("dpm", dict(solver_order=3, algorithm_type="sde-dpmsolver++", final_sigmas_type="zero"), 20),
    ("dpm", dict(solver_order=2, algorithm_type="sde-dpmsolver", final_sigmas_type="sigma_min"), 8),
    ("lcm", {}, 4),
    ("pndm", {}, 8),
])
def test_scheduler_step_matches_oracle(kind, kw, n):
    from oracle.schedulers import DDIMOracle, DPMSolverOracle, LCMOracle, PNDMOracle
    from sonicdiffusionbayeslab_amd.registry import schedulers_registry
    from sonicdiffusionbayeslab_amd.schedulers import PNDMConfigStub
    name = {"ddim": "ddim_scheduler", "dpm": "dpm_solver_scheduler", "lcm": "lcm_scheduler",
            "pndm": "pndm_scheduler"}[kind]
    s = schedulers_registry[name].from_config(PNDMConfigStub().config, **kw)
    o = {"ddim": DDIMOracle, "dpm": DPMSolverOracle, "lcm": LCMOracle, "pndm": PNDMOracle}[kind](**kw)
    s.set_timesteps(n, device="cuda"); o.set_timesteps(n)
    assert [int(t) for t in s.timesteps.cpu()] == [int(t) for t in o.timesteps]
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 4, 16, 16, generator=g)
    xo = x.clone()
    for i, t in enumerate(o.timesteps):
        e2 = torch.randn(4, 4, 16, 16, generator=g)          # [uncond | text] halves
        eo = e2[:2] + 7.5 * (e2[2:] - e2[:2])
        kwo, kws = {}, {}
        if kind == "lcm" and i < n - 1:
            z = torch.randn(2, 4, 16, 16, generator=g)
            kwo["noise"], kws["noise"] = z, z.cuda()
        if kw.get("algorithm_type", "").startswith("sde-"):
            z = torch.randn(2, 4, 16, 16, generator=g)
            kwo["variance_noise"], kws["variance_noise"] = z, z.cuda()
        so = o.step(eo, t, xo, **kwo)
        sg = s.step_fused(e2.cuda(), 7.5, x.cuda(), int(t), cfg=True, **kws)
        assert len(so) == len(sg)
        for got, ref in zip(sg, so):
            assert rel_l2(got, ref) < 1e-5, (i, rel_l2(got, ref))
        xo = so[0]
        x = xo.clone()                                          # teacher-force the next step


def test_scheduler_drop_in_step_signature(env):
    """`scheduler.step(noise_pred, t, latents, return_dict=False)` of src/models.py:253."""
    from oracle.schedulers import DDIMOracle
    cfg, sd, model = env
    s = _sched(model, "ddim_scheduler"); s.set_timesteps(10, device="cuda")
    o = DDIMOracle(); o.set_timesteps(10)
    g = torch.Generator().manual_seed(1)
    x, e = torch.randn(1, 4, 16, 16, generator=g), torch.randn(1, 4, 16, 16, generator=g)
    out = s.step(e.cuda().half(), s.timesteps[0], x.cuda().half(), return_dict=False)
    assert len(out) == 2 and out[0].dtype == torch.float16
    ref = o.step(e.half().float(), int(o.timesteps[0]), x.half().float())
    assert rel_l2(out[0], ref[0]) < 2e-3


# ---------------------------------------------------------------- whole loops through the plugin surface
def _oracle_loop(cfg, sd, sched, pe, ne, lat, n, gs, **kw):
    from oracle.pipeline import sample_loop
    return sample_loop(sd, oracle_cfg(cfg), sched, pe, ne, lat, n, gs, **kw)


def test_ddim_loop_free_running_and_teacher_forced(env):
    from oracle.schedulers import DDIMOracle
    from oracle.unet import unet_forward
    cfg, sd, model = env
    lat, pe, ne = synth_inputs(cfg, 2)
    _sched(model, "ddim_scheduler")
    n = 6
    out, secs, x0s = model(prompt_embeds=pe, negative_prompt_embeds=ne, latents=lat, num_inference_steps=n,
                           guidance_scale=7.5, output_type="latent")
    ref, _, ref_x0, traj = _oracle_loop(cfg, sd, DDIMOracle(), pe, ne, lat, n, 7.5)
    err, cs = rel_l2(out.images, ref), cosine(out.images, ref)
    print(f"DDIM {n} steps free-running rel-L2 {err:.3e} cos {cs:.5f}; loop {secs*1e3:.1f} ms")
    assert len(x0s) == n and secs > 0 and model.num_timesteps == n
    assert err < FREE_TOL and cs > FREE_COS
    # teacher-forced: feed the ORACLE's latents of step k to the HIP UNet, compare noise predictions
    ctx = torch.cat([ne, pe])
    model.unet.set_deepcache(-1)
    model.unet.set_context(ctx.cuda())
    ts = [int(t) for t in DDIMOracle().__class__().alphas_cumprod[:0]] or None
    o = DDIMOracle(); o.set_timesteps(n)
    for k in (0, 3, 5):
        x_in = lat if k == 0 else traj["latents"][k - 1]
        eps = model.unet.forward_latents(x_in.cuda(), 4, float(o.timesteps[k]))
        with torch.no_grad():
            e_ref = unet_forward(sd, oracle_cfg(cfg), torch.cat([x_in, x_in]), o.timesteps[k], ctx)
        assert rel_l2(eps, e_ref) < 2e-2, (k, rel_l2(eps, e_ref))


def test_dpm_solver_pp_loop(env):
    from oracle.schedulers import DPMSolverOracle
    cfg, sd, model = env
    lat, pe, ne = synth_inputs(cfg, 1, seed=31)
    kw = dict(solver_order=2, algorithm_type="dpmsolver++", final_sigmas_type="zero")
    _sched(model, "dpm_solver_scheduler", **kw)
    out, _, x0s = model(prompt_embeds=pe, negative_prompt_embeds=ne, latents=lat, num_inference_steps=5,
                        guidance_scale=7.5, output_type="latent")
    ref, _, _, _ = _oracle_loop(cfg, sd, DPMSolverOracle(**kw), pe, ne, lat, 5, 7.5)
    err, cs = rel_l2(out.images, ref), cosine(out.images, ref)
    print(f"DPM++ 5 steps rel-L2 {err:.3e} cos {cs:.5f}")
    assert err < FREE_TOL and cs > FREE_COS and len(x0s) == 5


def test_lcm_loop_no_cfg(env):
    from oracle.schedulers import LCMOracle
    cfg, sd, model = env
    lat, pe, ne = synth_inputs(cfg, 2, seed=33)
    _sched(model, "lcm_scheduler")
    g = torch.Generator().manual_seed(8)
    noise = torch.randn(3, 2, 4, 16, 16, generator=g)
    out, _, _ = model(prompt_embeds=pe, latents=lat, num_inference_steps=4, guidance_scale=0.0,
                      output_type="latent", step_noise=noise.cuda())
    ref, _, _, _ = _oracle_loop(cfg, sd, LCMOracle(), pe, None, lat, 4, 0.0, lcm_noise=noise)
    err, cs = rel_l2(out.images, ref), cosine(out.images, ref)
    print(f"LCM 4 steps rel-L2 {err:.3e} cos {cs:.5f}")
    assert err < FREE_TOL and cs > FREE_COS


_PLAIN = {}


@pytest.mark.parametrize("interval,branch", [(3, 0), (2, 0), (2, 1), (3, 4)])
def test_deepcache_loop(env, interval, branch):
    """DeepCacheSDHelper call pattern of src/experiments/deep_cache.py:24-29,58; the skip-step plan
    must reproduce the oracle's module-level cache semantics (A.5) for several branches."""
    from oracle.schedulers import DDIMOracle
    from oracle.unet import DeepCacheState
    from sonicdiffusionbayeslab_amd.deepcache import DeepCacheSDHelper
    cfg, sd, model = env
    lat, pe, ne = synth_inputs(cfg, 1, seed=41)
    _sched(model, "ddim_scheduler")
    helper = DeepCacheSDHelper(pipe=model)
    helper.set_params(cache_interval=interval, cache_branch_id=branch)
    helper.enable()
    try:
        out, _, _ = model(prompt_embeds=pe, negative_prompt_embeds=ne, latents=lat, num_inference_steps=7,
                          guidance_scale=7.5, output_type="latent")
    finally:
        helper.disable()
    dc = DeepCacheState(cache_interval=interval, cache_branch_id=branch, enabled=True)
    ref, _, _, _ = _oracle_loop(cfg, sd, DDIMOracle(), pe, ne, lat, 7, 7.5, deepcache=dc)
    if "plain" not in _PLAIN:           # the same inputs for every (interval, branch): one oracle loop without the cache
        _PLAIN["plain"] = _oracle_loop(cfg, sd, DDIMOracle(), pe, ne, lat, 7, 7.5)[0]
    plain = _PLAIN["plain"]
    err, cs = rel_l2(out.images, ref), cosine(out.images, ref)
    print(f"DeepCache N={interval} branch={branch}: rel-L2 {err:.3e} cos {cs:.5f}; cached-vs-plain {rel_l2(ref, plain):.3e}")
    assert err < FREE_TOL and cs > FREE_COS
    assert model._deepcache is None


def test_pndm_deepcache_literal_reference_setup(env):
    """The reference's `deep_cache` method as committed: checkpoint PNDM scheduler (N+1 UNet calls,
    duplicated timestep) + DeepCacheSDHelper, whose list.index() quirk maps both 961 steps to index 1."""
    from oracle.schedulers import PNDMOracle
    from oracle.unet import DeepCacheState
    from sonicdiffusionbayeslab_amd.deepcache import DeepCacheSDHelper
    cfg, sd, model = env
    lat, pe, ne = synth_inputs(cfg, 1, seed=43)
    _sched(model, "pndm_scheduler")
    helper = DeepCacheSDHelper(pipe=model)
    helper.set_params(cache_interval=3, cache_branch_id=0)
    helper.enable()
    try:
        out, _, x0s = model(prompt_embeds=pe, negative_prompt_embeds=ne, latents=lat, num_inference_steps=6,
                            guidance_scale=7.5, output_type="latent")
    finally:
        helper.disable()
    assert model.num_timesteps == 7 and x0s == []          # PNDM's step returns a 1-tuple
    dc = DeepCacheState(cache_interval=3, cache_branch_id=0, enabled=True)
    ref, _, _, _ = _oracle_loop(cfg, sd, PNDMOracle(), pe, ne, lat, 6, 7.5, deepcache=dc)
    err, cs = rel_l2(out.images, ref), cosine(out.images, ref)
    print(f"PNDM+DeepCache N=3: rel-L2 {err:.3e} cos {cs:.5f}")
    assert err < FREE_TOL and cs > FREE_COS


def test_prompt_strings_and_harness_call_shape(env):
    """`model(prompts, num_inference_steps=, guidance_scale=, generator=, output_type=)` ->
    (obj.images, seconds, x0_preds) as consumed by base_experiment.py:145-152."""
    cfg, sd, model = env
    _sched(model, "ddim_scheduler")
    g = torch.Generator(device="cpu").manual_seed(29)
    res = model(["a man on a snowboard is coming down a slope", "People swim in the ocean"],
                num_inference_steps=2, guidance_scale=7.5, generator=g, output_type="latent")
    assert len(res) == 3
    imgs, secs, x0s = res
    assert imgs.images.shape == (2, 4, 16, 16) and isinstance(secs, float) and len(x0s) == 2
    g2 = torch.Generator(device="cpu").manual_seed(29)
    again = model(["a man on a snowboard is coming down a slope", "People swim in the ocean"],
                  num_inference_steps=2, guidance_scale=7.5, generator=g2, output_type="latent")[0]
    assert torch.equal(imgs.images, again.images)        # deterministic: same seed, same kernels
    # "pil" / "np" (the reference's default is "pil", src/models.py:312-321): decoded, denormalised, post-processed on the host
    pil = model(["x"], num_inference_steps=1, output_type="pil")[0].images
    assert len(pil) == 1 and pil[0].size == (128, 128) and pil[0].mode == "RGB"
    arr = model(["x"], num_inference_steps=1, output_type="np")[0].images
    assert arr.shape == (1, 128, 128, 3) and arr.min() >= 0.0 and arr.max() <= 1.0
    with pytest.raises(NotImplementedError):
        model(["x"], num_inference_steps=1, output_type="jpeg")


@pytest.mark.parametrize("name,runs", [("ddim_config.yaml", 8), ("dpm_solver_config.yaml", 7),
                                        ("consistency_model_config.yaml", 7), ("deep_cache_config.yaml", 10)])
def test_method_plugins_run_from_yaml(env, monkeypatch, capsys, name, runs):
    """the four in-scope method plugins (src/experiments/{ddim,dpm_solver,consistency_model,
    deep_cache}.py) run `run_experiment()` end to end from their shipped YAML -- the reference's own sweep lists --
    through the registries (tiny UNet, one prompt batch of 2 per sweep point)."""
    import json, os
    from sonicdiffusionbayeslab_amd import models as M
    from sonicdiffusionbayeslab_amd.config import load_config
    from sonicdiffusionbayeslab_amd.registry import methods_registry
    from sonicdiffusionbayeslab_amd.weights import UNetConfig
    cfg, sd, _ = env
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    monkeypatch.chdir(root)
    monkeypatch.setattr(M.StableDiffusionModel, "from_pretrained",
                        classmethod(lambda c, *a, **k: c(unet_config=UNetConfig(sample_size=16), state_dict=dict(sd))))
    conf = load_config(os.path.join(root, "configs", name))
    conf.inference.batch_size = 2
    conf.inference.batch_count = 1
    conf.inference.output_type = "latent"
    m = methods_registry[conf.experiment.method](conf)
    m.run_experiment()
    lines = [json.loads(l) for l in capsys.readouterr().out.splitlines() if l.startswith("{")]
    assert len(lines) == runs, (len(lines), runs)
    for rec in lines:
        assert rec["images"] == 2 and rec["time_metric_s_per_image"] > 0 and rec["nfe"] >= 1
    if name == "deep_cache_config.yaml":      # checkpoint scheduler (PNDM): N steps = N + 1 UNet calls
        assert [r["nfe"] for r in lines[:5]] == [6, 11, 101, 501, 101]
    if name == "consistency_model_config.yaml":
        assert all("SYNTHETIC low-rank stand-in" in r["weights"] for r in lines)
