"""Generates tests/golden/loop_golden_64.npz: the ORACLE's latent trajectories of the full-length sampling loops of the
four GPU configs of BASELINE.json at the benchmark's resolution (64x64 latents), so that the driver-run ``-m gpu`` suite
compares the product pipeline against them in seconds instead of re-running ~12 minutes of fp32 CPU oracle
(tests/test_benchshapes_gpu.py::test_full_length_loops_at_64x64_against_the_oracle_fixtures).

Run from the repo root on the CPU:  ``python tests/golden/make_loop_golden.py``  (about 12 minutes on 8 cores).

What is stored (fp32; every seed equals the one the GPU test uses -- the test re-draws the inputs and asserts they are
bit-identical to the stored ones before comparing anything):
  ddim50      configs[1]: DDIM 50 steps, CFG 7.5, batch 1, synth_inputs seed 29 -- latents after steps 10/20/30/40/50
  dpmpp20     configs[2]: DPM-Solver++ (order 2, final sigma zero) 20 steps, CFG 7.5, batch 1, seed 31 -- after 10/20
  deepcache50 configs[3]: DeepCache N = 3, branch 0 on DDIM 50 steps, CFG 7.5, batch 1, seed 41 -- after 10/20/30/40/50
  ddim50_b8   configs[1] at the HEADLINE batch: DDIM 50 steps, CFG 7.5, batch 8 (UNet batch 16: the rep = 2 CFG plan, the
              128-row tiles, no split-K at the upper levels, the arena of bench.py), seed 43 -- samples 0 and 7 after steps
              10/20/30/40 and all 8 samples after step 50 (round 5; ~25 min of oracle on its own)
  lcm4        configs[4]: LCM 4 steps, no CFG, batch 2, seed 33, re-noising tensors from seed 8 -- after every step
  lcm4_fp8    the same loop under oracle.fp8.Fp8Emulation with the per-tensor activation scales of an ORACLE-side
              calibration (Fp8AmaxRecorder at t = 999 / 499 / 259 on the loop's own initial latents, margin 2: the
              product's rule); the scales are stored beside the latents and the GPU test restores them through
              sd_unet_set_fp8_scale, so both sides quantise at the same points with the same scales.

The oracle is this build's fp32 CPU restatement (oracle/__init__.py: PARITY UNPINNED -- the reference holds no fixtures
and its arithmetic lives in diffusers 0.32.1 / DeepCache 0.1.1, absent offline): these files pin HIP-vs-oracle, not
oracle-vs-reference.  Reference call sites restated by the loops: src/models.py:210-282, src/schedulers.py:98-187,
configs/consistency_model_config.yaml:1-34, src/experiments/deep_cache.py:20-58."""
import hashlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle.fp8 import Fp8AmaxRecorder, Fp8Emulation  # noqa: E402
from oracle.pipeline import sample_loop  # noqa: E402
from oracle.schedulers import DDIMOracle, DPMSolverOracle, LCMOracle  # noqa: E402
from oracle.unet import DeepCacheState, unet_forward  # noqa: E402
from sonicdiffusionbayeslab_amd.weights import UNetConfig, make_synthetic_state_dict  # noqa: E402
from tests.util import oracle_cfg, synth_inputs  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "loop_golden_64.npz")
WEIGHTS_SEED = 1234
DPM_KW = dict(solver_order=2, algorithm_type="dpmsolver++", final_sigmas_type="zero")
LCM_NOISE_SEED = 8
SEEDS = {"ddim50": 29, "dpmpp20": 31, "deepcache50": 41, "lcm4": 33, "ddim50_b8": 43}
B8_PROBES = (0, 7)


def weights_fingerprint(sd) -> str:
    """SHA-256 over a few parameters: the fixture is only valid for these synthetic weights."""
    h = hashlib.sha256()
    for k in ("conv_in.weight", "mid_block.resnets.0.conv1.weight", "up_blocks.3.attentions.2.transformer_blocks.0.ff.net.2.weight",
              "conv_out.bias"):
        h.update(sd[k].detach().float().contiguous().numpy().tobytes())
    return h.hexdigest()[:16]


def lcm_noise(cfg, batch):
    return torch.randn(3, batch, 4, cfg.sample_size, cfg.sample_size, generator=torch.Generator().manual_seed(LCM_NOISE_SEED))


def main():
    torch.set_num_threads(int(os.environ.get("ORACLE_THREADS", os.cpu_count() or 8)))
    cfg = UNetConfig(sample_size=64)
    ocfg = oracle_cfg(cfg)
    sd = make_synthetic_state_dict(cfg, seed=WEIGHTS_SEED)
    out = {"weights_fingerprint": np.array(weights_fingerprint(sd))}
    if os.path.exists(OUT):         # sections already there are kept (delete the file to regenerate everything)
        with np.load(OUT) as old:
            if str(old["weights_fingerprint"]) == str(out["weights_fingerprint"]):
                out.update({k: old[k] for k in old.files})
    t0 = time.time()

    def have(name):
        done = any(k.startswith(name + "/step") for k in out)
        if done:
            print(f"{name}: kept from {OUT}", flush=True)
        return done

    def save():
        np.savez(OUT, **out)

    def keep(name, traj, every):
        for i, x in enumerate(traj["latents"]):
            if (i + 1) % every == 0:
                out[f"{name}/step{i + 1}"] = x.float().numpy()

    if not have("ddim50"):
        lat, pe, ne = synth_inputs(cfg, 1, seed=SEEDS["ddim50"])
        out["ddim50/init"] = lat.numpy()
        _, _, _, traj = sample_loop(sd, ocfg, DDIMOracle(), pe, ne, lat, 50, 7.5)
        keep("ddim50", traj, 10)
        save()
        print(f"ddim50 done at {time.time() - t0:.0f} s", flush=True)

    if not have("dpmpp20"):
        lat, pe, ne = synth_inputs(cfg, 1, seed=SEEDS["dpmpp20"])
        out["dpmpp20/init"] = lat.numpy()
        _, _, _, traj = sample_loop(sd, ocfg, DPMSolverOracle(**DPM_KW), pe, ne, lat, 20, 7.5)
        keep("dpmpp20", traj, 10)
        save()
        print(f"dpmpp20 done at {time.time() - t0:.0f} s", flush=True)

    if not have("deepcache50"):
        lat, pe, ne = synth_inputs(cfg, 1, seed=SEEDS["deepcache50"])
        out["deepcache50/init"] = lat.numpy()
        dc = DeepCacheState(cache_interval=3, cache_branch_id=0, enabled=True)
        _, _, _, traj = sample_loop(sd, ocfg, DDIMOracle(), pe, ne, lat, 50, 7.5, deepcache=dc)
        keep("deepcache50", traj, 10)
        save()
        print(f"deepcache50 done at {time.time() - t0:.0f} s", flush=True)

    if not have("ddim50_b8"):
        lat, pe, ne = synth_inputs(cfg, 8, seed=SEEDS["ddim50_b8"])
        out["ddim50_b8/init"] = lat.numpy()
        _, _, _, traj = sample_loop(sd, ocfg, DDIMOracle(), pe, ne, lat, 50, 7.5)
        for i, x in enumerate(traj["latents"]):
            if (i + 1) % 10 == 0:
                x = x.float()
                out[f"ddim50_b8/step{i + 1}"] = (x if i + 1 == 50 else x[list(B8_PROBES)]).numpy()
        save()
        print(f"ddim50_b8 done at {time.time() - t0:.0f} s", flush=True)

    lat, pe, ne = synth_inputs(cfg, 2, seed=SEEDS["lcm4"])
    noise = lcm_noise(cfg, 2)
    if have("lcm4") and have("lcm4_fp8"):
        save()
        print(f"wrote {OUT}: {os.path.getsize(OUT) / 1e6:.2f} MB, {len(out)} arrays")
        return
    out["lcm4/init"] = lat.numpy()
    out["lcm4/noise"] = noise.numpy()
    _, _, _, traj = sample_loop(sd, ocfg, LCMOracle(), pe, None, lat, 4, 0.0, lcm_noise=noise)
    keep("lcm4", traj, 1)
    # oracle-side calibration of the e4m3 activation scales: first / middle / last timestep of the 4-step schedule
    rec = Fp8AmaxRecorder(sd)
    s = LCMOracle()
    s.set_timesteps(4)
    ts = [int(t) for t in s.timesteps]
    with torch.no_grad():
        for t in dict.fromkeys([ts[0], ts[len(ts) // 2], ts[-1]]):
            unet_forward(sd, ocfg, lat * s.init_noise_sigma, t, pe, fq=rec)
    scales = rec.calibrated_scales(margin=2.0)
    names = sorted(scales)
    out["lcm4_fp8/scale_names"] = np.array(names)
    out["lcm4_fp8/scales"] = np.array([scales[k] for k in names], dtype=np.float32)
    out["lcm4_fp8/amax"] = np.array([rec.amax[k] for k in names], dtype=np.float32)
    _, _, _, traj = sample_loop(sd, ocfg, LCMOracle(), pe, None, lat, 4, 0.0, lcm_noise=noise, fq=Fp8Emulation(sd, scales=scales))
    keep("lcm4_fp8", traj, 1)
    print(f"lcm4 + lcm4_fp8 done at {time.time() - t0:.0f} s ({len(names)} calibrated tensors, scales "
          f"{min(scales.values()):g} .. {max(scales.values()):g})", flush=True)

    save()
    print(f"wrote {OUT}: {os.path.getsize(OUT) / 1e6:.2f} MB, {len(out)} arrays")


if __name__ == "__main__":
    main()
