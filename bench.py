"""Headline benchmark: 512x512 images/sec at 50 DDIM steps (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch: a full 50-step DDIM sampling loop with CFG
(guidance 7.5) for `--batch` images per GPU (default 8 = BASELINE configs[1], "SD-1.5 512x512, DDIM
50 steps, batch=8 bf16 on 1xMI355X"), inputs resident in HBM.  With N > 1 every rank samples its
own shard of the global batch (weak scaling, no data-path collective except ONE RCCL all-gather of
the final latents per step).  Weights are SD-1.5-shaped synthetic (no checkpoints exist offline).

The JSON line also carries
  roofline     - the dominant kernel (implicit-GEMM 3x3 conv, ~50 % of UNet FLOPs): algorithmic
                 FLOPs per launch / hipEvent-measured launch time vs the 2.5 PFLOP/s bf16 MFMA peak
  cpu_baseline - the CPU oracle (PyTorch fp32 restatement of the reference loop) timed on this
                 box's host cores on a bounded sample (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense, MI355X_MICROARCH.md
MFMA_FP8_PEAK_TFLOPS = 5000.0    # dense (v_mfma_f32_16x16x128_f8f6f4)
HBM_PEAK_GBS = 8000.0
GFLOP_PER_SAMPLE_FWD = 803.3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2, help="timed sampling runs (K)")
    ap.add_argument("--warmup", type=int, default=1, help="untimed sampling runs (W)")
    ap.add_argument("--batch", type=int, default=8, help="images per GPU per run")
    ap.add_argument("--ddim-steps", type=int, default=50)
    ap.add_argument("--scheduler", default="ddim", choices=["ddim", "dpm", "lcm"])
    ap.add_argument("--cache-interval", type=int, default=0, help="DeepCache interval (0 = off)")
    ap.add_argument("--sample-size", type=int, default=64)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp8"],
                    help="operand type of the UNet's conv / FF / QKV contractions (fp8 = e4m3 weights + activations)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end (text encode + loop + VAE decode) report")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short runs of BASELINE configs[2..4] reported under other_configs (N=1 only)")
    ap.add_argument("--no-parity-full-length", action="store_true",
                    help="skip config.parity_full_length (50 batch-1 forwards against the committed oracle fixture): with it a "
                         "rocprofv3 --stats run of this script averages headline and batch-1 launches together")
    ap.add_argument("--cpu-seconds", type=float, default=45.0, help="budget for the CPU baseline sample")
    return ap.parse_args()


def cpu_model_string():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or "unknown"


def cpu_baseline(cfg, sd, args, gpu_traj=None):
    """Oracle timed on the host cores on a bounded sample of configs[0] (B=1, DDIM, CFG): fp32 primary, and a shorter
    fp16 leg (the reference loads torch_dtype=float16 on the CPU too, src/experiments/base_experiment.py:62).  The oracle's
    free-running latents after its n steps also CHECK the GPU path's latents after the same n steps from the same
    inputs (``gpu_traj``: latents after each of the first steps): returned as ``parity``."""
    from oracle.unet import UNetConfig as OC, unet_forward
    from oracle.schedulers import DDIMOracle
    import dataclasses
    ocfg = OC(**dataclasses.asdict(cfg))
    lat0, ctx = parity_inputs(cfg)

    def run(dtype, budget, max_steps):
        w = sd if dtype == torch.float32 else {k: v.to(dtype) for k, v in sd.items()}
        lat, c = lat0.to(dtype), ctx.to(dtype)
        sch = DDIMOracle()
        sch.set_timesteps(args.ddim_steps)
        n_done, t0 = 0, time.time()
        with torch.no_grad():
            for t in sch.timesteps[:max_steps]:
                e = unet_forward(w, ocfg, torch.cat([lat, lat]), t, c).float()
                u, cnd = e.chunk(2)
                lat = sch.step(u + 7.5 * (cnd - u), t, lat.float())[0].to(dtype)
                n_done += 1
                if time.time() - t0 > budget:
                    break
        return n_done, time.time() - t0, lat.float()

    n_done, dt, lat32 = run(torch.float32, args.cpu_seconds, len(gpu_traj) if gpu_traj else args.ddim_steps)
    per_step = dt / n_done
    out = {"value": 1.0 / (per_step * args.ddim_steps), "unit": "images/s", "cores": torch.get_num_threads(),
           "kind": "port", "cpu_model": cpu_model_string(), "dtype": "fp32",
           "sample": f"{n_done} of {args.ddim_steps} DDIM steps (CFG pair UNet forward + step, batch 1, fp32 "
                     f"PyTorch-CPU oracle) in {dt:.1f}s, extrapolated to {args.ddim_steps} steps"}
    # fp16 leg (the reference loads torch_dtype=float16 on the CPU as well).  ATen's half-precision CPU kernels are not
    # vectorised on every host (one 64x64 fp16 conv can take tens of seconds), so it is measured at 16x16 latents -- one DDIM
    # step in fp32 and one in fp16, same weights, each bounded by a deadline the oracle checks between UNet blocks -- and
    # reported as the fp16 : fp32 time ratio applied to the 64x64 fp32 figure above.
    import oracle.unet as ou
    small = dataclasses.replace(cfg, sample_size=16)
    osmall = OC(**dataclasses.asdict(small))
    lat_s, ctx_s = parity_inputs(small)

    def one_step(dtype, budget):
        w = sd if dtype == torch.float32 else {k: v.to(dtype) for k, v in sd.items()}
        sch = DDIMOracle()
        sch.set_timesteps(args.ddim_steps)
        t0 = time.time()
        ou.BLOCKS_DONE, ou.DEADLINE = 0, t0 + budget
        try:
            with torch.no_grad():
                t = sch.timesteps[0]
                e = unet_forward(w, osmall, torch.cat([lat_s, lat_s]).to(dtype), t, ctx_s.to(dtype)).float()
                u, cnd = e.chunk(2)
                sch.step(u + 7.5 * (cnd - u), t, lat_s)
            return time.time() - t0, None
        except TimeoutError:
            return time.time() - t0, ou.BLOCKS_DONE
        finally:
            ou.DEADLINE = None

    try:
        t32, _ = one_step(torch.float32, 20.0)
        t16, cut = one_step(torch.float16, 20.0)
        if cut is None:
            out["fp16"] = {"value": out["value"] * t32 / t16, "unit": "images/s", "fp16_over_fp32_time": t16 / t32,
                           "sample": f"one DDIM step at 16x16 latents: fp32 {t32:.2f}s, fp16 weights + activations {t16:.2f}s; the "
                                     "ratio applied to the 64x64 fp32 figure (what the reference's CPU path would run: torch_dtype=float16)"}
        else:
            out["fp16"] = {"value": None, "upper_bound": out["value"] * t32 / t16, "unit": "images/s",
                           "sample": f"one fp16 DDIM step at 16x16 latents did not finish within {t16:.0f}s ({cut} of 38 UNet blocks; "
                                     f"fp32: {t32:.2f}s): the half-precision ATen CPU kernels are far slower than fp32 on this host"}
    except Exception as e:                                    # an ATen CPU op without a half kernel: say so, do not fail the bench
        out["fp16"] = {"value": None, "error": f"{type(e).__name__}: {e}"[:200]}
    parity = None
    if gpu_traj and n_done <= len(gpu_traj):
        got = gpu_traj[n_done - 1].float().cpu()
        d = got - lat32
        parity = {"against": "fp32 CPU oracle (oracle/: this build's restatement, PARITY UNPINNED -- DESIGN.md 2)",
                  "what": f"free-running latents after the first {n_done} of {args.ddim_steps} DDIM steps, CFG 7.5, batch 1, "
                          f"{cfg.sample_size}x{cfg.sample_size} latents, same seeded inputs and weights",
                  "max_abs": d.abs().max().item(), "rel_l2": (d.norm() / lat32.norm()).item(),
                  "cosine": (got.flatten() @ lat32.flatten() / (got.norm() * lat32.norm())).item()}
    return out, parity


def parity_inputs(cfg):
    g = torch.Generator().manual_seed(29)
    lat = torch.randn((1, 4, cfg.sample_size, cfg.sample_size), generator=g)
    ctx = torch.randn((2, cfg.context_len, cfg.cross_attention_dim), generator=g)
    return lat, ctx


def gpu_free_running(model, cfg, args, dev, n_steps=12):
    """The GPU path on the cpu_baseline sample: latents after each of the first ``n_steps`` DDIM steps (CFG 7.5, batch 1)."""
    from sonicdiffusionbayeslab_amd.registry import schedulers_registry
    from sonicdiffusionbayeslab_amd.schedulers import PNDMConfigStub
    lat, ctx = parity_inputs(cfg)
    s = schedulers_registry["ddim_scheduler"].from_config(PNDMConfigStub().config)
    s.set_timesteps(args.ddim_steps, device=dev)
    model.unet.set_deepcache(-1)
    model.unet.set_context(ctx.to(dev))
    x, traj = lat.to(dev), []
    for t in s._timesteps_list[:n_steps]:
        eps = model.unet.forward_latents(x, 2, float(t))
        x, _ = s.step_fused(eps, 7.5, x, t, cfg=True)
        traj.append(x.clone())
    torch.cuda.synchronize()
    return traj


def full_length_parity(model, cfg, sd, args, dev):
    """The headline's own loop at FULL length -- 50 of 50 DDIM steps, CFG 7.5, 64x64 latents, batch 1 -- on the GPU path, against
    the oracle's trajectory COMMITTED as a fixture (tests/golden/loop_golden_64.npz, generated by
    tests/golden/make_loop_golden.py: data, not code -- no oracle runs here).  The inputs are re-drawn from the fixture's
    seed and must be bit-identical to the stored ones; the fixture must belong to this run's synthetic weights.  Returns
    None with a reason when it does not apply."""
    import hashlib
    import numpy as np
    from sonicdiffusionbayeslab_amd.registry import schedulers_registry
    from sonicdiffusionbayeslab_amd.schedulers import PNDMConfigStub
    path = os.path.join(ROOT, "tests", "golden", "loop_golden_64.npz")
    if not (os.path.exists(path) and cfg.sample_size == 64 and args.ddim_steps == 50 and args.dtype == "bf16"):
        return {"skipped": "needs the committed fixture, 64x64 latents, 50 DDIM steps, bf16"}
    z = np.load(path)
    h = hashlib.sha256()
    for k in ("conv_in.weight", "mid_block.resnets.0.conv1.weight", "up_blocks.3.attentions.2.transformer_blocks.0.ff.net.2.weight",
              "conv_out.bias"):
        h.update(sd[k].detach().float().contiguous().numpy().tobytes())
    if h.hexdigest()[:16] != str(z["weights_fingerprint"]):
        return {"skipped": "the fixture was generated for other synthetic weights"}
    g = torch.Generator().manual_seed(29)                      # tests/util.py::synth_inputs(cfg, 1, seed=29)
    lat = torch.randn((1, cfg.in_channels, 64, 64), generator=g)
    pe = torch.randn((1, cfg.context_len, cfg.cross_attention_dim), generator=g)
    ne = torch.randn((1, cfg.context_len, cfg.cross_attention_dim), generator=g)
    if not torch.equal(lat, torch.as_tensor(z["ddim50/init"])):
        return {"skipped": "the re-drawn inputs differ from the fixture's"}
    s = schedulers_registry["ddim_scheduler"].from_config(PNDMConfigStub().config)
    s.set_timesteps(50, device=dev)
    model.unet.set_deepcache(-1)
    model.unet.set_context(torch.cat([ne, pe]).to(dev))
    x, rows = lat.to(dev), {}
    for i, t in enumerate(s._timesteps_list):
        eps = model.unet.forward_latents(x, 2, float(t))
        x, _ = s.step_fused(eps, 7.5, x, t, cfg=True)
        if (i + 1) % 10 == 0:
            ref = torch.as_tensor(z[f"ddim50/step{i + 1}"])
            got = x.float().cpu()
            d = got - ref
            rows[f"after_{i + 1}_steps"] = {"rel_l2": (d.norm() / ref.norm()).item(), "max_abs": d.abs().max().item(),
                                            "cosine": (got.flatten() @ ref.flatten() / (got.norm() * ref.norm())).item()}
    return {"against": "the fp32 CPU oracle's trajectory committed as tests/golden/loop_golden_64.npz (oracle/ = this build's "
                       "restatement: PARITY UNPINNED -- DESIGN.md 2)",
            "what": "free-running latents of the headline loop at full length: 50 of 50 DDIM steps, CFG 7.5, 64x64, batch 1", **rows}


def end_to_end(model, args, dev):
    """next to the loop-only number, the time of one whole call from prompt STRINGS to decoded
    512x512 images -- CLIP text encoding (libsdhip text tower, seeded synthetic ViT-L/14-shaped weights, merge-free
    byte-level tokenizer: no CLIP checkpoint exists offline), the sampling loop, and the VAE decode (libsdhip
    AutoencoderKL decoder, synthetic weights).  Not part of `value`; the reference's timer covers the loop only."""
    import json as _json
    from sonicdiffusionbayeslab_amd.clip import (ClipBpeTokenizer, ClipPromptEncoder, ClipTextConfig, HipClipTextModel,
                                                 make_synthetic_clip_state_dict)
    tok = ClipBpeTokenizer.byte_level()
    ccfg = ClipTextConfig(vocab_size=len(tok.encoder), bos_token_id=tok.bos_token_id, eos_token_id=tok.eos_token_id,
                          pad_token_id=tok.pad_token_id)
    enc = ClipPromptEncoder(tok, HipClipTextModel(ccfg, make_synthetic_clip_state_dict(ccfg, seed=777), device=str(dev)))
    model.text_encoder = enc
    root = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(root, "data", "dataset", "img2annotations_test.json")) as f:
        prompts = list(_json.load(f).values())[: args.batch]
    guidance = 0.0 if args.scheduler == "lcm" else 7.5
    g = torch.Generator().manual_seed(29)
    best = None
    for _ in range(2):                                   # first pass builds the VAE / warms up
        torch.cuda.synchronize()
        t0 = time.time()
        pe = enc(prompts)
        ne = enc([""] * len(prompts)) if guidance > 1 else None
        torch.cuda.synchronize()
        t1 = time.time()
        out, loop_s, _ = model(prompt_embeds=pe, negative_prompt_embeds=ne, num_inference_steps=args.ddim_steps,
                               guidance_scale=guidance, generator=g, output_type="pt", collect_x0=False)
        torch.cuda.synchronize()
        t2 = time.time()
        best = {"s_per_image": (t2 - t0) / len(prompts), "text_encode_ms": 1e3 * (t1 - t0), "loop_s": loop_s,
                "vae_decode_and_postprocess_ms": 1e3 * (t2 - t1 - loop_s), "images": list(out.images.shape),
                "note": "prompt strings -> [B,3,512,512]; synthetic CLIP / VAE weights, byte-level tokenizer"}
    return best


PROFILE_ROUND = "round5"
TRAFFIC_FILE = f"profiles/{PROFILE_ROUND}_conv_traffic.json"
PMC_FILE = f"profiles/{PROFILE_ROUND}_pmc_forward.json"
TRAFFIC_SOURCE = (f"{TRAFFIC_FILE} (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/run_profiles_r5.sh traffic over "
                  "tools/forward_once.py at this bench shape; NOT measured in this run; nulled when the kernel sources "
                  "have changed since that profile was taken)")
PMC_SOURCE = (f"{PMC_FILE} (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES ... GRBM_GUI_ACTIVE pass of tools/run_profiles_r5.sh fwd; "
              "NOT measured in this run; nulled when the kernel sources have changed since)")


def kernel_sources_sha16():
    """Fingerprint of the kernel sources a committed PMC profile belongs to (the GPU box has no .git): sha256 over the
    kernels and everything that decides which of them run with what split / tile (the plan builder in unet.hip included) --
    every file of csrc/ -- first 16 hex digits.  tools/pmc_*.py stamp it into the profile JSON and the readers below refuse a
    profile whose stamp is not the current one (a stale counter is worse than none)."""
    import hashlib
    h = hashlib.sha256()
    root = os.path.join(ROOT, "sonicdiffusionbayeslab_amd", "csrc")
    for f in [os.path.join(root, n) for n in sorted(os.listdir(root)) if n.endswith((".hip", ".h"))]:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _profile(path):
    """(json, None) of a committed profile if it carries the current kernel fingerprint, else (None, reason)."""
    try:
        with open(os.path.join(ROOT, path)) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None, f"{path} absent"
    have = d.get("kernel_sources_sha16")
    if have != kernel_sources_sha16():
        return None, f"{path} is stale (taken on kernel sources {have}, current {kernel_sources_sha16()})"
    return d, None


def conv_mfma_busy():
    """MFMA-busy fraction of the dominant kernel from the committed PMC pass: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x
    shader cycles of the dispatch); (None, reason) if the profile is absent or stale."""
    d, why = _profile(PMC_FILE)
    if d is None:
        return None, why
    rows = [v for k, v in d.items() if k.startswith("conv_halo_kernel<0") and isinstance(v, dict)]     # the bf16 product kernel
    try:
        return float(max(rows, key=lambda v: v["launches"])["mfma_busy_frac"]), None
    except (KeyError, ValueError, TypeError):
        return None, f"{PMC_FILE} has no conv_halo_kernel<0, ...> row"


def conv_traffic_bytes():
    """HBM bytes per launch of the dominant kernel from the committed PMC passes ((2*FETCH_SIZE + WRITE_SIZE)*1024
    averaged over the kernel's launches of one forward at this bench shape, MI355X_MICROARCH.md's gfx950 correction).
    PMC counters cannot be read live inside this process; (None, reason) if the profile is absent or stale."""
    d, why = _profile(TRAFFIC_FILE)
    if d is None:
        return None, why
    try:
        return float(d["hbm_bytes_per_launch"]), None
    except (KeyError, ValueError):
        return None, f"{TRAFFIC_FILE} has no hbm_bytes_per_launch"


def conv_stream_ceiling(ub, dev):
    """Measured ceiling of the dominant kernel's own MFMA stream: the halo conv at three of the forward's shapes with its
    LDS-DMA, tap barriers and fragment reads removed (sd_op_conv3x3_ablate mode 8; results are wrong by design) -- every
    v_mfma of the real kernel, nothing else -- against the unmodified kernel on the same operands.  The nominal 2.5 PFLOP/s is
    the peak at 2.4 GHz; under a sustained MFMA load the chip holds ~1.6 GHz (MI355X_MICROARCH.md 'DVFS give-back'), and
    this is what that leaves.  hipEvent timing on the launch stream, random bf16 operands."""
    from sonicdiffusionbayeslab_amd import _lib
    # the ablation kernels live in the SD_ABLATE build of the library only (a second, independent instance: nothing of
    # the product path below runs on it)
    lib = _lib.load_ablate()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device=dev).manual_seed(7)
    tot = {0: 0.0, 8: 0.0}
    flops = 0.0
    for res, c in ((64, 320), (32, 640), (16, 1280)):
        x = torch.randn((ub, res, res, c), device=dev, generator=g).to(torch.bfloat16)
        w = torch.randn((c, c // 64, 9, 64), device=dev, generator=g).to(torch.bfloat16)
        y = torch.empty((ub, res, res, c), device=dev, dtype=torch.bfloat16)
        flops += 2.0 * ub * res * res * c * 9 * c
        for mode in (0, 8):
            f = lambda: _lib.check(lib.sd_op_conv3x3_ablate(st, x.data_ptr(), w.data_ptr(), y.data_ptr(), ub, res, res, c, c, mode), lib=lib)
            for _ in range(3):
                f()
            s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s0.record()
            for _ in range(10):
                f()
            s1.record()
            torch.cuda.synchronize()
            tot[mode] += s0.elapsed_time(s1) / 10 * 1e-3
    return {"kernel_tflops": flops / tot[0] / 1e12, "bare_mfma_stream_tflops": flops / tot[8] / 1e12,
            "kernel_over_stream": tot[8] / tot[0],
            "what": "conv_halo_kernel at 64x64x320, 32x32x640, 16x16x1280 (UNet batch of this run), unmodified vs its bare MFMA "
                    "stream (no LDS-DMA, barriers or fragment reads: sd_op_conv3x3_ablate 8 of libsdhip_ablate.so); measured in this run, at this "
                    "batch: launch, fill and drain of 1-2 items per CU are in both numbers (a pure MFMA loop sustains 2.03 PFLOP/s and "
                    "the ablated kernel 1.8-1.96 at many items per CU: profiles/round5_notes.md 9)"}


def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: start N fresh ranks with torch.distributed.run (one process
    per GPU, RCCL).  Runs BEFORE this process has touched the GPU; the parent only waits and passes the exit code on."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def timed_runs(model, args_like, dev, rank, world, sdist):
    """W warm-up + K timed sampling runs of one workload; returns (images/s, ms per run, loop seconds, final latents).
    The timed region is bracketed by a barrier + device synchronisation on both sides, MAX over ranks."""
    from sonicdiffusionbayeslab_amd.deepcache import DeepCacheSDHelper
    from sonicdiffusionbayeslab_amd.registry import schedulers_registry
    a = args_like
    cfg = model.unet_config
    name = {"ddim": "ddim_scheduler", "dpm": "dpm_solver_scheduler", "lcm": "lcm_scheduler"}[a.scheduler]
    from sonicdiffusionbayeslab_amd.schedulers import PNDMConfigStub
    model.scheduler = schedulers_registry[name].from_config(PNDMConfigStub().config)   # the checkpoint's scheduler config
    helper = None
    if a.cache_interval > 0:
        helper = DeepCacheSDHelper(pipe=model)
        helper.set_params(cache_interval=a.cache_interval, cache_branch_id=0)
        helper.enable()
    guidance = 0.0 if a.scheduler == "lcm" else 7.5
    B = a.batch
    gb = B * world
    lo, hi = sdist.shard_range(gb, rank, world)
    lat_all = sdist.global_latents(gb, 4, cfg.sample_size, seed=29)
    g = torch.Generator().manual_seed(30)
    pe_all = torch.randn((gb, cfg.context_len, cfg.cross_attention_dim), generator=g)
    ne = torch.randn((1, cfg.context_len, cfg.cross_attention_dim), generator=g)
    lat = lat_all[lo:hi].to(dev)
    pe = pe_all[lo:hi].to(dev)
    neg = ne.repeat(hi - lo, 1, 1).to(dev)

    def one_run():
        out, secs, _ = model(prompt_embeds=pe, negative_prompt_embeds=neg, latents=lat,
                             num_inference_steps=a.ddim_steps, guidance_scale=guidance,
                             output_type="latent", collect_x0=False)
        return sdist.gather_latents(out.images, world, gb), secs

    for _ in range(a.warmup):
        one_run()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.time()
    loop_secs = 0.0
    for _ in range(a.steps):
        final, secs = one_run()
        loop_secs += secs
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    elapsed = time.time() - t0
    if world > 1:
        backend = torch.distributed.get_backend()
        tmax = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tmax.item())
    assert torch.isfinite(final).all(), "non-finite latents"
    if helper is not None:
        helper.disable()
    return gb * a.steps / elapsed, 1e3 * elapsed / a.steps, loop_secs, lat, guidance


def other_configs(model, args, dev, sdist, sd):
    """Short driver-visible runs of the other single-GPU-sized BASELINE configs (per-GPU batch of each):
    configs[2] DPM-Solver++ 20 steps batch 32; configs[3] DeepCache N=3 on DDIM-50, 16/GPU; configs[4] LCM 4 steps,
    32/GPU (bf16 weights here).  1 warm-up + 2 timed runs each; images/s on this one GPU."""
    import types
    out = []
    for label, sched, steps, batch, interval in (
            ("configs[2]: DPM-Solver++ (order 2) 20 steps, CFG 7.5, batch 32", "dpm", 20, 32, 0),
            ("configs[3] per-GPU share: DeepCache N=3 branch 0, DDIM 50 steps, CFG 7.5, batch 16", "ddim", 50, 16, 3),
            ("configs[4] per-GPU share: LCM 4 steps, no CFG, batch 32", "lcm", 4, 32, 0)):
        a = types.SimpleNamespace(scheduler=sched, ddim_steps=steps, batch=batch, cache_interval=interval, steps=2, warmup=1)
        ips, ms, loop_secs, _, _ = timed_runs(model, a, dev, 0, 1, sdist)
        out.append({"workload": label, "value": ips, "unit": "images/s", "ms_per_step": ms, "steps": 2, "warmup": 1,
                    "dtype": "bf16", "loop_only_s_per_image": loop_secs / (2 * batch)})
    # configs[4] as BASELINE names it: fp8 MFMA weights (a second weight replica, e4m3 conv / FF / QKV contractions)
    from sonicdiffusionbayeslab_amd.models import StableDiffusionModel
    m8 = StableDiffusionModel(unet_config=model.unet_config, state_dict=dict(sd), source="synthetic(seed=1234)",
                              weight_dtype="fp8").to(dev)
    a = types.SimpleNamespace(scheduler="lcm", ddim_steps=4, batch=32, cache_interval=0, steps=2, warmup=1)
    ips, ms, loop_secs, lat8, _ = timed_runs(m8, a, dev, 0, 1, sdist)
    prof = m8.unet.forward_profiled(lat8, 32, 501.0)
    prof = m8.unet.forward_profiled(lat8, 32, 501.0)
    c8, g8 = prof["conv3x3_fp8"], prof["gemm_fp8"]
    # where the fp8 forward's time goes, next to the bf16 model's forward at the SAME batch (32, no CFG): the accounting behind
    # "what would fp8 have to remove to pay more" 
    model.unet.set_deepcache(-1)
    model.unet.set_context(torch.randn((32, model.unet_config.context_len, model.unet_config.cross_attention_dim), device=dev))
    p16 = model.unet.forward_profiled(lat8, 32, 501.0)
    p16 = model.unet.forward_profiled(lat8, 32, 501.0)
    brk = lambda pr: {k: {"ms": round(v["ms"], 3), "launches": v["launches"]} for k, v in pr.items() if v["launches"]}
    fwd_breakdown = {"fp8_e4m3": brk(prof), "bf16": brk(p16), "fp8_total_ms": round(sum(v["ms"] for v in prof.values()), 3),
                     "bf16_total_ms": round(sum(v["ms"] for v in p16.values()), 3), "unet_batch": 32}
    # quality next to the speed: one UNet forward (batch 2, t = 499, same latents / prompt embeddings) of the fp8 model against
    # the bf16 model of this run -- the error of the e4m3 scheme (calibrated per-tensor scales) on these synthetic weights
    gq = torch.Generator().manual_seed(31)
    lat_q = torch.randn((1, 4, model.unet_config.sample_size, model.unet_config.sample_size), generator=gq).to(dev)
    ctx_q = torch.randn((2, model.unet_config.context_len, model.unet_config.cross_attention_dim), generator=gq).to(dev)
    model.unet.set_deepcache(-1); model.unet.set_context(ctx_q)
    e16 = model.unet.forward_latents(lat_q, 2, 499.0).double()
    m8.unet.set_context(ctx_q)
    e8 = m8.unet.forward_latents(lat_q, 2, 499.0).double()
    fp8_err = {"rel_l2": float((e8 - e16).norm() / e16.norm()),
               "cosine": float((e8.flatten() @ e16.flatten()) / (e8.norm() * e16.norm())),
               "what": "noise prediction of one UNet forward (batch 2, t = 499), fp8-e4m3 model vs the bf16 model, same inputs"}
    scales = m8.unet.fp8_scales(with_amax=True)
    out.append({"workload": "configs[4] per-GPU share: LCM 4 steps, no CFG, batch 32, fp8-e4m3 weights + activations in the "
                            "resnet conv / FF / QKV / proj_in contractions", "value": ips, "unit": "images/s",
                "ms_per_step": ms, "steps": 2, "warmup": 1, "dtype": "fp8_e4m3", "loop_only_s_per_image": loop_secs / (2 * 32),
                "fp8_forward_vs_bf16": fp8_err, "forward_breakdown_ms": fwd_breakdown,
                "fp8_activation_scales": {"tensors": len(scales), "calibrated": sum(1 for _, a in scales.values() if a > 0),
                                          "min_scale": min(s for s, _ in scales.values()) if scales else None,
                                          "max_scale": max(s for s, _ in scales.values()) if scales else None,
                                          "max_amax": max(a for _, a in scales.values()) if scales else None},
                "roofline": {"bound": "mfma", "kernel": "conv_halo_kernel<fp8>", "peak": MFMA_FP8_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "achieved": c8["flops"] / (c8["ms"] * 1e-3) / 1e12,
                             "frac": c8["flops"] / (c8["ms"] * 1e-3) / 1e12 / MFMA_FP8_PEAK_TFLOPS,
                             "launches_per_forward": c8["launches"], "avg_launch_ms": c8["ms"] / max(c8["launches"], 1),
                             "gemm_fp8_tflops": g8["flops"] / (g8["ms"] * 1e-3) / 1e12 if g8["ms"] else None}})
    del m8
    torch.cuda.empty_cache()
    return out


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    from sonicdiffusionbayeslab_amd import dist as sdist
    # SD_BENCH_BACKEND=gloo is a single-GPU rehearsal of the N>1 code path (several ranks share cuda:0);
    # the real run uses RCCL ("nccl"), one rank per GPU
    backend = os.environ.get("SD_BENCH_BACKEND", "nccl")
    rank, local_rank, world = sdist.init_process_group(backend)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    dev_index = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    from sonicdiffusionbayeslab_amd.models import StableDiffusionModel
    from sonicdiffusionbayeslab_amd.weights import UNetConfig, make_synthetic_state_dict

    cfg = UNetConfig(sample_size=args.sample_size)
    sd = make_synthetic_state_dict(cfg, seed=1234)
    model = StableDiffusionModel(unet_config=cfg, state_dict=dict(sd), source="synthetic(seed=1234)", weight_dtype=args.dtype)
    model.to(dev)
    B = args.batch
    gb = B * world
    # synthetic inputs, resident in HBM before the timed region; global draw sliced per rank
    value, ms_per_step, loop_secs, lat, guidance = timed_runs(model, args, dev, rank, world, sdist)
    elapsed = ms_per_step * args.steps / 1e3
    is_headline = (args.scheduler == "ddim" and args.ddim_steps == 50 and B == 8 and not args.cache_interval
                   and args.sample_size == 64 and args.dtype == "bf16")
    dtype_name = "bf16" if args.dtype == "bf16" else "fp8_e4m3"

    res = {
        "metric": f"512x512 images/sec at {args.ddim_steps} {args.scheduler.upper()} steps",   # default: BASELINE's metric
        "value": value, "unit": "images/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": dtype_name, "data": "synthetic",
        "config": {"workload": f"SD-1.5 512x512 {args.scheduler.upper()} {args.ddim_steps} steps, CFG {guidance}, "
                               f"batch={B}/GPU" + (" (BASELINE configs[1])" if is_headline else "")
                               + (f", DeepCache N={args.cache_interval}" if args.cache_interval else ""),
                   "global_batch": gb, "sample_size": cfg.sample_size, "parallelism": f"batch-shard x{world}",
                   "weights": "SD-1.5-shaped synthetic seed 1234", "loop_only_s_per_image": loop_secs / (args.steps * B)},
    }
    ub = (2 if guidance > 1 else 1) * B
    nfwd = args.ddim_steps
    res["config"]["unet_tflops_effective"] = (GFLOP_PER_SAMPLE_FWD * (cfg.sample_size / 64.0) ** 2 * ub * nfwd * args.steps / elapsed / 1e3
                                              if not args.cache_interval else None)

    if rank == 0 and not args.no_roofline:
        prof = model.unet.forward_profiled(lat, ub, 501.0)       # warm
        prof = model.unet.forward_profiled(lat, ub, 501.0)
        fp8 = args.dtype == "fp8"
        c3 = prof["conv3x3_fp8"] if fp8 else prof["conv3x3"]
        peak = MFMA_FP8_PEAK_TFLOPS if fp8 else MFMA_BF16_PEAK_TFLOPS
        ach = c3["flops"] / (c3["ms"] * 1e-3) / 1e12
        res["roofline"] = {"bound": "mfma",
                           "kernel": ("conv_halo_kernel<fp8> (3x3 resnet convs on v_mfma_f32_16x16x128_f8f6f4, LDS-resident input halo)"
                                      if fp8 else
                                      "conv_halo_kernel<0,0,8,0> (3x3 conv, LDS-resident input halo, 9-tap mode; the stride-1 convs of a forward: 44 of its "
                                      "50 conv launches -- the 16x16->32x32 and 32x32->64x64 sub-pixel upsamplers run on the same kernel's 4-tap mode "
                                      "(kernel_breakdown.conv3x3_halo_subpix), the 3 stride-2 convs and the 8x8->16x16 upsampler on the implicit-GEMM "
                                      "kernel (kernel_breakdown.conv3x3_gemm))"),
                           "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                           "frac": ach / peak, "traffic": None if fp8 else conv_traffic_bytes()[0],
                           "traffic_source": None if fp8 else (conv_traffic_bytes()[1] or TRAFFIC_SOURCE),
                           "mfma_busy": None if fp8 else conv_mfma_busy()[0],
                           "mfma_busy_source": None if fp8 else (conv_mfma_busy()[1] or PMC_SOURCE),
                           "kernel_sources_sha16": kernel_sources_sha16(),
                           "mfma_stream_ceiling": None if fp8 else conv_stream_ceiling(ub, dev),
                           "launches_per_forward": c3["launches"], "avg_launch_ms": c3["ms"] / max(c3["launches"], 1),
                           "flops_per_launch": c3["flops"] / max(c3["launches"], 1)}
        tot = sum(v["ms"] for v in prof.values())
        res["kernel_breakdown"] = {
            k: {"ms": round(v["ms"], 4), "launches": v["launches"], "share": round(v["ms"] / tot, 4),
                "tflops": (round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) if v["flops"] and v["ms"] else None),
                "alg_GBs": (round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["bytes"] and v["ms"] else None)}
            for k, v in prof.items()}
    if rank == 0 and world == 1 and is_headline and not args.no_other_configs:
        res["other_configs"] = other_configs(model, args, dev, sdist, sd)
        args_sched = {"ddim": "ddim_scheduler", "dpm": "dpm_solver_scheduler", "lcm": "lcm_scheduler"}[args.scheduler]
        from sonicdiffusionbayeslab_amd.registry import schedulers_registry
        from sonicdiffusionbayeslab_amd.schedulers import PNDMConfigStub
        model.scheduler = schedulers_registry[args_sched].from_config(PNDMConfigStub().config)
    if rank == 0 and world == 1 and not args.no_e2e and args.sample_size == 64:
        res["config"]["end_to_end"] = end_to_end(model, args, dev)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        traj = gpu_free_running(model, cfg, args, dev) if args.dtype == "bf16" and args.scheduler == "ddim" else None
        res["cpu_baseline"], parity = cpu_baseline(cfg, sd, args, traj)
        res["config"]["parity"] = parity
    if rank == 0 and world == 1 and is_headline and not args.no_parity_full_length:
        res["config"]["parity_full_length"] = full_length_parity(model, cfg, sd, args, dev)
    # the SD_* switches change kernels, split factors and rounding points: a line's environment is part of its result
    res["config"]["sd_env"] = {k: v for k, v in sorted(os.environ.items()) if k.startswith("SD_")}
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
