"""CPU: the N>1 path (batch shard + ONE all-gather of final latents) with world_size 2 over gloo --
the collective itself, and the harness (`BaseMethod.generate`) sharding prompt batches around a stub
pipeline (the real pipeline needs the GPU: tests/test_dist_gpu.py)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, gb, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from sonicdiffusionbayeslab_amd.dist import gather_latents, global_latents, shard_range
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lat = global_latents(gb, 4, 8, seed=29)
    lo, hi = shard_range(gb, rank, world)
    local = lat[lo:hi] * 2.0 + 1.0          # stand-in for the per-rank sampling result
    full = gather_latents(local, world, gb)
    q.put((rank, full.numpy()))          # by value: a torch tensor travels as an fd the consumer must fetch while this rank lives
    dist.barrier()
    dist.destroy_process_group()


def _run(gb):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, gb, q)) for r in range(2)]
    [p.start() for p in ps]
    outs = {r: torch.from_numpy(a) for r, a in (q.get(timeout=120) for _ in range(2))}
    [p.join(60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    return outs


def test_gather_equals_single_process_even_and_ragged():
    from sonicdiffusionbayeslab_amd.dist import global_latents
    for gb in (8, 5):
        outs = _run(gb)
        want = global_latents(gb, 4, 8, seed=29) * 2.0 + 1.0
        assert torch.equal(outs[0], want) and torch.equal(outs[1], want)


# ---------------------------------------------------------------------------------------------------
# BaseMethod.generate under 2 ranks: shard -> per-rank pipeline call -> ONE gather (stub pipeline on CPU)
# ---------------------------------------------------------------------------------------------------
class _StubOut:
    def __init__(self, images):
        self.images = images


class _StubPipeline:
    """Deterministic per-image function of (prompt, initial latent, LCM noise): what a sampler is to the harness."""
    weights_source = "stub"
    num_timesteps = 3

    def __init__(self, lcm):
        from sonicdiffusionbayeslab_amd.weights import UNetConfig
        from sonicdiffusionbayeslab_amd.schedulers import SchedulerConfig
        self.unet_config = UNetConfig(sample_size=8)
        self.scheduler = type("S", (), {})()
        self.scheduler.config = SchedulerConfig(timestep_scaling=10.0) if lcm else SchedulerConfig()
        self.calls = []

    def to(self, device):
        return self

    def __call__(self, prompts, num_inference_steps=3, guidance_scale=7.5, generator=None, output_type="latent",
                 latents=None, **kw):
        from sonicdiffusionbayeslab_amd import dist as sdist
        n = len(prompts)
        if latents is None:
            latents = sdist.randn((n, 4, 8, 8), generator)           # what prepare_latents does
        out = latents.clone()
        lcm = "timestep_scaling" in self.scheduler.config
        for i in range(num_inference_steps - 1 if lcm else 0):       # a stochastic sampler: one Gaussian per step
            z = sdist.randn((n, 4, 8, 8), generator)                 # (LCMScheduler / sde-DPM-Solver step_fused)
            out = out * 0.5 + z
        key = torch.tensor([float(sum(map(ord, p)) % 97) for p in prompts]).view(n, 1, 1, 1)
        self.calls.append(n)
        return _StubOut(out + key), 0.25 + 0.01 * n, [out[0:1]]


def _harness_worker(rank, world, port, lcm, nprompts, batch, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), SD_DIST_BACKEND="gloo")
    out, t, calls = _harness_run(lcm, nprompts, batch)
    q.put((rank, out.numpy(), t, calls))         # by value (see _worker)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _harness_run(lcm, nprompts, batch=5):
    from sonicdiffusionbayeslab_amd.config import _wrap
    from sonicdiffusionbayeslab_amd.experiments.base_experiment import BaseMethod
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    class M(BaseMethod):
        def setup_model(self):
            self.model = _StubPipeline(lcm)

        def setup_scheduler(self, **kw):
            pass

        def run_experiment(self):
            pass

    conf = _wrap({"experiment_name": "stub", "experiment": {"method": "stub", "seed": 29},
                  "dataset": {"img_dataset": "", "prompts": os.path.join(root, "data", "dataset", "img2annotations_test.json")},
                  "inference": {"batch_size": batch, "batch_count": (nprompts + batch - 1) // batch, "output_type": "latent"}})
    m = M(conf)
    m.test_dataset.image_files = m.test_dataset.image_files[:nprompts]
    images, _ = m.generate(m.test_dataset.batches(batch), 3, batch, guidance_scale=0.0 if lcm else 7.5)
    return torch.stack(images), float(m.time_metric.compute()), list(m.model.calls)


def _harness_world(world, lcm, nprompts, batch=5):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_harness_worker, args=(r, world, port, lcm, nprompts, batch, q)) for r in range(world)]
    [p.start() for p in ps]
    outs = {r[0]: (torch.from_numpy(r[1]),) + tuple(r[2:]) for r in (q.get(timeout=300) for _ in range(world))}
    [p.join(60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    return outs


def _harness_world2(lcm, nprompts, batch=5):
    return _harness_world(2, lcm, nprompts, batch)


def test_generate_shards_batches_and_gathers_like_single_process():
    # even split; a stochastic sampler (a Gaussian per step: LCM / sde-DPM-Solver) over ragged batches; a rank with 0
    # prompts; a stochastic sampler with a prompt-less rank in EVERY batch (its generator must follow rank 0's)
    for lcm, nprompts, batch in ((False, 10, 5), (True, 7, 5), (False, 6, 5), (True, 6, 5), (True, 3, 1)):
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            os.environ.pop(k, None)
        want, t1, calls1 = _harness_run(lcm, nprompts, batch)       # world 1, this process
        outs = _harness_world2(lcm, nprompts, batch)
        for r in (0, 1):
            got, t2, calls = outs[r]
            assert torch.equal(got, want), (lcm, nprompts, r)
            assert sum(calls) + sum(outs[1 - r][2]) == nprompts
        # the slowest rank's loop time is what every rank accumulates
        assert outs[0][1] == outs[1][1] and outs[0][1] <= t1


def test_generate_at_the_world_sizes_and_global_batches_of_baseline_configs_3_and_4():
    """BASELINE configs[3] (global batch 64 over 4 GPUs, DDIM + DeepCache: a deterministic sampler) and configs[4] (global
    batch 256 over 8 GPUs, LCM: a Gaussian per step from the shared generator), plus a ragged 250 over 8 (ranks 0-1 hold 32
    images, the rest 31), through ``BaseMethod.generate`` (src/experiments/base_experiment.py:122-163) with the stub
    pipeline: every rank ends with the single-process result bit for bit, every image computed exactly once.  World 4 and 8
    on gloo here; RCCL at N > 1 has never run (no multi-GPU box in any round so far)."""
    for world, lcm, nprompts, batch in ((4, False, 64, 64), (8, True, 256, 256), (8, True, 250, 250), (8, False, 5, 5)):
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            os.environ.pop(k, None)
        want, t1, _ = _harness_run(lcm, nprompts, batch)
        outs = _harness_world(world, lcm, nprompts, batch)
        assert sorted(outs) == list(range(world))
        per_rank = [sum(outs[r][2]) for r in range(world)]
        assert sum(per_rank) == nprompts and max(per_rank) - min(per_rank) <= 1, per_rank      # contiguous, balanced shards
        for r in range(world):
            assert torch.equal(outs[r][0], want), (world, lcm, nprompts, r)
            assert outs[r][1] == outs[0][1]
