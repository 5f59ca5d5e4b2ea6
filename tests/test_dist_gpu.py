"""GPU: the product's N>1 path end to end -- two ranks (gloo rendezvous, both on cuda:0: one-GPU box), the real
pipeline on a tiny UNet, prompt batches sharded inside ``BaseMethod.generate`` and gathered by ONE collective --
against the same harness run as a single process.

Tolerance: the inputs of every image are bit-identical for every world size (global draws from the shared CPU
generator); outputs agree to rel-L2 <= 5e-3, not bit for bit, because the split-K factors of the GEMM / conv kernels
depend on the per-rank UNet batch (fp32 partial sums are grouped differently before the bf16 rounding)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

SHARD_TOL = 5e-3


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _run_method(config_name, nprompts, batch, overrides=None):
    """`methods_registry[...]` from its YAML on a 16x16-latent SD-1.5-width UNet; returns gathered latents."""
    from sonicdiffusionbayeslab_amd import models as M
    from sonicdiffusionbayeslab_amd.config import load_config
    from sonicdiffusionbayeslab_amd.registry import methods_registry
    from sonicdiffusionbayeslab_amd.weights import UNetConfig, make_synthetic_state_dict
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.chdir(root)
    # two levels of the SD-1.5 widths: this file tests the sharding, not the kernels (a fifth of the parameters to generate,
    # pack and upload in every process)
    ucfg = UNetConfig(sample_size=16, block_out_channels=(320, 640), attn_levels=(True, False))
    sd = make_synthetic_state_dict(ucfg, seed=1234)
    M.StableDiffusionModel.from_pretrained = classmethod(
        lambda c, *a, **k: c(unet_config=ucfg, state_dict=dict(sd), source="synthetic(seed=1234) tiny"))
    conf = load_config(os.path.join(root, "configs", config_name))
    conf.inference.batch_size = batch
    conf.inference.batch_count = (nprompts + batch - 1) // batch
    conf.inference.output_type = "latent"
    for k, v in (overrides or {}).items():
        conf.experiment_params[k] = v
    m = methods_registry[conf.experiment.method](conf)
    m.test_dataset.image_files = m.test_dataset.image_files[:nprompts]
    steps = 4
    gs = getattr(m, "guidance_scale", 7.5)
    m.model.to(m.device)
    images, _ = m.generate(m.test_dataset.batches(batch), steps, batch, guidance_scale=gs)
    return torch.stack(images), float(m.time_metric.compute())


def _worker(rank, world, port, cases, q, backend="gloo", force=False):
    """One rank: every case in turn on ONE process group (a case = (config, prompts, batch, overrides))."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), SD_DIST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if force:
        os.environ["SD_DIST_FORCE_INIT"] = "1"
    import torch.distributed as dist
    for idx, (config_name, nprompts, batch, overrides) in enumerate(cases):
        out, t = _run_method(config_name, nprompts, batch, overrides)
        assert dist.is_initialized() and dist.get_backend() == backend and dist.get_world_size() == world
        q.put((rank, idx, out.cpu().numpy(), t))       # by value: no fd hand-off that needs this rank alive
    dist.barrier()
    dist.destroy_process_group()


SDE = {"algorithm_type": "sde-dpmsolver++"}       # a Gaussian per step from the shared generator (src/schedulers.py:134-147)
CASES = [("ddim_config.yaml", 5, 5, None),                   # CFG, ragged 3 + 2
         ("consistency_model_config.yaml", 6, 4, None),      # LCM noise, 2 batches
         ("dpm_solver_config.yaml", 4, 3, SDE)]              # stochastic DPM-Solver: ragged 2 + 1, then a batch whose rank 1 is empty


@pytest.fixture(scope="module")
def world2_runs():
    """The single-process references of every case (this process), then ONE pair of ranks that runs every case on one
    process group: three processes touch the GPU instead of the five-per-case of round 3 (GPU-box process guard, run time)."""
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SD_DIST_FORCE_INIT"):
        os.environ.pop(k, None)
    want = [_run_method(*c) for c in CASES]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, CASES, q)) for r in range(2)]
    [p.start() for p in ps]
    got = {}
    for _ in range(2 * len(CASES)):
        rank, idx, out, t = q.get(timeout=600)
        got[(rank, idx)] = (torch.from_numpy(out), t)
    [p.join(120) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    return want, got


@pytest.mark.parametrize("idx", range(len(CASES)), ids=[c[0] for c in CASES])
def test_world2_sharded_generate_matches_single_process(world2_runs, idx):
    from tests.util import rel_l2
    want, got = world2_runs
    config_name, nprompts = CASES[idx][0], CASES[idx][1]
    (o0, t0), (o1, t1) = got[(0, idx)], got[(1, idx)]
    assert torch.equal(o0, o1), "both ranks must hold the same gathered batch"
    err = rel_l2(o0, want[idx][0])
    print(f"{config_name}: world-2 vs world-1 rel-L2 {err:.3e} over {nprompts} images")
    assert o0.shape == want[idx][0].shape and err < SHARD_TOL
    assert t0 > 0 and t0 == t1


def test_rccl_world1_rehearsal_of_the_gather_path():
    """RCCL itself (backend "nccl"), once, before an 8-GPU node runs it: a fresh child process with WORLD_SIZE 1 and
    SD_DIST_FORCE_INIT=1 creates the communicator bound to its GPU (``device_id``), shards (trivially) and goes through
    ``gather_latents`` -> ``all_gather_into_tensor`` on device memory.  Same images as the plain single-process run, bit
    for bit (same batch, same kernels).  No scaling claim: one rank."""
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SD_DIST_FORCE_INIT"):
        os.environ.pop(k, None)
    want, _ = _run_method("ddim_config.yaml", 3, 3)
    ctx = mp.get_context("spawn")            # the child touches the GPU only after it has started (no exec of a GPU process)
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(0, 1, _free_port(), [("ddim_config.yaml", 3, 3, None)], q, "nccl", True))
    p.start()
    rank, _, got, t = q.get(timeout=600)
    got = torch.from_numpy(got)
    p.join(120)
    assert p.exitcode == 0
    assert got.shape == want.shape and torch.equal(got, want) and t > 0
