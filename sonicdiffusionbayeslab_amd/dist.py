"""Batch sharding across the GPUs of one node (new relative to the reference,
which is single-device).  Independent prompt batches are split contiguously over ranks -- one
process per GPU, full weight replica each -- and the ONLY data-path collective is one
all-gather of the final latents (RCCL over xGMI with backend "nccl", gloo on CPU in tests).
EVERY Gaussian the sampling loop consumes -- initial latents, LCM re-noising, the SDE variants of DPM-Solver, whatever
a variant pipeline draws -- goes through ``randn`` below: inside ``shard_draws`` it is drawn for the GLOBAL batch from the
one shared seeded CPU generator, in the order the single-process run consumes it, and sliced per rank, so every world
size feeds every image bit-identical inputs (a rank whose shard is empty adopts rank 0's generator state, which rides
in the one all-gather).  Outputs then agree with the W = 1 run to
rel-L2 <= 5e-3, not bit for bit: the split-K factors of the GEMM / conv kernels depend on the per-rank
UNet batch (fp32 partial sums are grouped differently before the bf16 rounding)."""
from __future__ import annotations

import contextlib
import os
import socket
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def env_world() -> Tuple[int, int, int]:
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_process_group(backend: str = "nccl") -> Tuple[int, int, int]:
    """One process per GPU under ``torch.distributed.run``.  At WORLD_SIZE 1 no communicator is created unless
    ``SD_DIST_FORCE_INIT=1`` asks for one (a one-GPU rehearsal of the RCCL path: communicator creation, the ``device_id``
    bind and ``all_gather_into_tensor`` on device memory all run, over a world of one)."""
    rank, local_rank, world = env_world()
    force = os.environ.get("SD_DIST_FORCE_INIT", "0") == "1"
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            if world != 1:
                # ranks of a world > 1 job started without a launcher would each pick their own port and hang in rendezvous
                raise RuntimeError("MASTER_PORT is not set: start the ranks with `python -m torch.distributed.run --nnodes=1 "
                                   "--nproc-per-node N --master-addr 127.0.0.1 --master-port P ...` (or export MASTER_PORT)")
            # the forced world-1 rehearsal without a launcher: any free port
            s = socket.socket(); s.bind(("127.0.0.1", 0)); os.environ["MASTER_PORT"] = str(s.getsockname()[1]); s.close()
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        kw = {}
        if backend == "nccl" and torch.cuda.device_count() > 0:
            # bind the rank to its GPU BEFORE the RCCL communicator exists (one process per GPU)
            dev = torch.device("cuda", local_rank % torch.cuda.device_count())
            torch.cuda.set_device(dev)
            kw["device_id"] = dev
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def active() -> bool:
    """True when the harness must go through the collective (world > 1, or the forced world-1 rehearsal)."""
    return dist.is_available() and dist.is_initialized()


# ---- global draws -------------------------------------------------------------------------------------------------
_SHARD: Optional[Tuple[int, int, int]] = None        # (global batch, lo, hi) of the pipeline call in progress


@contextlib.contextmanager
def shard_draws(n_global: int, lo: int, hi: int):
    """While active, ``randn`` draws for ``n_global`` images and returns rows [lo, hi)."""
    global _SHARD
    prev, _SHARD = _SHARD, (int(n_global), int(lo), int(hi))
    try:
        yield
    finally:
        _SHARD = prev


def randn(shape, generator=None) -> torch.Tensor:
    """The ONE place the sampling loop draws Gaussians (``prepare_latents``, ``LCMScheduler.step``, the SDE branches of
    ``DPMSolverScheduler.step``): fp32, on the generator's device.  Inside ``shard_draws`` the leading dimension is the
    local shard and the draw is the global batch's, sliced."""
    shape = tuple(int(d) for d in shape)
    gdev = generator.device if generator is not None else torch.device("cpu")
    if _SHARD is None:
        return torch.randn(shape, generator=generator, device=gdev, dtype=torch.float32)
    n, lo, hi = _SHARD
    if shape[0] != hi - lo:
        raise ValueError(f"sharded draw: leading dimension {shape[0]} is not this rank's shard of {hi - lo} images")
    full = torch.randn((n,) + shape[1:], generator=generator, device=gdev, dtype=torch.float32)
    return full[lo:hi].clone()


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous split [lo, hi) of n units; earlier ranks take the remainder."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def global_latents(global_batch: int, channels: int, size: int, seed: int) -> torch.Tensor:
    """The whole job's initial latents from ONE seeded CPU generator (slice per rank)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randn((global_batch, channels, size, size), generator=g, dtype=torch.float32)


def gather_latents(local: torch.Tensor, world: int, global_batch: int, seconds: float = None,
                   generator: Optional[torch.Generator] = None):
    """ONE all-gather of the per-rank results (final latents, or decoded images when the VAE decode is
    sharded too): every rank ends with [global_batch, ...].  Shards may be ragged (global_batch % world != 0),
    so each rank pads to the largest shard.  With ``seconds`` (this rank's loop time) one extra element rides
    in the same collective and the return value is (tensor, max seconds over ranks) -- the job is as slow as
    its slowest rank.  With ``generator`` and more ranks than images, rank 0's generator state rides along too and
    every rank adopts it: a rank with an empty shard ran no pipeline call and drew nothing."""
    if not active():
        return local if seconds is None else (local, float(seconds))
    per = (global_batch + world - 1) // world
    dev = local.device
    if dist.get_backend() == "gloo" and local.is_cuda:      # CPU rehearsal backend
        local = local.cpu()
    item = 1
    for d in local.shape[1:]:
        item *= int(d)
    n = per * item
    state = generator.get_state() if (generator is not None and global_batch < world) else None
    extra = 1 + (state.numel() if state is not None else 0)
    send = torch.zeros(n + extra, dtype=torch.float32, device=local.device)
    send[: local.shape[0] * item] = local.reshape(-1).to(torch.float32)
    send[n] = float(seconds) if seconds is not None else 0.0
    if state is not None:
        send[n + 1:] = state.to(torch.float32).to(local.device)      # bytes 0..255 are exact in fp32
    recv = torch.empty(world * (n + extra), dtype=torch.float32, device=local.device)
    dist.all_gather_into_tensor(recv, send)
    recv = recv.view(world, n + extra)
    if state is not None:
        generator.set_state(recv[0, n + 1:].to(torch.uint8).cpu())
    parts = []
    for r in range(world):
        lo, hi = shard_range(global_batch, r, world)
        parts.append(recv[r, : (hi - lo) * item].view((hi - lo,) + tuple(local.shape[1:])))
    full = torch.cat(parts).to(dtype=local.dtype).to(dev)
    if seconds is None:
        return full
    return full, float(recv[:, n].max().item())
