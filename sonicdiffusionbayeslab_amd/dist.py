"""Batch sharding across the GPUs of one node (new relative to the reference,
which is single-device).  Independent prompt batches are split contiguously over ranks -- one
process per GPU, full weight replica each -- and the ONLY data-path collective is one
all-gather of the final latents (RCCL over xGMI with backend "nccl", gloo on CPU in tests).
To make world-size-W output bit-comparable with W=1, the GLOBAL initial latents (and LCM
noise) are drawn from one seeded CPU generator and sliced per rank."""
from __future__ import annotations

import os
from typing import Tuple

import torch
import torch.distributed as dist


def env_world() -> Tuple[int, int, int]:
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_process_group(backend: str = "nccl") -> Tuple[int, int, int]:
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        kw = {}
        if backend == "nccl" and torch.cuda.device_count() > 0:
            # bind the rank to its GPU BEFORE the RCCL communicator exists (one process per GPU)
            dev = torch.device("cuda", local_rank % torch.cuda.device_count())
            torch.cuda.set_device(dev)
            kw["device_id"] = dev
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


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


def gather_latents(local: torch.Tensor, world: int, global_batch: int) -> torch.Tensor:
    """All-gather of final latents: every rank ends with [global_batch, 4, H, W].  Shards may be
    ragged (global_batch % world != 0), so each rank pads to the largest shard."""
    if world == 1:
        return local
    per = (global_batch + world - 1) // world
    dev = local.device
    if dist.get_backend() == "gloo" and local.is_cuda:      # CPU rehearsal backend
        local = local.cpu()
    pad = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad.contiguous())
    parts = []
    for r in range(world):
        lo, hi = shard_range(global_batch, r, world)
        parts.append(out[r][: hi - lo])
    return torch.cat(parts).to(dev)
