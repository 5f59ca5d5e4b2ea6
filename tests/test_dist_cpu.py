"""CPU: the N>1 path (batch shard + ONE all-gather of final latents) with world_size 2 over gloo."""
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
    q.put((rank, full))
    dist.barrier()
    dist.destroy_process_group()


def _run(gb):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, gb, q)) for r in range(2)]
    [p.start() for p in ps]
    outs = dict(q.get(timeout=120) for _ in range(2))
    [p.join(60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    return outs


def test_gather_equals_single_process_even_and_ragged():
    from sonicdiffusionbayeslab_amd.dist import global_latents
    for gb in (8, 5):
        outs = _run(gb)
        want = global_latents(gb, 4, 8, seed=29) * 2.0 + 1.0
        assert torch.equal(outs[0], want) and torch.equal(outs[1], want)
