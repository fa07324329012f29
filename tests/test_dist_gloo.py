"""CPU, gloo, world sizes 2 and 4: the multi-GPU edge collectives (weight-blob broadcast, gather of uneven shards, metric
reduce, max-over-ranks timing) -- the only communication of the data-parallel sampler."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch
    from ddpm_image_restoration_amd import dist as D
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    env = D.init_from_env("gloo")
    # 1. weight blob broadcast: rank 0 owns the packed bytes, the others adopt them
    blob = torch.arange(4096, dtype=torch.int64).to(torch.uint8) if rank == 0 else torch.zeros(4096, dtype=torch.uint8)
    D.broadcast_blob(blob, src=0)
    ok_blob = bool((blob == torch.arange(4096, dtype=torch.int64).to(torch.uint8)).all())
    # 2. uneven shards (5 images on 2 ranks: 3 + 2; 6 on 4 ranks: 2 + 2 + 1 + 1), gathered back in global order
    n_total = world + 2 + (world == 2)
    lo, hi = D.shard_range(n_total, rank, world)
    local = torch.arange(lo, hi, dtype=torch.float32).view(-1, 1, 1, 1).expand(-1, 3, 2, 2).contiguous()
    full = D.gather_images(local, n_total, env)
    ok_gather = full.shape[0] == n_total and bool((full[:, 0, 0, 0] == torch.arange(n_total, dtype=torch.float32)).all())
    # 3. metric all-reduce and max-over-ranks timing
    sums = D.reduce_metrics([float(rank + 1), 1.0], env)
    mx = D.max_over_ranks(10.0 + rank, env)
    D.barrier(env)
    q.put((rank, ok_blob, ok_gather, sums, mx))
    import torch.distributed as dist
    dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("world", [2, 4])
def test_gloo_edges(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=150) for _ in range(world))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    for rank, ok_blob, ok_gather, sums, mx in res:
        assert ok_blob and ok_gather
        assert sums == [world * (world + 1) / 2.0, float(world)]
        assert mx == 10.0 + world - 1


def test_shard_ranges_cover_every_image_once():
    from ddpm_image_restoration_amd import dist as D
    for n in (1, 5, 16, 64, 127, 128):
        for world in (1, 2, 3, 4, 8):
            spans = [D.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
