"""One process per GPU; independent image batches shard across ranks; RCCL only at the edges.

The sampling loop has no cross-GPU dependency: images are independent, so the batch
is split contiguously across ranks and no collective runs inside the T-step loop.  RCCL (torch.distributed
backend "nccl" on ROCm) is used for exactly two things: one broadcast of the packed weight blob from rank 0
over xGMI at start-up, and a final gather / all-reduce of results and metrics.  Device noise is keyed by the
GLOBAL image index, so results do not depend on the number of GPUs.  The same code runs on CPU with the gloo
backend (tests/test_dist_gloo.py).
"""
from __future__ import annotations

import os
from dataclasses import dataclass


@dataclass
class DistEnv:
    rank: int
    world: int
    local_rank: int
    backend: str


def init_from_env(backend: str | None = None) -> DistEnv:
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torch.distributed.run) and joins the process group."""
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if backend is None:
        # DDPMIR_DIST_BACKEND=gloo: rehearse the multi-rank path with several ranks sharing ONE GPU (RCCL refuses that)
        backend = os.environ.get("DDPMIR_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return DistEnv(rank, world, local, backend)


def shard_range(n_images: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous split; the first (n % world) ranks take one extra image."""
    base, extra = divmod(n_images, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def broadcast_blob(blob, src: int = 0):
    """In-place broadcast of the packed weight blob (a uint8 tensor on the rank's device)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(blob, src=src)
    return blob


def load_weights_broadcast(model, state_dict_fn, env: DistEnv):
    """Rank 0 loads + packs the checkpoint; every other rank receives the packed blob over RCCL/xGMI and adopts
    it without touching the checkpoint.  `state_dict_fn()` is only called on rank 0."""
    import torch
    nbytes = model.weight_blob_bytes()
    blob = torch.empty(nbytes, dtype=torch.uint8, device=model.device)
    if env.rank == 0:
        sd = state_dict_fn()
        for name, v in sd.items():
            model._load_one(name, v)
        model.finalize(blob=blob, fill=True)
    torch.cuda.synchronize(model.device)
    broadcast_blob(blob, src=0)
    if env.rank != 0:
        model.finalize(blob=blob, fill=False)
    return model


def gather_images(local, n_total: int, env: DistEnv):
    """All-gather the per-rank restored images back into global order (uneven shards are padded)."""
    import torch
    import torch.distributed as dist
    if env.world == 1:
        return local
    per = -(-n_total // env.world)
    pad = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    outs = [torch.empty_like(pad) for _ in range(env.world)]
    dist.all_gather(outs, pad)
    pieces = []
    for r in range(env.world):
        a, b = shard_range(n_total, r, env.world)
        pieces.append(outs[r][: b - a])
    return torch.cat(pieces, dim=0)


def reduce_metrics(sums, env: DistEnv, device=None):
    """All-reduce (SUM) a small list of float accumulators, e.g. [sum_psnr, sum_ssim, count]."""
    import torch
    import torch.distributed as dist
    t = torch.tensor(list(sums), dtype=torch.float64, device=device)
    if env.world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.tolist()


def max_over_ranks(value: float, env: DistEnv, device=None) -> float:
    import torch
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device=device)
    if env.world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier(env: DistEnv):
    import torch.distributed as dist
    if env.world > 1:
        dist.barrier()
