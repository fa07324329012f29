"""Worker of tests/test_gpu_bench_and_dist.py: one rank of a 2-rank restoration job whose ranks share GPU 0 (gloo collectives;
RCCL needs one GPU per rank).  Rank 0 also restores the whole batch alone and checks the gathered shards against it."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddpm_image_restoration_amd import dist as D  # noqa: E402
from ddpm_image_restoration_amd import spec, synth  # noqa: E402
from ddpm_image_restoration_amd.harness import psnr  # noqa: E402
from ddpm_image_restoration_amd.model import JPEGDiffusionModel  # noqa: E402
from ddpm_image_restoration_amd.sampler import DDRMJPEGSampler  # noqa: E402

env = D.init_from_env()                      # DDPMIR_DIST_BACKEND=gloo is set by the test
assert env.world == 2 and env.backend == "gloo"
dev = torch.device("cuda", 0)
model = JPEGDiffusionModel(device=dev)       # fp32: deterministic, so shards must reproduce the full batch bit for bit
loaded = []
D.load_weights_broadcast(model, lambda: loaded.append(1) or synth.synth_state_dict(spec.ARCH_JPEG), env)
assert (len(loaded) == 1) == (env.rank == 0)                      # only rank 0 touches the checkpoint
n_total = 5                                                        # uneven split: 3 + 2
x0 = torch.from_numpy(synth.synth_images(n_total, 32, seed=9)).to(dev)
lo, hi = D.shard_range(n_total, env.rank, env.world)
sampler = DDRMJPEGSampler(model)
mine = sampler.sample(x0[lo:hi], 10, steps=4, seed=3, image_offset=lo)
full = D.gather_images(mine, n_total, env)
sums = D.reduce_metrics([psnr(mine, x0[lo:hi]) * (hi - lo), hi - lo], env, device=dev)
worst = D.max_over_ranks(float(env.rank + 1), env, device=dev)
D.barrier(env)
if env.rank == 0:
    alone = sampler.sample(x0, 10, steps=4, seed=3)
    assert torch.equal(full, alone), float((full - alone).abs().max())
    assert int(sums[1]) == n_total and worst == 2.0
    per_image = np.mean([psnr(alone[i:i + 1], x0[i:i + 1]) for i in range(n_total)])
    print(f"DIST-GPU-OK images={n_total} mean_psnr={sums[0] / sums[1]:.3f} per_image_mean={per_image:.3f}", flush=True)
torch.distributed.destroy_process_group()
