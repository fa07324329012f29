"""Worker of tests/test_gpu_bench_and_dist.py::test_rccl_loopback: the ONLY RCCL execution a one-GPU box allows -- a process group
of world size 1 on backend "nccl" (= RCCL on ROCm), created exactly as dist.init_from_env creates the multi-rank one, then every
collective dist.py issues, with the dtypes and shapes it issues them with: the uint8 weight-blob broadcast, the padded float32
all_gather of restored images, float64 SUM / MAX all-reduces, barrier.  It cannot show scaling; it does show that this torch
build's RCCL initialises on the device (dmabuf IPC mode), accepts `device_id`, and carries these dtypes."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddpm_image_restoration_amd import spec, synth  # noqa: E402
from ddpm_image_restoration_amd.model import JPEGDiffusionModel  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29547")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"
# the weight blob, as load_weights_broadcast moves it
model = JPEGDiffusionModel(device=dev, dtype="bf16")
nbytes = model.weight_blob_bytes()
blob = torch.empty(nbytes, dtype=torch.uint8, device=dev)
for name, v in synth.synth_state_dict(spec.ARCH_JPEG).items():
    model._load_one(name, v)
model.finalize(blob=blob, fill=True)
torch.cuda.synchronize(dev)
before = int(blob[:: 4097].to(torch.int64).sum())
dist.broadcast(blob, src=0)
torch.cuda.synchronize(dev)
assert int(blob[:: 4097].to(torch.int64).sum()) == before
# gather of restored images (padded shards), metric reductions, the bench's max over ranks, barrier
img = torch.randn(3, 3, 32, 32, device=dev)
outs = [torch.empty_like(img)]
dist.all_gather(outs, img)
assert torch.equal(outs[0], img)
t = torch.tensor([1.5, 2.0, 3.0], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.SUM)
m = torch.tensor([7.25], dtype=torch.float64, device=dev)
dist.all_reduce(m, op=dist.ReduceOp.MAX)
dist.barrier()
assert t.tolist() == [1.5, 2.0, 3.0] and float(m.item()) == 7.25
# the model still runs on the adopted blob
x = torch.from_numpy(synth.synth_images(1, 32, seed=1)).to(dev)
y = model(x, torch.tensor([0.5], device=dev), torch.tensor([0.5], device=dev))
assert torch.isfinite(y).all()
print(f"RCCL-LOOPBACK-OK blob_bytes={nbytes}", flush=True)
dist.destroy_process_group()
