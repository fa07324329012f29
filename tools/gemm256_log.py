"""Which kernel takes each Cout >= 256 3x3 conv of one forward (DDPMIR_GEMM256_LOG=1): configs[4] shape in bf16 and fp8 mode,
and the headline shape (256 x 256, B=16)."""
import os
import sys

os.environ["DDPMIR_GEMM256_LOG"] = "1"
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddpm_image_restoration_amd import spec, synth  # noqa: E402
from ddpm_image_restoration_amd.model import JPEGDiffusionModel  # noqa: E402

dev = torch.device("cuda", 0)
sd = synth.synth_state_dict(spec.ARCH_JPEG)
for name, kw, R, B in (("configs[4] bf16", dict(dtype="bf16"), 512, 8), ("configs[4] fp8", dict(dtype="fp8"), 512, 8),
                       ("headline shape bf16", dict(dtype="bf16"), 256, 16)):
    m = JPEGDiffusionModel(device=dev, attn_levels="bottleneck", **kw)
    m.load_state_dict(sd)
    x = torch.from_numpy(synth.synth_images(B, R, seed=1)).to(dev)
    t = torch.full((B,), 0.5, device=dev)
    print(f"==== {name}", file=sys.stderr, flush=True)
    m(x, t, t)
    torch.cuda.synchronize()
    del m
