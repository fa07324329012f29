"""Level-0 attention launch time per image inside the sampling loop for B = 12 ... 18 (HIP-event spans of the library's profile)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddpm_image_restoration_amd import codec, spec, synth  # noqa: E402
from ddpm_image_restoration_amd.model import JPEGDiffusionModel  # noqa: E402
from ddpm_image_restoration_amd.sampler import DDRMJPEGSampler  # noqa: E402

dev = torch.device("cuda", 0)
m = JPEGDiffusionModel(device=dev, dtype="bf16")
m.load_state_dict(synth.synth_state_dict(spec.ARCH_JPEG))
s = DDRMJPEGSampler(m)
yall = codec.jpeg_compress(torch.from_numpy(synth.synth_images(18, 256, seed=3)), 10).to(dev)
for rep in range(2):
    for B in (12, 14, 15, 16, 17, 18):
        y = yall[:B].contiguous()
        s.sample(y, 10, steps=1000, start_step=999, num_steps=1, seed=1)
        torch.cuda.synchronize()
        m.profile(True)
        s.sample(y, 10, steps=1000, start_step=998, num_steps=4, seed=1)
        torch.cuda.synchronize()
        m.profile(False)
        a_ms, a_n, _ = m.profile_read("attention_l0")
        f_ms, f_n, _ = m.profile_read("forward")
        print(f"B={B:2d}: level-0 attention {a_ms / a_n:7.3f} ms per launch = {a_ms / a_n / B:.4f} ms per image; forward {f_ms / f_n:.2f} ms = {f_ms / f_n / B:.4f} per image", flush=True)
