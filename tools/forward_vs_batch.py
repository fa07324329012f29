"""Forward time vs batch (JPEG family, 256x256, bf16): total, the two level-0 attention launches, and the rest -- how much of
a forward is a fixed cost (latency-bound deep levels) that micro-batch lanes pay twice."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddpm_image_restoration_amd import spec, synth
from ddpm_image_restoration_amd.model import JPEGDiffusionModel
m = JPEGDiffusionModel(device="cuda:0", dtype="bf16")
m.load_state_dict(synth.synth_state_dict(spec.ARCH_JPEG))
for B in (1, 2, 4, 8, 16, 32):
    x = torch.from_numpy(synth.synth_images(B, 256, seed=1)).cuda()
    t = torch.full((B,), 0.5, device="cuda")
    for _ in range(2): m(x, t, t)
    torch.cuda.synchronize()
    m.profile(True, detail=True)
    for _ in range(5): m(x, t, t)
    torch.cuda.synchronize()
    m.profile(False)
    f_ms, f_n, _ = m.profile_read("forward")
    a_ms, a_n, _ = m.profile_read("attention_l0")
    al_ms, al_n, _ = m.profile_read("attention")
    f, a0, aall = f_ms / f_n, a_ms / f_n, al_ms / f_n
    print(f"B={B:2d}: forward {f:7.2f} ms  level-0 attention {a0:7.2f}  all attention {aall:7.2f}  rest {f - aall:6.2f} ms ({(f - aall) / B:5.2f} per image)", flush=True)
