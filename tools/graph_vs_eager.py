"""U-Net forward as a replayed hipGraph vs eager launches (BASELINE configs[1] shape)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ddpm_image_restoration_amd import synth, spec
from ddpm_image_restoration_amd.model import JPEGDiffusionModel
sd = synth.synth_state_dict(spec.ARCH_JPEG)
x = torch.from_numpy(synth.synth_images(16, 256, seed=1)).cuda(); t = torch.full((16,), 0.5, device="cuda")
for graph in (False, True):
    m = JPEGDiffusionModel(device="cuda:0", dtype="bf16", use_graph=graph); m.load_state_dict(sd)
    for _ in range(2): y = m(x, t, t)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): y = m(x, t, t)
    torch.cuda.synchronize()
    print(f"use_graph={graph}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per forward", flush=True)
    del m
