"""Run-to-run spread of the bf16 forward (split-K float atomics make it non-bit-reproducible): prints max / mean |diff|."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ddpm_image_restoration_amd import synth, spec
from ddpm_image_restoration_amd.model import MODEL_BY_ARCH
arch = int(sys.argv[1]) if len(sys.argv) > 1 else 0
m = MODEL_BY_ARCH[arch](device="cuda:0", dtype="bf16"); m.load_state_dict(synth.synth_state_dict(arch))
for B, R in ((1, 32), (1, 64), (2, 64), (4, 64), (1, 128), (2, 256)):
    x = torch.from_numpy(synth.synth_images(B, R, seed=5)).cuda(); t = torch.full((B,), 0.4, device="cuda")
    ys = [m(x, t, t).clone() for _ in range(3)]
    d01, d12 = (ys[0] - ys[1]).abs(), (ys[1] - ys[2]).abs()
    print(f"B={B} R={R}: run0-run1 max {float(d01.max()):.2e} mean {float(d01.mean()):.2e}; run1-run2 max {float(d12.max()):.2e} mean {float(d12.mean()):.2e}", flush=True)
