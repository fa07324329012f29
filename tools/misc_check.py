"""GPU sanity/timing: 512x512 forward, WebP/AVIF step times with the host codec, config-1-like small case."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddpm_image_restoration_amd import spec, synth
from ddpm_image_restoration_amd.model import MODEL_BY_ARCH
from ddpm_image_restoration_amd.sampler import SAMPLER_BY_ARCH
from oracle import ddpm_oracle as O

dev = torch.device("cuda", 0)
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)

def step_time(arch, B, R, q, steps_total, n, dtype="bf16", graph=False):
    m = MODEL_BY_ARCH[arch](device=dev, dtype=dtype, use_graph=graph)
    m.load_state_dict(synth.synth_state_dict(arch))
    s = SAMPLER_BY_ARCH[arch](m)
    y = tt(O.compress(synth.synth_images(B, R, seed=1), q, O.KIND[arch]))
    s.sample(y, q, steps=steps_total, start_step=steps_total - 1, num_steps=2)
    torch.cuda.synchronize()
    t0 = time.time()
    out = s.sample(y, q, steps=steps_total, start_step=steps_total - 3, num_steps=n)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / n
    assert torch.isfinite(out).all()
    del m
    return dt * 1e3

what = sys.argv[1] if len(sys.argv) > 1 else "all"
if what in ("all", "512"):
    m = MODEL_BY_ARCH[1](device=dev, dtype="bf16")
    m.load_state_dict(synth.synth_state_dict(1))
    x = tt(synth.synth_images(1, 512, seed=3)); t = torch.full((1,), 0.5, device=dev)
    y = m(x, t, t); torch.cuda.synchronize()
    t0 = time.time(); y = m(x, t, t); torch.cuda.synchronize()
    print(f"512x512 B=1 bf16 forward: {(time.time()-t0)*1e3:.1f} ms, finite={bool(torch.isfinite(y).all())}, |y| mean {y.abs().mean().item():.3f}", flush=True)
    del m
if what in ("all", "codecs"):
    for arch, nm in ((0, "webp"), (2, "avif"), (1, "jpeg")):
        ms = step_time(arch, 16, 256, 10, 1000, 5)
        print(f"{nm} B=16 256x256 bf16: {ms:.1f} ms/step -> {16/ms:.4f} img/s at T=1000", flush=True)
if what in ("all", "small"):
    for graph in (False, True):
        for dtype in ("f32", "bf16"):
            ms = step_time(1, 4, 32, 10, 50, 10, dtype=dtype, graph=graph)
            print(f"config-1 size (JPEG B=4 32x32) {dtype} graph={graph}: {ms:.2f} ms/step -> {4/(ms*50/1e3):.2f} img/s at T=50", flush=True)
