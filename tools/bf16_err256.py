"""bf16-mode forward error on the 256 x 256 fixtures (tests/golden/*_fwd256_oracle.npz), max / mean abs; environment switches of
the library (DDPMIR_NO_FREQ_FUSE=1, ...) apply, so this is the bisecting companion of tests/test_gpu_fullsize.py."""
import sys, os, glob
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ddpm_image_restoration_amd import synth, spec
from ddpm_image_restoration_amd.model import MODEL_BY_ARCH
gold = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
opts = dict(kv.split("=") for kv in sys.argv[1:])
for name, arch in (("webp", spec.ARCH_WEBP),):
    f = os.path.join(gold, name + "_fwd256_oracle.npz")
    if not os.path.exists(f): continue
    g = np.load(f)
    s = int(g["stride"]) if "stride" in g.files else 1
    x = tt(synth.synth_images(1, g["y"].shape[-1] * s, seed=int(g["seed_x"])))
    m = MODEL_BY_ARCH[arch](device=torch.device("cuda", 0), attn_levels="all", dtype="bf16")
    m.load_state_dict(synth.synth_state_dict(arch, seed=0))
    for k, v in opts.items(): m.set_option(k, int(v))
    y = m(x, tt(g["t"]), tt(g["cl"])).cpu().numpy()
    e = np.abs(y[:, :, ::s, ::s] - g["y"])
    i = np.unravel_index(e.argmax(), e.shape)
    print(f"{name} 256 {opts}: max {e.max():.5f} at {i} mean {e.mean():.6f}; errors > 0.02: {(e > 0.02).sum()}", flush=True)
