import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddpm_image_restoration_amd import spec, synth, codec
from ddpm_image_restoration_amd.model import MODEL_BY_ARCH
dev = torch.device("cuda", 0)
m = MODEL_BY_ARCH[2](device=dev, dtype="bf16"); m.load_state_dict(synth.synth_state_dict(2))
for B in (8, 16):
    x = torch.from_numpy(synth.synth_images(B, 256, seed=1)).to(dev); t = torch.full((B,), 0.5, device=dev)
    y = m(x, t, t); torch.cuda.synchronize()
    t0 = time.time(); y = m(x, t, t); torch.cuda.synchronize()
    print(f"AVIF forward B={B}: {(time.time()-t0)*1e3:.1f} ms", flush=True)
    m.profile(True, detail=True); y = m(x, t, t); torch.cuda.synchronize()
    for cls in ("attention_l0", "attention", "forward"):
        ms, n, fl = m.profile_read(cls); print("  ", cls, round(ms, 2), n)
    m.profile(False)
u8 = np.clip(synth.synth_images(16, 256, seed=1) * 127.5 + 127.5, 0, 255).astype(np.uint8)
for B in (8, 16):
    codec.roundtrip_u8(u8[:B], 10, "avif")
    t0 = time.time(); codec.roundtrip_u8(u8[:B], 10, "avif"); print(f"AVIF codec pool B={B}: {(time.time()-t0)*1e3:.1f} ms", flush=True)
