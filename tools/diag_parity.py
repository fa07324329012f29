"""GPU diagnostic: forward error statistics vs goldens, trajectory divergence, first timings."""
import sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddpm_image_restoration_amd import spec, synth
from ddpm_image_restoration_amd.model import MODEL_BY_ARCH
from ddpm_image_restoration_amd.sampler import SAMPLER_BY_ARCH
from oracle import ddpm_oracle as O

tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
for nm, arch in (("webp", 0), ("jpeg", 1), ("avif", 2)):
    g = dict(np.load(f"tests/golden/{nm}.npz"))
    m = MODEL_BY_ARCH[arch](device=torch.device("cuda", 0))
    m.load_state_dict(synth.synth_state_dict(arch))
    y = m(tt(g["fwd32_x"]), tt(g["fwd32_t"]), tt(g["fwd32_cl"])).cpu().numpy()
    d = np.abs(y - g["fwd32_y"])
    print(nm, "fwd32 err max %.3e mean %.3e" % (d.max(), d.mean()))
    y = m(tt(g["fwd64_x"]), tt(g["fwd64_t"]), tt(g["fwd64_t"])).cpu().numpy()
    d = np.abs(y - g["fwd64_y"])
    print(nm, "fwd64 err max %.3e mean %.3e" % (d.max(), d.mean()))
    steps, q, b = int(g["samp_steps"]), int(g["samp_q"]), g["samp_y"].shape[0]
    noise = synth.synth_noise(steps, b, 32, 32, seed=7)
    out, tr = SAMPLER_BY_ARCH[arch](m).sample(tt(g["samp_y"]), q, steps=steps, noise=tt(noise), trace=True, device_codec=False)
    tr = tr.cpu().numpy(); out = out.cpu().numpy()
    for i in sorted((int(v) for v in g["samp_keep"]), reverse=True):
        th = tr[steps - 1 - i, 0]
        dd = np.abs(th - g[f"samp_theta_{i}"])
        print(nm, f"  step i={i:3d} theta err max {dd.max():.3e} frac>1e-3 {(dd > 1e-3).mean():.4f}")
    dd = np.abs(out - g["samp_final"])
    print(nm, "final err max %.3e frac>1e-3 %.4f psnr %.4f vs %.4f" % (dd.max(), (dd > 1e-3).mean(), O.psnr(out, g["samp_x0"]), O.psnr(g["samp_final"], g["samp_x0"])))
    if arch == 1:
        for (B, R) in ((4, 32), (4, 64), (2, 128), (1, 256)):
            x = tt(synth.synth_images(B, R, seed=1)); t = torch.full((B,), 0.5).cuda()
            m(x, t, t); torch.cuda.synchronize()
            t0 = time.time(); n = 3
            for _ in range(n): m(x, t, t)
            torch.cuda.synchronize()
            print(f"  forward B={B} R={R}: {(time.time()-t0)/n*1e3:.2f} ms")
    del m
