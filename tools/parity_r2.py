"""GPU diagnostic behind the round-2 parity tests: measures what the tests then bound (run once, numbers quoted in the tests).
  1. which image of the 50-step JPEG golden run leaves the reference trajectory, and between which kept steps
  2. bf16-mode teacher-forced error along the golden trajectories (all three families)
  3. restored PSNR of bf16 mode vs fp32 mode over 16 images (JPEG q10, 32x32 and 64x64, the harness' step rule)
  4. fp32 / bf16 error against the full-size oracle fixtures
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ddpm_image_restoration_amd import codec, harness, spec, synth  # noqa: E402
from ddpm_image_restoration_amd.model import MODEL_BY_ARCH  # noqa: E402
from ddpm_image_restoration_amd.sampler import SAMPLER_BY_ARCH  # noqa: E402

tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
NAMES = {0: "webp", 1: "jpeg", 2: "avif"}
what = sys.argv[1:] or ["flip", "tf16", "psnr", "full"]


def model(arch, **kw):
    m = MODEL_BY_ARCH[arch](device=torch.device("cuda", 0), **kw)
    m.load_state_dict(synth.synth_state_dict(arch))
    return m


def psnr_each(a, b):
    a01, b01 = np.clip(a * 0.5 + 0.5, 0, 1), np.clip(b * 0.5 + 0.5, 0, 1)
    mse = ((a01 - b01) ** 2).reshape(a.shape[0], -1).mean(axis=1)
    return -10 * np.log10(mse + 1e-8)


for arch in (0, 1, 2):
    nm = NAMES[arch]
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", nm + ".npz")))
    m32 = model(arch)
    m16 = model(arch, dtype="bf16")
    steps, q, b = int(g["samp_steps"]), int(g["samp_q"]), g["samp_y"].shape[0]
    if "flip" in what:
        noise = synth.synth_noise(steps, b, 32, 32, seed=7)
        out, tr = SAMPLER_BY_ARCH[arch](m32).sample(tt(g["samp_y"]), q, steps=steps, noise=tt(noise), trace=True, device_codec=False)
        tr, out = tr.cpu().numpy(), out.cpu().numpy()
        for i in sorted((int(v) for v in g["samp_keep"]), reverse=True):
            e_in = np.abs(tr[steps - 1 - i - 1, 1] - g[f"samp_xt_in_{i}"]).reshape(b, -1).max(1) if i < steps - 1 else np.zeros(b)
            e_th = np.abs(tr[steps - 1 - i, 0] - g[f"samp_theta_{i}"]).reshape(b, -1).max(1)
            print(nm, f"i={i:3d} per-image max err: x_t in", " ".join(f"{v:.1e}" for v in e_in), "| theta", " ".join(f"{v:.1e}" for v in e_th))
        print(nm, "final per-image", np.abs(out - g["samp_final"]).reshape(b, -1).max(1))
        if arch == 1:
            gs = np.load(os.path.join(ROOT, "tests", "golden", "jpeg_steps49_39.npz"))
            u8 = lambda v: np.clip(v * np.float32(127.5) + np.float32(127.5), 0, 255).astype(np.uint8)  # noqa: E731
            for i in range(49, 38, -1):
                th, th_ref = tr[steps - 1 - i, 0], gs[f"theta_{i}"]
                e_th = np.abs(th - th_ref).reshape(b, -1).max(1)
                e_in = np.abs(tr[steps - 2 - i, 1] - gs[f"xt_in_{i}"]).reshape(b, -1).max(1) if i < steps - 1 else np.zeros(b)
                flips = (u8(th) != u8(th_ref)).reshape(b, -1).sum(1)
                print(f"jpeg step i={i}: x_t in", " ".join(f"{v:.1e}" for v in e_in), "| theta", " ".join(f"{v:.1e}" for v in e_th), "| u8 flips", flips)
        for mode, mm in (("f32", m32), ("bf16", m16)):
            n2 = synth.synth_noise(int(g["samp2_steps"]), 2, 32, 32, seed=8)
            o2 = SAMPLER_BY_ARCH[arch](mm).sample(tt(g["samp2_y"]), int(g["samp2_q"]), steps=int(g["samp2_steps"]), noise=tt(n2), device_codec=False).cpu().numpy()
            print(nm, mode, "q50 run per-image", np.abs(o2 - g["samp2_final"]).reshape(2, -1).max(1),
                  "psnr", psnr_each(o2, g["samp_x0"][:2]), "ref", psnr_each(g["samp2_final"], g["samp_x0"][:2]))
    if "tf16" in what:
        for i in g["samp_keep"]:
            t = (np.full((b,), int(i), np.float32) / np.float32(steps)).astype(np.float32)
            th = m16(tt(g[f"samp_xt_in_{int(i)}"]), tt(t), tt(t)).cpu().numpy()
            e = np.abs(th - g[f"samp_theta_{int(i)}"])
            print(nm, f"bf16 teacher-forced i={int(i):3d}: max {e.max():.3e} mean {e.mean():.3e}")
    if "psnr" in what:
        for R in (32, 64):
            for qq in ((10, 50) if arch == 1 else (10,)):
                n_img = 16
                x0 = synth.synth_images(n_img, R, seed=500 + R)
                y = harness.COMPRESS[arch](torch.from_numpy(x0), qq).cuda()
                st = harness.steps_for_quality(arch, qq)
                r32 = SAMPLER_BY_ARCH[arch](m32).sample(y, qq, steps=st, seed=3).cpu().numpy()
                r16 = SAMPLER_BY_ARCH[arch](m16).sample(y, qq, steps=st, seed=3).cpu().numpy()
                p32, p16, pc = psnr_each(r32, x0), psnr_each(r16, x0), psnr_each(y.cpu().numpy(), x0)
                d = p16 - p32
                print(nm, f"R={R} q={qq} steps={st}: compressed {pc.mean():.4f} dB, restored f32 {p32.mean():.4f} bf16 {p16.mean():.4f}; "
                      f"mean diff {d.mean():+.4f} dB, per-image |diff| max {np.abs(d).max():.4f}, pixels max {np.abs(r16 - r32).max():.3f} "
                      f"batch-PSNR f32 {harness.psnr(torch.from_numpy(r32), torch.from_numpy(x0)):.4f} bf16 {harness.psnr(torch.from_numpy(r16), torch.from_numpy(x0)):.4f}")
    if "psnrN" in what and arch == 1:
        import time
        for R, nbatch, bs in ((32, 4, 256), (64, 2, 128)):
            d_all, p_all = [], []
            t0 = time.time()
            for k in range(nbatch):
                x0 = synth.synth_images(bs, R, seed=900 + 10 * R + k)
                y = harness.COMPRESS[arch](torch.from_numpy(x0), 10).cuda()
                st = harness.steps_for_quality(arch, 10)
                r32 = SAMPLER_BY_ARCH[arch](m32).sample(y, 10, steps=st, seed=3, image_offset=k * bs).cpu().numpy()
                r16 = SAMPLER_BY_ARCH[arch](m16).sample(y, 10, steps=st, seed=3, image_offset=k * bs).cpu().numpy()
                p32, p16 = psnr_each(r32, x0), psnr_each(r16, x0)
                d_all.append(p16 - p32); p_all.append(p32)
            d = np.concatenate(d_all); p32 = np.concatenate(p_all)
            print(nm, f"R={R} N={d.size}: restored f32 mean {p32.mean():.4f} dB; bf16-f32 mean {d.mean():+.5f} dB, std {d.std():.4f}, "
                  f"|d| max {np.abs(d).max():.4f}, p99 {np.percentile(np.abs(d), 99):.4f}; wall {time.time() - t0:.1f} s", flush=True)
    if "psnr256" in what and arch in (0, 1):
        # the benchmarked size: 16 images at 256x256, q10, the harness' step rule, fp32 parity mode vs bf16 (and fp8) mode
        import time
        R, bs, qq = 256, 16, 10
        x0 = synth.synth_images(bs, R, seed=777)
        y = harness.COMPRESS[arch](torch.from_numpy(x0), qq).cuda()
        st = harness.steps_for_quality(arch, qq)
        res = {}
        for mode, mm in (("f32", m32), ("bf16", m16), ("fp8", model(arch, dtype="fp8"))):
            t0 = time.time()
            res[mode] = SAMPLER_BY_ARCH[arch](mm).sample(y, qq, steps=st, seed=3).cpu().numpy()
            print(nm, f"R=256 {mode}: {st} steps in {time.time() - t0:.1f} s", flush=True)
        p32 = psnr_each(res["f32"], x0)
        pc = psnr_each(y.cpu().numpy(), x0)
        for mode in ("bf16", "fp8"):
            d = psnr_each(res[mode], x0) - p32
            print(nm, f"R=256 N={bs} q={qq} steps={st}: compressed {pc.mean():.4f} dB, restored f32 {p32.mean():.4f} dB; {mode}-f32 mean {d.mean():+.5f} dB, "
                  f"std {d.std():.4f}, |d| max {np.abs(d).max():.4f}; pixels: max |diff| {np.abs(res[mode] - res['f32']).max():.3f}, "
                  f"mean {np.abs(res[mode] - res['f32']).mean():.5f}", flush=True)
    del m32, m16
    torch.cuda.empty_cache()

if "full" in what:
    for name, arch, aml in (("webp_fwd256_oracle", 0, 0), ("avif_fwd128_oracle", 2, 0), ("avif_fwd256_oracle", 2, 0),
                            ("jpeg_fwd512_bott_oracle", 1, 5), ("avif_fwd128_ref", 2, 0)):
        p = os.path.join(ROOT, "tests", "golden", name + ".npz")
        if not os.path.exists(p):
            print(name, "missing")
            continue
        g = np.load(p)
        s = int(g["stride"]) if "stride" in g else 1
        size = g["y"].shape[-1] * s
        x = tt(synth.synth_images(1, size, seed=int(g["seed_x"])))
        for dtype in ("f32", "bf16"):
            m = model(arch, dtype=dtype, attn_levels="all" if aml == 0 else "bottleneck")
            y = m(x, tt(g["t"]), tt(g["cl"])).cpu().numpy()
            e = np.abs(y[:, :, ::s, ::s] - g["y"])
            extra = f" sum {y.astype(np.float64).sum():.6f} vs {float(g['sum']):.6f}" if "sum" in g else ""
            print(name, dtype, f"max {e.max():.3e} mean {e.mean():.3e}" + extra)
            del m
            torch.cuda.empty_cache()
