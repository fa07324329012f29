"""Evaluation harness mirroring test_webp_restoration / test_avif_restoration (webp_inference.py:604-797,
avif_inference.py:462-657): for every quality level compress -> pick steps -> restore -> score, and report the
same JSON keys (`compressed_psnr` ... `restored_l2`).  PSNR and L2 are first-party formulas of the reference;
SSIM restates the published pytorch_msssim algorithm the reference calls (package absent here: parity unpinned);
LPIPS/FID need pretrained AlexNet/Inception weights that cannot be fetched offline, so those keys are omitted.
"""
from __future__ import annotations

import json
import math
import os

from . import codec, spec

COMPRESS = {spec.ARCH_WEBP: codec.webp_compress, spec.ARCH_JPEG: codec.jpeg_compress, spec.ARCH_AVIF: codec.avif_compress}
DEFAULT_QUALITIES = {spec.ARCH_WEBP: [0, 5, 10, 30, 50, 70, 90],      # webp_inference.py:604
                     spec.ARCH_JPEG: [10, 20, 30, 50],                # svd.ipynb c1
                     spec.ARCH_AVIF: [1, 10, 20, 30, 50, 70, 90]}     # avif_inference.py:462


def steps_for_quality(arch: int, quality: int) -> int:
    """init_t = int((100 - q) / 100 * 100), clamped to [lo, hi]   (webp_inference.py:678-679, avif_inference.py:534-535).
    The reference forms the product in floating point, which gives one step fewer than 100 - q at q = 42, 43, 71."""
    k = spec.CONSTS[arch]
    return max(k.steps_lo, min(k.steps_hi, int((100 - quality) / 100 * 100)))


def psnr(a, b):
    """-10 log10(mse + 1e-8) on clamp(x*0.5+0.5, 0, 1)   (webp_inference.py:691-697)."""
    import torch
    a01 = torch.clamp(a * 0.5 + 0.5, 0, 1)
    b01 = torch.clamp(b * 0.5 + 0.5, 0, 1)
    mse = torch.mean((a01 - b01) ** 2).item()
    return -10.0 * math.log10(mse + 1e-8)


def l2(a, b):
    """torch.norm(a - b) / sqrt(numel)   (webp_inference.py:700-707)."""
    import torch
    a01 = torch.clamp(a * 0.5 + 0.5, 0, 1)
    b01 = torch.clamp(b * 0.5 + 0.5, 0, 1)
    return (torch.norm(a01 - b01) / math.sqrt(a01.numel())).item()


def ssim(a, b, win_size=11, sigma=1.5, k1=0.01, k2=0.03):
    """pytorch_msssim.ssim(x01, y01, data_range=1.0) as called at webp_inference.py:698: separable 11-tap Gaussian
    (sigma 1.5, 'valid'), per-channel SSIM map averaged over space, then over channels and batch.  A spatial dim
    smaller than the window is left unsmoothed, as the package does."""
    import torch
    import torch.nn.functional as F
    x = torch.clamp(a * 0.5 + 0.5, 0, 1).double()
    y = torch.clamp(b * 0.5 + 0.5, 0, 1).double()
    g = torch.arange(win_size, dtype=torch.float64, device=x.device) - win_size // 2
    g = torch.exp(-(g ** 2) / (2 * sigma ** 2)); g = g / g.sum()
    C = x.shape[1]

    def blur(z):
        if z.shape[2] >= win_size:
            z = F.conv2d(z, g.view(1, 1, -1, 1).repeat(C, 1, 1, 1), groups=C)
        if z.shape[3] >= win_size:
            z = F.conv2d(z, g.view(1, 1, 1, -1).repeat(C, 1, 1, 1), groups=C)
        return z
    c1, c2 = k1 ** 2, k2 ** 2
    mu1, mu2 = blur(x), blur(y)
    s1 = blur(x * x) - mu1 * mu1
    s2 = blur(y * y) - mu2 * mu2
    s12 = blur(x * y) - mu1 * mu2
    cs = (2 * s12 + c2) / (s1 + s2 + c2)
    m = ((2 * mu1 * mu2 + c1) / (mu1 * mu1 + mu2 * mu2 + c1)) * cs
    return m.flatten(2).mean(-1).mean().item()


def test_restoration(model, sampler, images, quality_levels=None, output_dir=None, seed=0, env=None):
    """images: iterable of [B,3,H,W] tensors in [-1,1].  Returns {quality: averaged metrics} like the reference's
    avg_results and writes metrics_summary.json when output_dir is given.  With a DistEnv the image list is
    sharded across ranks and the sums are all-reduced.  The device noise of every image is keyed by (seed, GLOBAL image
    index), so restored images and metrics do not depend on the number of ranks."""
    import torch
    from . import dist as D
    arch = model.ARCH
    qs = quality_levels or DEFAULT_QUALITIES[arch]
    batches = list(images)
    first = [0]                                   # global index of the first image of every batch
    for b in batches:
        first.append(first[-1] + int(b.shape[0]))
    if env is not None and env.world > 1:
        lo, hi = D.shard_range(len(batches), env.rank, env.world)
        batches, first = batches[lo:hi], first[lo:hi]
    results = {}
    for q in qs:
        acc = [0.0] * 7
        for bi, x0 in enumerate(batches):
            x0 = x0.to(model.device)
            y = COMPRESS[arch](x0, q)
            restored = sampler.sample(y, q, steps=steps_for_quality(arch, q), seed=seed, image_offset=first[bi])
            n = x0.shape[0]
            acc[0] += psnr(y, x0) * n; acc[1] += psnr(restored, x0) * n
            acc[2] += l2(y, x0) * n; acc[3] += l2(restored, x0) * n; acc[4] += n
            acc[5] += ssim(y, x0) * n; acc[6] += ssim(restored, x0) * n
        if env is not None:
            acc = D.reduce_metrics(acc, env, device=model.device)
        n = max(acc[4], 1.0)
        results[q] = {"compressed_psnr": acc[0] / n, "compressed_ssim": acc[5] / n, "compressed_l2": acc[2] / n,
                      "restored_psnr": acc[1] / n, "restored_ssim": acc[6] / n, "restored_l2": acc[3] / n,
                      "psnr_gain": (acc[1] - acc[0]) / n, "count": int(acc[4])}
    if output_dir and (env is None or env.rank == 0):
        os.makedirs(output_dir, exist_ok=True)
        with open(os.path.join(output_dir, "metrics_summary.json"), "w") as f:
            json.dump({str(k): v for k, v in results.items()}, f, indent=2)
    return results


def _load_image_dir(path, size, limit):
    """ImageFolderFlat + Resize + ToTensor + Normalize(0.5, 0.5) of webp_inference.py:32-58, as [1,3,S,S] tensors."""
    import numpy as np
    import torch
    from PIL import Image
    names = sorted(f for f in os.listdir(path) if f.lower().endswith((".png", ".jpg", ".jpeg", ".bmp", ".webp")))
    for f in names[:limit]:
        im = Image.open(os.path.join(path, f)).convert("RGB").resize((size, size), Image.BILINEAR)
        a = torch.from_numpy(np.asarray(im, dtype=np.uint8).copy()).permute(2, 0, 1).float().div(255)
        yield a.sub(0.5).div(0.5).unsqueeze(0)


def main(argv=None):
    """CLI mirror of `python webp_inference.py` / `avif_inference.py` main(): restore a folder of images (or a
    synthetic set) at several qualities and write metrics_summary.json.  Multi-GPU: launch with torch.distributed.run."""
    import argparse
    import torch
    from . import dist as D
    from . import synth
    from .model import MODEL_BY_ARCH, load_checkpoint
    from .sampler import SAMPLER_BY_ARCH
    ap = argparse.ArgumentParser(description=main.__doc__)
    ap.add_argument("--arch", choices=list(spec.ARCH_NAMES), default="webp")
    ap.add_argument("--checkpoint", default=None, help="best_ddrm_*_model.pth (state_dict or {'model_state_dict': ...}); "
                                                        "omit for synthetic weights")
    ap.add_argument("--images", default=None, help="folder of images; omit for synthetic images")
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--num-images", type=int, default=8)
    ap.add_argument("--qualities", type=int, nargs="*", default=None)
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32")
    ap.add_argument("--output-dir", default="./restoration_results")
    args = ap.parse_args(argv)
    env = D.init_from_env()
    dev = torch.device("cuda", env.local_rank)
    arch = spec.ARCH_NAMES[args.arch]
    model = MODEL_BY_ARCH[arch](device=dev, dtype=args.dtype)
    if args.checkpoint:
        D.load_weights_broadcast(model, lambda: (lambda c: c["model_state_dict"] if "model_state_dict" in c else c)(
            torch.load(args.checkpoint, map_location="cpu")), env)
    else:
        D.load_weights_broadcast(model, lambda: synth.synth_state_dict(arch), env)
    print(f"Model parameters: {sum(p.numel() for p in model.parameters()):,}")
    if args.images:
        batches = list(_load_image_dir(args.images, args.size, args.num_images))
    else:
        batches = [torch.from_numpy(synth.synth_images(1, args.size, seed=1000 + i)) for i in range(args.num_images)]
    res = test_restoration(model, SAMPLER_BY_ARCH[arch](model), batches, args.qualities, args.output_dir, env=env)
    if env.rank == 0:
        for q, r in res.items():
            print(f"quality {q:3d}: PSNR {r['compressed_psnr']:.2f} -> {r['restored_psnr']:.2f} dB "
                  f"(gain {r['psnr_gain']:+.2f}), SSIM {r['compressed_ssim']:.4f} -> {r['restored_ssim']:.4f}, L2 {r['compressed_l2']:.4f} -> {r['restored_l2']:.4f}, n={r['count']}")
    return res


if __name__ == "__main__":
    main()
