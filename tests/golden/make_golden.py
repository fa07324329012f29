#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE's own code (build container only).

The reference (Azure0413/DDPM_Image_Restoration) cannot be imported as a module (module-level dataset
listing, missing torchvision/lpips), so its class/function definitions are exec()'d by line range from the
files where they lie under /root/reference -- nothing from the reference is copied into this repo, only
inputs and the outputs it produced.  torchvision's ToPILImage/ToTensor are replaced by an exact PIL+numpy
shim (uint8 CHW <-> PIL HWC, /255), see SURVEY.md section 8c.

Weights are the deterministic synthetic state of ddpm_image_restoration_amd.synth (seed 0), loaded into the
reference modules with load_state_dict(strict=True), so tests regenerate them instead of storing 458 MB.

Usage:  python tests/golden/make_golden.py            (writes next to this file; skips if no /root/reference)
"""
from __future__ import annotations

import io
import json
import math
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = os.environ.get("DDPMIR_REFERENCE", "/root/reference")

from ddpm_image_restoration_amd import spec, synth  # noqa: E402


def _namespace():
    import torch
    import torch.nn.functional as F
    from PIL import Image
    from torch import nn

    tv = types.ModuleType("torchvision")
    tr = types.ModuleType("torchvision.transforms")

    class ToPILImage:
        def __call__(self, t):
            return Image.fromarray(np.ascontiguousarray(t.permute(1, 2, 0).numpy()))

    class ToTensor:
        def __call__(self, im):
            a = np.asarray(im.convert("RGB"), dtype=np.uint8).copy()
            return torch.from_numpy(a).permute(2, 0, 1).float().div(255)

    tr.ToPILImage, tr.ToTensor = ToPILImage, ToTensor
    tv.transforms = tr
    return dict(torch=torch, nn=nn, F=F, math=math, io=io, np=np, Image=Image, device=torch.device("cpu"),
                tqdm=lambda it, **k: it, torchvision=tv)


def _lines(path, ranges):
    src = open(path, encoding="utf-8").read().split("\n")
    return "\n".join("\n".join(src[a - 1:b]) for a, b in ranges)


def _nb_cell(path, cell, ranges):
    nb = json.load(open(path, encoding="utf-8"))
    code = [c for c in nb["cells"]]
    src = "".join(code[cell]["source"]).split("\n")
    return "\n".join("\n".join(src[a - 1:b]) for a, b in ranges)


def load_reference(arch):
    """Returns (namespace, ModelClass, SamplerClass, compress_fn)."""
    ns = _namespace()
    if arch == spec.ARCH_WEBP:
        exec(_lines(f"{REF}/webp_inference.py", [(80, 102), (135, 473)]), ns)
        return ns, ns["WebPDiffusionModel"], ns["DDRMWebPSampler"], ns["webp_compress"]
    if arch == spec.ARCH_AVIF:
        exec(_lines(f"{REF}/avif_inference.py", [(64, 98), (102, 459)]), ns)
        return ns, ns["AVIFDiffusionModel"], ns["DDRMAVIFSampler"], ns["avif_compress"]
    exec(_nb_cell(f"{REF}/svd.ipynb", 0, [(45, 69), (102, 440)]), ns)
    return ns, ns["JPEGDiffusionModel"], ns["DDRMJPEGSampler"], ns["jpeg_compress"]


def build_model(arch, seed=0):
    import torch
    ns, Model, Sampler, comp = load_reference(arch)
    m = Model().eval()
    sd = synth.synth_state_dict(arch, seed)
    ref_sd = m.state_dict()
    assert list(ref_sd.keys()) == list(sd.keys()), "state_spec names differ from the reference"
    for k, v in ref_sd.items():
        assert tuple(v.shape) == tuple(sd[k].shape), (k, v.shape, sd[k].shape)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    n_params = sum(p.numel() for p in m.parameters())
    assert n_params == spec.param_count(arch), (n_params, spec.param_count(arch))
    return ns, m, Sampler, comp


def run_sampler(ns, model, Sampler, y, quality, steps, noise):
    """Run the reference sampler with torch.randn_like replaced by replaying noise[i] in loop order."""
    import torch
    order = list(range(steps - 1, 0, -1))
    it = iter(order)
    real = torch.randn_like

    def fake(x, *a, **k):
        return torch.from_numpy(noise[next(it)])

    torch.randn_like = fake
    try:
        out = Sampler(model).sample(torch.from_numpy(y), quality, steps=steps)
    finally:
        torch.randn_like = real
    return out.numpy()


def main():
    import torch
    if not os.path.isdir(REF):
        print("reference not present; nothing generated")
        return 0
    torch.manual_seed(0)
    torch.set_num_threads(8)
    names = {spec.ARCH_WEBP: "webp", spec.ARCH_JPEG: "jpeg", spec.ARCH_AVIF: "avif"}

    # ---------------- codec goldens (pin the Pillow / libjpeg-turbo / libwebp / libavif builds)
    codec = {}
    img32 = synth.synth_images(2, 32, seed=5)
    img64 = synth.synth_images(1, 64, seed=6)
    for arch in (spec.ARCH_WEBP, spec.ARCH_JPEG, spec.ARCH_AVIF):
        ns, Model, Sampler, comp = load_reference(arch)
        for q in (10, 20, 50):
            for tag, img in (("32", img32), ("64", img64)):
                codec[f"{names[arch]}_q{q}_{tag}"] = comp(torch.from_numpy(img), q).numpy()
    codec["in_32"], codec["in_64"] = img32, img64
    np.savez_compressed(os.path.join(HERE, "codec.npz"), **codec)

    for arch in (spec.ARCH_WEBP, spec.ARCH_JPEG, spec.ARCH_AVIF):
        nm = names[arch]
        ns, m, Sampler, comp = build_model(arch)
        out = {}
        # ---------------- forward goldens + intermediates of ONE block via hooks
        x = synth.synth_images(2, 32, seed=11)
        t = np.array([0.37, 0.81], dtype=np.float32)
        cl = np.array([0.37, 0.05], dtype=np.float32)
        taps = {}
        hooks = []
        for key in ("down2", "down2.attn", "down2.freq_guide", "bottleneck.1", "up5", "time_embed"):
            mod = m.get_submodule(key)
            hooks.append(mod.register_forward_hook(
                lambda _m, _i, o, key=key: taps.__setitem__(key, (o[0] if isinstance(o, tuple) else o).detach().numpy().copy())))
        hooks.append(m.get_submodule("down2").register_forward_hook(
            lambda _m, i, o: taps.__setitem__("down2.in", i[0].detach().numpy().copy())))
        with torch.no_grad():
            y = m(torch.from_numpy(x), torch.from_numpy(t), torch.from_numpy(cl)).numpy()
        for h in hooks:
            h.remove()
        out.update(fwd32_x=x, fwd32_t=t, fwd32_cl=cl, fwd32_y=y)
        for k_, v in taps.items():
            out["tap32_" + k_] = v
        x64 = synth.synth_images(1, 64, seed=12)
        t64 = np.array([0.5], dtype=np.float32)
        with torch.no_grad():
            y64 = m(torch.from_numpy(x64), torch.from_numpy(t64), torch.from_numpy(t64)).numpy()
        out.update(fwd64_x=x64, fwd64_t=t64, fwd64_y=y64)
        # non-square, not a power of two: 96 x 64
        x96 = synth.synth_images(1, 96, seed=13)[:, :, :, :64].copy()
        with torch.no_grad():
            y96 = m(torch.from_numpy(x96), torch.from_numpy(t64), torch.from_numpy(t64)).numpy()
        out.update(fwd96_x=x96, fwd96_y=y96)
        print(nm, "forward  |y| mean", float(np.abs(y).mean()), "max", float(np.abs(y).max()))

        # ---------------- sampler trajectories with injected noise
        q, steps, B = 10, (50 if arch == spec.ARCH_JPEG else 20), (4 if arch == spec.ARCH_JPEG else 2)
        x0 = synth.synth_images(B, 32, seed=123)
        ydeg = comp(torch.from_numpy(x0), q).numpy()
        noise = synth.synth_noise(steps, B, 32, 32, seed=7)
        # trajectory: hook the model to record x_t fed at each step and x_theta
        xs, thetas = [], []
        def _rec(_m, i, o):
            xs.append(i[0].numpy().copy())
            thetas.append(o.numpy().copy())

        hk = m.register_forward_hook(_rec)
        final = run_sampler(ns, m, Sampler, ydeg, q, steps, noise)
        hk.remove()
        keep = sorted(set(list(range(0, steps, 10)) + [steps - 1, 1, 0]))  # loop indices i to keep
        out.update(samp_x0=x0, samp_y=ydeg, samp_final=final, samp_q=np.int32(q), samp_steps=np.int32(steps),
                   samp_keep=np.array(keep, dtype=np.int32))
        for i in keep:
            idx = steps - 1 - i
            out[f"samp_xt_in_{i}"] = xs[idx]
            out[f"samp_theta_{i}"] = thetas[idx]
        # a high-quality run (no phase fix; JPEG 4:4:4) with few steps
        q2, steps2 = 50, 6
        y2 = comp(torch.from_numpy(x0[:2]), q2).numpy()
        n2 = synth.synth_noise(steps2, 2, 32, 32, seed=8)
        out.update(samp2_y=y2, samp2_final=run_sampler(ns, m, Sampler, y2, q2, steps2, n2),
                   samp2_q=np.int32(q2), samp2_steps=np.int32(steps2))
        # phase_consistency alone
        pc_a = synth.synth_images(2, 32, seed=31)
        pc_b = synth.synth_images(2, 32, seed=32)
        out["pc_x"], out["pc_ref"] = pc_a, pc_b
        out["pc_out"] = ns["phase_consistency"](torch.from_numpy(pc_a), torch.from_numpy(pc_b)).numpy()
        np.savez_compressed(os.path.join(HERE, f"{nm}.npz"), **out)
        print(nm, "sampler final mean", float(final.mean()))
        del m

    # ---------------- eps-predict DDPM loop (experiments/code/conv.ipynb cell 1)
    ns = _namespace()
    exec(_nb_cell(f"{REF}/experiments/code/conv.ipynb", 1, [(45, 105), (112, 125)]), ns)
    net = ns["UNetDenoiser"]().eval()
    sd = synth.synth_eps_unet_state(3)
    assert [(k, tuple(v.shape)) for k, v in net.state_dict().items()] == [(k, tuple(v.shape)) for k, v in sd.items()]
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    x0 = (synth.synth_images(1, 32, seed=41) * 0.5 + 0.5).astype(np.float32)
    ns["jpeg_compress"] = lambda x, quality: x          # ddpm_inference starts from jpeg_compress(x0): identity here
    ns["alpha_t_list"] = ns["alpha_t_list"]
    with torch.no_grad():
        eps1 = net(torch.from_numpy(x0), torch.tensor([[5.0]])).numpy()
        outp = ns["ddpm_inference"](net, torch.from_numpy(x0), T=8).numpy()
    np.savez_compressed(os.path.join(HERE, "ddpm_eps.npz"), x0=x0, eps_t5=eps1, out_T8=outp,
                        betas=ns["beta_t"].numpy())
    print("done")
    return 0


if __name__ == "__main__":
    sys.exit(main())
