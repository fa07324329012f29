"""Drop-in samplers: DDRM{WebP,JPEG,AVIF}Sampler(model).sample(x_t, quality, steps, eta, eta_b).

Reference: webp_inference.py:553-602, svd.ipynb c0:391-440, avif_inference.py:410-459.  The T-step loop runs
inside libddpmir.so (ddpmir_ddrm_sample_loop).  The codec round trip of each step is built into the library: JPEG on the
device (bit-exact libjpeg-turbo simulator), WebP / AVIF through libwebp / libavif's C APIs on a C++ thread pool (the
copies bundled with this interpreter's Pillow, bit-identical to the reference's PIL calls).  host_codec="pillow" routes
it through a callback into Pillow instead -- literally the library calls the reference makes.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib, codec, spec


class _DDRMSampler:
    KIND = "webp"

    def __init__(self, model):
        self.model = model
        self._lib = _lib.load()

    def _codec_cb(self):
        kind = self.KIND

        def cb(_user, in_p, out_p, b, h, w, quality):
            try:
                n = b * 3 * h * w
                src = np.ctypeslib.as_array(in_p, shape=(n,)).reshape(b, 3, h, w)
                dst = np.ctypeslib.as_array(out_p, shape=(n,)).reshape(b, 3, h, w)
                codec.roundtrip_u8(src, quality, kind, out=dst)
                return 0
            except Exception:  # never unwind through the C frame
                import traceback
                traceback.print_exc()
                return 1

        return _lib.CODEC_FN(cb)

    def sample(self, x_t, quality, steps=100, eta=0.85, eta_b=1.0, noise=None, seed=0, image_offset=0,
               trace=False, device_codec=None, start_step=-1, num_steps=0, host_codec=None):
        """noise: optional tensor [steps,B,3,H,W]; noise[i] replaces torch.randn_like at loop index i.
        trace=True additionally returns a [steps,2,B,3,H,W] tensor of (x_theta, x_t) per executed step.
        device_codec (JPEG): None/True = on-device codec, False = host codec.  host_codec: "native" (the library's
        libwebp / libavif pool; default for WebP / AVIF) or "pillow" (callback into Pillow; the only host JPEG codec)."""
        import torch
        m = self.model
        m.eval()
        x = x_t.to(m.device, torch.float32).contiguous()
        b, c, h, w = x.shape
        out = torch.empty_like(x)
        opts = _lib.LoopOpts()
        cb = self._codec_cb()
        # JPEG has a bit-exact device codec (no host round trip); WebP / AVIF use the library's host pool or the callback
        use_dev = (self.KIND == "jpeg") if device_codec is None else bool(device_codec)
        if host_codec is None:
            import os
            host_codec = os.environ.get("DDPMIR_HOST_CODEC", "native")
        if host_codec not in ("native", "pillow"):
            raise ValueError("host_codec must be 'native' or 'pillow'")
        builtin = (use_dev and self.KIND == "jpeg") or (host_codec == "native" and self.KIND in ("webp", "avif"))
        opts.codec = _lib.CODEC_FN(0) if builtin else cb
        opts.codec_user = None
        nz = None
        if noise is not None:
            nz = noise.to(m.device, torch.float32).contiguous()
            if tuple(nz.shape) != (steps, b, c, h, w):
                raise RuntimeError("noise must be [steps,B,3,H,W]")
            opts.noise = nz.data_ptr()
        opts.seed = int(seed)
        opts.image_offset = int(image_offset)
        tr = None
        if trace:
            tr = torch.empty((num_steps if num_steps > 0 else steps, 2, b, c, h, w), device=m.device,
                             dtype=torch.float32)
            opts.trace = tr.data_ptr()
        opts.noise_coeff = -1.0
        opts.phase_q = -1
        opts.phase_every = 0
        opts.phase_alpha = -1.0
        opts.start_step = int(start_step)
        opts.num_steps = int(num_steps)
        torch.cuda.current_stream(m.device).synchronize()
        rc = self._lib.ddpmir_ddrm_sample_loop(m._ctx, x.data_ptr(), out.data_ptr(), b, h, w, int(quality), int(steps),
                                               float(eta), float(eta_b), C.byref(opts))
        _lib.check(self._lib, m._ctx, rc, "ddrm_sample_loop")
        return (out, tr) if trace else out


class DDRMWebPSampler(_DDRMSampler):
    KIND = "webp"


class DDRMJPEGSampler(_DDRMSampler):
    KIND = "jpeg"


class DDRMAVIFSampler(_DDRMSampler):
    KIND = "avif"


SAMPLER_BY_ARCH = {spec.ARCH_WEBP: DDRMWebPSampler, spec.ARCH_JPEG: DDRMJPEGSampler, spec.ARCH_AVIF: DDRMAVIFSampler}


def phase_consistency(model, x, ref, alpha=0.7):
    """webp_inference.py:531-550 on the device."""
    import torch
    lib = _lib.load()
    x = x.to(model.device, torch.float32).contiguous()
    ref = ref.to(model.device, torch.float32).contiguous()
    out = torch.empty_like(x)
    b, c, h, w = x.shape
    torch.cuda.current_stream(model.device).synchronize()
    rc = lib.ddpmir_phase_consistency(model._ctx, x.data_ptr(), ref.data_ptr(), out.data_ptr(), b, h, w, float(alpha))
    _lib.check(lib, model._ctx, rc, "phase_consistency")
    return out


def jpeg_roundtrip_device(model, u8_nchw, quality):
    """decode(encode(img)) of a uint8 [B,3,H,W] CUDA tensor with the on-device libjpeg-turbo simulator."""
    import torch
    lib = _lib.load()
    x = u8_nchw.to(model.device, torch.uint8).contiguous()
    out = torch.empty_like(x)
    b, c, h, w = x.shape
    torch.cuda.current_stream(model.device).synchronize()
    rc = lib.ddpmir_jpeg_roundtrip_u8(model._ctx, x.data_ptr(), out.data_ptr(), b, h, w, int(quality))
    _lib.check(lib, model._ctx, rc, "jpeg_roundtrip_u8")
    return out
