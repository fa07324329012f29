"""Host codec round trips (the degradation operator y = codec(x, q)).

Same calls as the reference's webp_compress / jpeg_compress / avif_compress (webp_inference.py:506-528,
svd.ipynb c0:45-69, avif_inference.py:64-98): Pillow encode -> decode per image, same quality clamps, JPEG
subsampling 4:4:4 above q30 else 4:2:0, AVIF silently falling back to JPEG on an encoder error.  Images move as
uint8 NCHW; a thread pool spreads the per-image work over the host cores (Pillow releases the GIL inside the
codec libraries).
"""
from __future__ import annotations

import io
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

_pool = None


def _get_pool() -> ThreadPoolExecutor:
    global _pool
    if _pool is None:
        _pool = ThreadPoolExecutor(max_workers=max(1, min(32, (os.cpu_count() or 4))))
    return _pool


def _one(img_chw: np.ndarray, quality: int, kind: str) -> np.ndarray:
    from PIL import Image
    pil = Image.fromarray(np.ascontiguousarray(img_chw.transpose(1, 2, 0)))
    buf = io.BytesIO()
    if kind == "webp":
        pil.save(buf, format="WEBP", quality=max(0, min(100, int(quality))))
    elif kind == "jpeg":
        q = max(1, min(100, int(quality)))
        pil.save(buf, format="JPEG", quality=q, subsampling="4:4:4" if q > 30 else "4:2:0")
    elif kind == "avif":
        q = max(1, min(100, int(quality)))
        try:
            pil.save(buf, format="AVIF", quality=q)
        except Exception:
            buf = io.BytesIO()
            pil.save(buf, format="JPEG", quality=q)
    else:
        raise ValueError(f"unknown codec {kind!r}")
    buf.seek(0)
    return np.asarray(Image.open(buf).convert("RGB"), dtype=np.uint8).transpose(2, 0, 1)


def roundtrip_u8(u8_nchw: np.ndarray, quality: int, kind: str, out: np.ndarray | None = None) -> np.ndarray:
    """decode(encode(img, quality)) for every image of a uint8 [B,3,H,W] batch."""
    b = u8_nchw.shape[0]
    if out is None:
        out = np.empty_like(u8_nchw)
    if b == 1:
        out[0] = _one(u8_nchw[0], quality, kind)
        return out
    for i, res in enumerate(_get_pool().map(lambda im: _one(im, quality, kind), list(u8_nchw))):
        out[i] = res
    return out


def _compress(x, quality: int, kind: str):
    """[-1,1] float NCHW torch tensor -> decoded [-1,1] tensor on the same device (reference semantics)."""
    import torch
    u8 = (x * 127.5 + 127.5).clamp(0, 255).to(torch.uint8).cpu().numpy()
    dec = torch.from_numpy(roundtrip_u8(u8, quality, kind))
    return dec.to(x.device).float().div(255).sub(0.5).mul(2.0)


def webp_compress(x, quality):
    return _compress(x, quality, "webp")


def jpeg_compress(x, quality):
    return _compress(x, quality, "jpeg")


def avif_compress(x, quality):
    return _compress(x, quality, "avif")
