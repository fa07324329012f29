"""Host codec round trips (the degradation operator y = codec(x, q)).

Same calls as the reference's webp_compress / jpeg_compress / avif_compress (webp_inference.py:506-528,
svd.ipynb c0:45-69, avif_inference.py:64-98): Pillow encode -> decode per image, same quality clamps, JPEG
subsampling 4:4:4 above q30 else 4:2:0, AVIF silently falling back to JPEG on an encoder error.  Images move as
uint8 NCHW; a thread pool spreads the per-image work over the host cores (Pillow releases the GIL inside libjpeg /
libwebp: 16 WebP images take 83 ms sequentially and 12 ms from 16 threads on the MI355X host).  Pillow's AVIF encoder
does not scale across threads (591 ms -> 454 ms), so AVIF batches go to a pool of worker PROCESSES over shared memory
(`_codec_worker.py`; DDPMIR_AVIF_WORKERS=0 disables it, =N sets the worker count).
"""
from __future__ import annotations

import atexit
import io
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

_pool = None
_procs = None


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


class _ProcPool:
    """Worker processes + two shared-memory blocks (batch in / batch out); see _codec_worker.py."""

    def __init__(self, workers: int, nbytes: int):
        from multiprocessing import shared_memory
        self.nbytes = nbytes
        self.shm_in = shared_memory.SharedMemory(create=True, size=nbytes)
        self.shm_out = shared_memory.SharedMemory(create=True, size=nbytes)
        env = dict(os.environ)
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
        cmd = [sys.executable, "-m", "ddpm_image_restoration_amd._codec_worker", self.shm_in.name, self.shm_out.name]
        self.procs = [subprocess.Popen(cmd, stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, bufsize=1, env=env)
                      for _ in range(workers)]

    def run(self, u8: np.ndarray, quality: int, kind: str, out: np.ndarray) -> None:
        b, _, h, w = u8.shape
        np.ndarray(u8.shape, dtype=np.uint8, buffer=self.shm_in.buf)[...] = u8
        n = len(self.procs)
        for i in range(b):   # image i -> worker i mod n; every worker serves its requests in order
            self.procs[i % n].stdin.write(f"{i} {h} {w} {int(quality)} {kind}\n")
        for p in self.procs[:min(n, b)]:
            p.stdin.flush()
        for i in range(b):
            reply = self.procs[i % n].stdout.readline().strip()
            if reply != "ok":
                raise RuntimeError(f"codec worker failed on image {i}: {reply or 'worker exited'}")
        out[...] = np.ndarray(u8.shape, dtype=np.uint8, buffer=self.shm_out.buf)

    def close(self) -> None:
        for p in self.procs:
            try:
                p.stdin.write("quit\n"); p.stdin.flush(); p.stdin.close()
            except Exception:
                pass
        for p in self.procs:
            try:
                p.wait(timeout=5)
            except Exception:
                p.kill()
        for shm in (self.shm_in, self.shm_out):
            try:
                shm.close(); shm.unlink()
            except Exception:
                pass


def _avif_workers(batch: int) -> int:
    env = os.environ.get("DDPMIR_AVIF_WORKERS")
    if env is not None:
        return max(0, int(env))
    return max(1, min(batch, 16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 4)))


def _get_procs(nbytes: int, batch: int):
    global _procs
    workers = _avif_workers(batch)
    if workers == 0:
        return None
    if _procs is not None and (_procs.nbytes < nbytes or len(_procs.procs) < workers):
        _procs.close()
        _procs = None
    if _procs is None:
        _procs = _ProcPool(workers, nbytes)
        atexit.register(_shutdown_procs)
    return _procs


def _shutdown_procs() -> None:
    global _procs
    if _procs is not None:
        _procs.close()
        _procs = None


def native_roundtrip_u8(u8_nchw: np.ndarray, quality: int, kind: str, out: np.ndarray | None = None):
    """The library's built-in host codec pool (csrc/host_codec.cpp: libwebp / libavif C APIs on a C++ thread pool; the same
    bits as the Pillow path, tests/test_host_codec.py).  Returns None when that codec is unavailable."""
    import ctypes as C
    from . import _lib, spec
    arch = {"webp": spec.ARCH_WEBP, "avif": spec.ARCH_AVIF}.get(kind)
    if arch is None:
        return None
    lib = _lib.load()
    src = np.ascontiguousarray(u8_nchw, dtype=np.uint8)
    dst = out if (out is not None and out.flags["C_CONTIGUOUS"] and out.dtype == np.uint8) else np.empty_like(src)
    b, _, h, w = src.shape
    rc = lib.ddpmir_host_codec_roundtrip_u8(arch, src.ctypes.data_as(C.c_void_p), dst.ctypes.data_as(C.c_void_p), b, h, w, int(quality))
    if rc != 0:
        return None
    if out is not None and dst is not out:
        out[...] = dst
        return out
    return dst


def roundtrip_u8(u8_nchw: np.ndarray, quality: int, kind: str, out: np.ndarray | None = None, backend: str = "pillow") -> np.ndarray:
    """decode(encode(img, quality)) for every image of a uint8 [B,3,H,W] batch.  backend="native" uses the library's codec
    pool where it exists (WebP / AVIF) and falls back to the Pillow calls below (e.g. the reference's AVIF -> JPEG fallback
    on an encoder error lives there)."""
    b = u8_nchw.shape[0]
    if backend == "native":
        res = native_roundtrip_u8(u8_nchw, quality, kind, out)
        if res is not None:
            return res
    if out is None:
        out = np.empty_like(u8_nchw)
    if b == 1:
        out[0] = _one(u8_nchw[0], quality, kind)
        return out
    if kind == "avif":
        pool = _get_procs(u8_nchw.nbytes, b)
        if pool is not None:
            pool.run(np.ascontiguousarray(u8_nchw), quality, kind, out)
            return out
    for i, res in enumerate(_get_pool().map(lambda im: _one(im, quality, kind), list(u8_nchw))):
        out[i] = res
    return out


def _compress(x, quality: int, kind: str):
    """[-1,1] float NCHW torch tensor -> decoded [-1,1] tensor on the same device (reference semantics)."""
    import torch
    u8 = (x * 127.5 + 127.5).clamp(0, 255).to(torch.uint8).cpu().numpy()
    dec = torch.from_numpy(roundtrip_u8(u8, quality, kind, backend=os.environ.get("DDPMIR_HOST_CODEC", "native")))
    return dec.to(x.device).float().div(255).sub(0.5).mul(2.0)


def webp_compress(x, quality):
    return _compress(x, quality, "webp")


def jpeg_compress(x, quality):
    return _compress(x, quality, "jpeg")


def avif_compress(x, quality):
    return _compress(x, quality, "avif")
