import ctypes as C, glob, numpy as np, time, sys, threading
from concurrent.futures import ProcessPoolExecutor, ThreadPoolExecutor
d='/usr/local/lib/python3.10/dist-packages/pillow.libs/'
C.CDLL(glob.glob(d+'libsharpyuv-*.so*')[0], mode=C.RTLD_GLOBAL)
lib = C.CDLL(glob.glob(d+'libwebp-*.so*')[0])
lib.WebPEncodeRGB.restype = C.c_size_t
lib.WebPEncodeRGB.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.POINTER(C.c_void_p)]
lib.WebPDecodeRGBInto.restype = C.c_void_p
lib.WebPDecodeRGBInto.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
lib.WebPFree.argtypes = [C.c_void_p]
a = np.ascontiguousarray(np.clip(np.random.default_rng(0).normal(128, 40, (256, 256, 3)), 0, 255).astype(np.uint8))
def c_rt(_):
    t0 = time.time()
    out = C.c_void_p()
    n = lib.WebPEncodeRGB(a.ctypes.data, 256, 256, 768, 10.0, C.byref(out))
    dec = np.empty_like(a)
    lib.WebPDecodeRGBInto(out, n, dec.ctypes.data, dec.nbytes, 768)
    lib.WebPFree(out)
    return time.time() - t0
if __name__ == "__main__":
    c_rt(0)
    t = time.time(); per = [c_rt(0) for _ in range(16)]; print("seq total", round((time.time() - t)*1e3), "per-call", round(np.mean(per)*1e3, 1))
    for nt in (2, 4, 8):
        with ThreadPoolExecutor(nt) as ex:
            t = time.time(); per = list(ex.map(c_rt, range(16))); print(nt, "threads total", round((time.time() - t)*1e3), "per-call", round(np.mean(per)*1e3, 1))
    with ProcessPoolExecutor(8) as ex:
        list(ex.map(c_rt, range(8)))
        t = time.time(); per = list(ex.map(c_rt, range(16))); print("8 procs total", round((time.time() - t)*1e3), "per-call", round(np.mean(per)*1e3, 1))
    import io
    from PIL import Image
    def pil_rt(fmt):
        def f(_):
            t0 = time.time()
            buf = io.BytesIO(); Image.fromarray(a).save(buf, format=fmt, quality=10); buf.seek(0)
            np.asarray(Image.open(buf).convert("RGB"))
            return time.time() - t0
        return f
    import os
    print("cpus", os.cpu_count(), len(os.sched_getaffinity(0)))
    for fmt in ("WEBP", "AVIF"):
        f = pil_rt(fmt); f(0)
        t = time.time(); per = [f(0) for _ in range(16)]; print(fmt, "pillow seq total", round((time.time() - t)*1e3), "per-call", round(np.mean(per)*1e3, 1))
        for nt in (4, 16):
            with ThreadPoolExecutor(nt) as ex:
                t = time.time(); per = list(ex.map(f, range(16))); print(fmt, nt, "threads total", round((time.time() - t)*1e3), "per-call", round(np.mean(per)*1e3, 1))
