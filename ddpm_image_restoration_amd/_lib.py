"""ctypes binding of libddpmir.so (include/ddpmir.h).  No fallback: a missing library is an error."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DDPMIR_LIB_PATH") or os.path.join(HERE, "libddpmir.so")   # override: A/B of two builds

CODEC_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_int,
                       C.c_int)


class Config(C.Structure):
    _fields_ = [("arch", C.c_int), ("dtype", C.c_int), ("attn_min_level", C.c_int), ("device", C.c_int),
                ("use_graph", C.c_int)]


class LoopOpts(C.Structure):
    _fields_ = [("codec", CODEC_FN), ("codec_user", C.c_void_p), ("noise", C.c_void_p), ("seed", C.c_uint64),
                ("image_offset", C.c_uint64), ("trace", C.c_void_p), ("noise_coeff", C.c_double),
                ("phase_q", C.c_int), ("phase_every", C.c_int), ("phase_alpha", C.c_double),
                ("start_step", C.c_int), ("num_steps", C.c_int)]


# every symbol include/ddpmir.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "ddpmir_create": (C.c_void_p, [C.POINTER(Config)]),
    "ddpmir_destroy": (None, [C.c_void_p]),
    "ddpmir_last_error": (C.c_char_p, [C.c_void_p]),
    "ddpmir_load_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.c_int]),
    "ddpmir_weight_layout_version": (C.c_int, []),
    "ddpmir_set_codec_library": (C.c_int, [C.c_int, C.c_char_p]),
    "ddpmir_host_codec_roundtrip_u8": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "ddpmir_host_codec_threads": (C.c_int, []),
    "ddpmir_weight_blob_bytes": (C.c_size_t, [C.c_void_p]),
    "ddpmir_finalize_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "ddpmir_param_count": (C.c_int64, [C.c_void_p]),
    "ddpmir_predict": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                 C.c_int, C.c_void_p]),
    "ddpmir_ddrm_sample_loop": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.c_double, C.c_double, C.POINTER(LoopOpts)]),
    "ddpmir_create_eps_unet": (C.c_void_p, [C.c_int]),
    "ddpmir_predict_noise": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "ddpmir_p_sample_loop": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_void_p, C.c_int]),
    "ddpmir_phase_consistency": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                           C.c_double]),
    "ddpmir_jpeg_roundtrip_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "ddpmir_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "ddpmir_profile_read": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int64),
                                      C.POINTER(C.c_double)]),
    "ddpmir_version": (C.c_char_p, []),
    "ddpmir_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_double]),
}

_lib = None


def _preload_hip_runtime():
    """libddpmir.so takes device pointers from torch tensors, so both must sit on ONE HIP runtime.  The torch
    wheel bundles its own libamdhip64 (same SONAME as /opt/rocm's): load that copy first so the dynamic linker
    binds libddpmir.so's libamdhip64.so.7 dependency to it instead of opening a second runtime."""
    try:
        import torch
    except ImportError:
        return   # standalone use: the system ROCm runtime is picked up through DT_NEEDED
    cand = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def load() -> C.CDLL:
    """Load libddpmir.so and bind every declared symbol.  Raises if the library or a symbol is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc, gfx950).  There is no CPU fallback for the restoration path.")
    _preload_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    _bind_pillow_codec_libraries(lib)
    return lib


def _bind_pillow_codec_libraries(lib) -> None:
    """Point the built-in WebP / AVIF host codecs at the libwebp / libavif copies bundled with this interpreter's Pillow
    (pillow.libs/), so the library's codec round trips are those of the reference's `pil_img.save(...)` calls with the very
    same codec build.  Without Pillow the library falls back to $DDPMIR_LIBWEBP / $DDPMIR_LIBAVIF or the system's copies."""
    import glob
    try:
        import PIL
    except ImportError:
        return
    libs = os.path.join(os.path.dirname(os.path.dirname(PIL.__file__)), "pillow.libs")
    for arch, pat in ((0, "libwebp-*.so*"), (2, "libavif-*.so*")):
        hits = sorted(glob.glob(os.path.join(libs, pat)))
        if hits:
            lib.ddpmir_set_codec_library(arch, hits[0].encode())


def check(lib, ctx, rc: int, what: str):
    if rc != 0:
        msg = lib.ddpmir_last_error(ctx)
        raise RuntimeError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")
