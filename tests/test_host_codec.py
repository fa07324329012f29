"""The built-in host codec pool of libddpmir.so (csrc/host_codec.cpp: libwebp / libavif through their C APIs on a C++ thread
pool) against the reference's codec calls -- runs on CPU, no GPU needed (ddpmir_host_codec_roundtrip_u8 takes host buffers).

Reference: webp_compress (webp_inference.py:506-528), avif_compress (avif_inference.py:64-98): per image
`pil_img.save(buffer, format=..., quality=q)` then `Image.open`.  tests/golden/codec.npz holds the outputs of those very
functions (executed from the reference's source by tests/golden/make_golden.py); Pillow itself is the second witness on
larger and odd-sized images.  Bit-exact is the bar: this is integer codec work.
"""
import ctypes as C
import io

import os

import numpy as np
import pytest

from ddpm_image_restoration_amd import _lib, build, spec, synth

KIND = {"webp": spec.ARCH_WEBP, "avif": spec.ARCH_AVIF}


@pytest.fixture(scope="module")
def lib():
    build.build()
    return _lib.load()


def native(lib, kind, u8, q):
    u8 = np.ascontiguousarray(u8, dtype=np.uint8)
    out = np.empty_like(u8)
    b, _, h, w = u8.shape
    rc = lib.ddpmir_host_codec_roundtrip_u8(KIND[kind], u8.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), b, h, w, int(q))
    assert rc == 0, lib.ddpmir_last_error(None)
    return out


def pillow(kind, u8, q):
    from PIL import Image
    out = np.empty_like(u8)
    for i, img in enumerate(u8):
        buf = io.BytesIO()
        Image.fromarray(np.ascontiguousarray(img.transpose(1, 2, 0))).save(buf, format=kind.upper(), quality=q)
        buf.seek(0)
        out[i] = np.asarray(Image.open(buf).convert("RGB")).transpose(2, 0, 1)
    return out


def to_u8(x):
    return np.clip(x * np.float32(127.5) + np.float32(127.5), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("kind", ["webp", "avif"])
def test_native_codec_reproduces_the_reference_codec_goldens(kind, lib, golden):
    g = golden("codec")
    for q in (10, 20, 50):
        for tag in ("32", "64"):
            ref = g[f"{kind}_q{q}_{tag}"]                       # decoded images in [-1, 1] as the reference returns them
            got = native(lib, kind, to_u8(g["in_" + tag]), q).astype(np.float32) / np.float32(255)
            got = (got - np.float32(0.5)) * np.float32(2.0)
            assert np.array_equal(got, ref), (kind, q, tag)


@pytest.mark.parametrize("kind,qualities", [("webp", (0, 1, 5, 30, 75, 90, 100)), ("avif", (1, 5, 30, 75, 100))])
def test_native_codec_equals_pillow_bit_for_bit(kind, qualities, lib):
    rng = np.random.default_rng(5)
    imgs = [to_u8(synth.synth_images(3, 64, seed=1)),                         # a batch: the thread pool path
            to_u8(synth.synth_images(1, 256, seed=2)),                        # BASELINE's image size
            rng.integers(0, 256, (2, 3, 32, 96), dtype=np.uint8),             # white noise, non-square
            to_u8(synth.synth_images(1, 96, seed=3))[:, :, :40, :72].copy(),  # sizes that are not multiples of 16
            np.zeros((1, 3, 32, 32), np.uint8), np.full((1, 3, 32, 32), 255, np.uint8)]
    for q in qualities:
        for u8 in imgs:
            assert np.array_equal(native(lib, kind, u8, q), pillow(kind, u8, q)), (kind, q, u8.shape)


def test_quality_clamps_follow_the_reference(lib):
    """webp: clamp(int(q), 0, 100) (webp_inference.py:518); avif: clamp(int(q), 1, 100) (avif_inference.py:76)."""
    u8 = to_u8(synth.synth_images(1, 32, seed=4))
    assert np.array_equal(native(lib, "webp", u8, -5), native(lib, "webp", u8, 0))
    assert np.array_equal(native(lib, "webp", u8, 250), native(lib, "webp", u8, 100))
    assert np.array_equal(native(lib, "avif", u8, 0), native(lib, "avif", u8, 1))
    assert np.array_equal(native(lib, "avif", u8, 101), native(lib, "avif", u8, 100))


def test_bad_arguments_and_unknown_family_are_errors(lib):
    u8 = np.zeros((1, 3, 32, 32), np.uint8)
    out = np.empty_like(u8)
    assert lib.ddpmir_host_codec_roundtrip_u8(spec.ARCH_JPEG, u8.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), 1, 32, 32, 10) < 0
    assert b"device codec" in lib.ddpmir_last_error(None)
    assert lib.ddpmir_host_codec_roundtrip_u8(spec.ARCH_WEBP, None, out.ctypes.data_as(C.c_void_p), 1, 32, 32, 10) < 0
    assert lib.ddpmir_set_codec_library(7, b"/nowhere.so") < 0


def test_codec_pool_is_sized_per_rank_of_the_host():
    """One rank per GPU: every rank owns a codec pool over the SAME host cores, so the pool takes cpus / LOCAL_WORLD_SIZE
    threads (torch.distributed.run exports it), at most 32; DDPMIR_CODEC_THREADS overrides.  Read in a fresh process each."""
    import subprocess
    import sys
    code = ("import os, sys; sys.path.insert(0, %r); from ddpm_image_restoration_amd import _lib; "
            "print(_lib.load().ddpmir_host_codec_threads())" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

    def threads(**env):
        e = {k: v for k, v in os.environ.items() if k not in ("LOCAL_WORLD_SIZE", "DDPMIR_CODEC_THREADS")}
        e.update(env)
        return int(subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, check=True).stdout.split()[-1])
    cpus = len(os.sched_getaffinity(0))
    assert threads() == max(1, min(cpus, 32))
    assert threads(LOCAL_WORLD_SIZE="4") == max(1, min(cpus // 4, 32))
    assert threads(LOCAL_WORLD_SIZE="1024") == 1
    assert threads(LOCAL_WORLD_SIZE="8", DDPMIR_CODEC_THREADS="3") == 3


def test_sanitizer_build_of_the_host_codec_pool(tmp_path):
    """SURVEY.md section 5 (race detection / sanitizers): the host-side native code (thread pool + codec bindings) built
    with gcc -fsanitize=address,undefined and driven from two threads at once; any report fails the run."""
    import glob
    import os
    import shutil
    import subprocess
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("g++ not available")
    import PIL
    libs = os.path.join(os.path.dirname(os.path.dirname(PIL.__file__)), "pillow.libs")
    webp, avif = sorted(glob.glob(os.path.join(libs, "libwebp-*.so*"))), sorted(glob.glob(os.path.join(libs, "libavif-*.so*")))
    if not webp or not avif:
        pytest.skip("Pillow's bundled codec libraries not found")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "host_codec_asan")
    subprocess.run([gxx, "-std=c++17", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                    os.path.join(root, "tests", "c_abi", "host_codec_asan.cpp"),
                    os.path.join(root, "ddpm_image_restoration_amd", "csrc", "host_codec.cpp"), "-o", exe, "-ldl", "-lpthread"],
                   check=True, capture_output=True, text=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1")     # the codec libraries keep global tables
    r = subprocess.run([exe, webp[0], avif[0]], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "ERROR" not in r.stderr and "runtime error" not in r.stderr, r.stdout + r.stderr
