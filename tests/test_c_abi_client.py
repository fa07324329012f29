"""The drop-in boundary from the other side: a plain-C program (tests/c_abi/client.c, gcc, no Python / torch in the
process) drives libddpmir.so through include/ddpmir.h and must reproduce what the Python binding returns."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

from ddpm_image_restoration_amd import spec, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "ddpm_image_restoration_amd")


def _build_client(tmp_path):
    gcc = shutil.which("gcc")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    if gcc is None or not os.path.exists(os.path.join(rocm, "include", "hip", "hip_runtime_api.h")):
        pytest.skip("gcc or the HIP headers are not available")
    exe = str(tmp_path / "ddpmir_c_client")
    cmd = [gcc, "-O1", "-std=c11", "-Wall", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(rocm, "include"),
           os.path.join(ROOT, "tests", "c_abi", "client.c"), "-o", exe, "-L", LIB, "-lddpmir",
           "-L", os.path.join(rocm, "lib"), "-lamdhip64", "-Wl,-rpath," + LIB, "-Wl,-rpath," + os.path.join(rocm, "lib")]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


def test_c_client_compiles_against_the_public_header(tmp_path):
    """CPU part: the header is valid C11 and every symbol the client uses links against the in-tree library."""
    if not os.path.exists(os.path.join(LIB, "libddpmir.so")):
        pytest.skip("libddpmir.so not built")
    _build_client(tmp_path)


def _pillow_codec_env():
    import glob
    import PIL
    libs = os.path.join(os.path.dirname(os.path.dirname(PIL.__file__)), "pillow.libs")
    env = dict(os.environ)
    for var, pat in (("DDPMIR_LIBWEBP", "libwebp-*.so*"), ("DDPMIR_LIBAVIF", "libavif-*.so*")):
        hits = sorted(glob.glob(os.path.join(libs, pat)))
        if hits:
            env[var] = hits[0]
    return env


@pytest.mark.gpu
@pytest.mark.parametrize("arch,name", [(spec.ARCH_JPEG, "jpeg"), (spec.ARCH_WEBP, "webp")])
def test_c_client_matches_python_binding(arch, name, tmp_path):
    """JPEG: device codec.  WebP: the sampling loop with NO codec callback -- the library's built-in libwebp pool -- from a
    process that has neither Python nor Pillow in it; equal, bit for bit, to the Python binding with the Pillow callback."""
    import torch
    from ddpm_image_restoration_amd.model import MODEL_BY_ARCH
    from ddpm_image_restoration_amd.sampler import SAMPLER_BY_ARCH
    exe = _build_client(tmp_path)
    sd = synth.synth_state_dict(arch, 0)
    wpath, ipath, opath = str(tmp_path / "w.bin"), str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(wpath, "wb") as f:
        for name, v in sd.items():
            a = np.ascontiguousarray(v, dtype=np.float32)
            nb = name.encode()
            f.write(struct.pack("<i", len(nb))); f.write(nb)
            f.write(struct.pack("<i", a.ndim)); f.write(np.asarray(a.shape, dtype=np.int64).tobytes())
            f.write(a.tobytes())
    B, H, W, q, steps = 2, 32, 32, 10, 3
    x = synth.synth_images(B, H, seed=77)
    t = np.array([0.25, 0.75], np.float32)
    with open(ipath, "wb") as f:
        f.write(struct.pack("<5i", B, H, W, q, steps)); f.write(x.tobytes()); f.write(t.tobytes()); f.write(t.tobytes())
    r = subprocess.run([exe, wpath, ipath, opath, str(arch)], capture_output=True, text=True, timeout=300, env=_pillow_codec_env())
    assert r.returncode == 0, r.stdout + r.stderr
    assert "params 114398409" in r.stdout and "expected error" in r.stdout
    got = np.fromfile(opath, dtype=np.float32).reshape(2, B, 3, H, W)
    m = MODEL_BY_ARCH[arch](device=torch.device("cuda", 0))
    m.load_state_dict(sd)
    xt = torch.from_numpy(x).cuda()
    tt = torch.from_numpy(t).cuda()
    assert np.array_equal(got[0], m(xt, tt, tt).cpu().numpy())
    want = SAMPLER_BY_ARCH[arch](m).sample(xt, q, steps=steps, seed=7, host_codec="pillow").cpu().numpy()
    assert np.array_equal(got[1], want)
