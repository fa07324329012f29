"""Builds libddpmir.so (HIP, gfx950 only) in-tree with hipcc.  Used by __graft_entry__.build()."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libddpmir.so")
SOURCES = ["igemm.hip", "gemm16.hip", "gemm256.hip", "attention.hip", "attention_bf16.hip", "pointwise.hip", "freq_fused.hip", "head_fused.hip", "conv_tile.hip", "sampler_kernels.hip", "jpeg_sim.hip", "model.hip", "eps_unet.hip", "api.hip", "ops_api.hip", "host_codec.cpp"]
# MFMA results feed VALU (softmax) directly: keep C/D in VGPRs instead of copying through AGPRs every tile
VGPR_MFMA = {"attention_bf16.hip", "attention.hip", "gemm16.hip", "gemm256.hip", "freq_fused.hip", "head_fused.hip", "conv_tile.hip"}
# packed-f32 VALU (v_pk_add_f32) beside MFMAs costs ~3x a scalar add on gfx950: keep the softmax sums scalar
NO_SLP = {"attention_bf16.hip"}
NO_CONTRACT = {"sampler_kernels.hip"}   # bit-exact uint8 truncation path: no fma contraction
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]


def source_hash() -> str:
    """sha256/16 over csrc/* and include/*.h (names + contents, sorted): what ddpmir_version() of a fresh build reports."""
    import hashlib
    h = hashlib.sha256()
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))]
    files += [os.path.join(HERE, "..", "include", f) for f in ("ddpmir.h", "ddpmir_ops.h")]
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _hipcc_version(hipcc: str) -> str:
    try:
        out = subprocess.run([hipcc, "--version"], capture_output=True, text=True).stdout
        for ln in out.splitlines():
            if "HIP version" in ln:
                return ln.split(":", 1)[1].strip().replace(" ", "")
    except Exception:
        pass
    return "unknown"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required to build libddpmir.so)")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "ddpmir.h"), os.path.join(HERE, "..", "include", "ddpmir_ops.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True, experiments: bool = False) -> str:
    """experiments=True builds libddpmir_exp.so with -DDDPMIR_EXPERIMENTS: the A/B environment switches and extra kernel
    instantiations the measurement tools under tools/ use (select it with DDPMIR_LIB_PATH); never the shipped library."""
    if experiments:
        return _build(os.path.join(HERE, "libddpmir_exp.so"), os.path.join(HERE, "build_exp"), ["-DDDPMIR_EXPERIMENTS"], verbose)
    if not force and not needs_build():
        return LIB
    return _build(LIB, os.path.join(HERE, "build"), [], verbose)


def _file_sha(path: str) -> str:
    import hashlib
    with open(path, "rb") as fh:
        return hashlib.sha256(fh.read()).hexdigest()[:16]


def _build(lib: str, objdir: str, defines, verbose: bool) -> str:
    import json
    hipcc = _hipcc()
    hipcc_ver = _hipcc_version(hipcc)
    os.makedirs(objdir, exist_ok=True)
    procs = []
    objs = []
    # incremental by CONTENT, not mtime (a source restored with an old mtime, or a compiler upgrade, must not leave a stale
    # object behind a fresh ddpmir_version() string): an object is kept while the sidecar next to it records the same source
    # hash, header hash, flags and hipcc version
    hdrs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
    hdrs += [os.path.join(HERE, "..", "include", f) for f in ("ddpmir.h", "ddpmir_ops.h")] + [os.path.abspath(__file__)]
    hdr_sha = "".join(_file_sha(h) for h in hdrs)
    stamps = {}
    for src in SOURCES:
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        extra = ["-ffp-contract=off"] if src in NO_CONTRACT else []
        if src == "api.hip":   # ddpmir_version(): source hash + compiler version, so a stale prebuilt binary is detectable
            extra += ['-DDDPMIR_SRC_HASH="%s"' % source_hash(), '-DDDPMIR_HIPCC_VERSION="%s"' % hipcc_ver]
        if src in VGPR_MFMA:
            extra += ["-mllvm", "-amdgpu-mfma-vgpr-form"]
        if src in NO_SLP:
            extra += ["-fno-slp-vectorize"]
        stamp = {"src": _file_sha(os.path.join(CSRC, src)), "hdr": hdr_sha, "flags": [*FLAGS, *defines, *extra], "hipcc": hipcc_ver}
        stamps[obj] = stamp
        try:
            if os.path.exists(obj) and json.load(open(obj + ".stamp")) == stamp:
                continue
        except Exception:
            pass
        try:
            os.remove(obj + ".stamp")
        except OSError:
            pass
        cmd = [hipcc, *FLAGS, *defines, *extra, "-c", os.path.join(CSRC, src), "-o", obj]
        procs.append((src, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for src, obj, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"--- {src} ---\n{out}\n")
        else:
            with open(obj + ".stamp", "w") as fh:
                json.dump(stamps[obj], fh)
            if verbose and out.strip():
                sys.stderr.write(f"--- {src} ---\n{out}\n")
    if failed:
        raise RuntimeError("hipcc failed")
    subprocess.check_call([hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", lib, *objs, "-ldl", "-lpthread"])
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, experiments="--experiments" in sys.argv))
