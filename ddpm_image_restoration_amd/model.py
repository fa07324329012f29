"""Drop-in model objects: same call signature as the reference's nn.Modules, backed by libddpmir.so.

    model = WebPDiffusionModel().to(device)            # webp_inference.py:618
    model.load_state_dict(ckpt['model_state_dict'])    # webp_inference.py:621-627
    x_theta = model(x_t, t, compression_level)         # webp_inference.py:577

torch is used for device memory and checkpoint unpickling only; all arithmetic runs in the HIP library.
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict

import numpy as np

from . import _lib, spec


def _torch():
    import torch
    return torch


def _host_array(v):
    """(contiguous host ndarray, DDPMIR dtype code) of a state-dict entry.  bf16 / fp16 checkpoints cross the C ABI in their
    storage type (ddpmir_load_tensor widens them exactly); everything else is converted to float32 here."""
    if hasattr(v, "detach"):
        torch = _torch()
        t = v.detach().cpu().contiguous()
        if t.dtype == torch.bfloat16:
            return t.view(torch.int16).numpy(), 1
        if t.dtype == torch.float16:
            return t.numpy(), 2
        return np.ascontiguousarray(t.float().numpy(), dtype=np.float32), 0
    a = np.asarray(v)
    if a.dtype == np.float16:
        return np.ascontiguousarray(a), 2
    return np.ascontiguousarray(a, dtype=np.float32), 0


PACKED_MAGIC = b"DDPMIRW\x00"


class _RestorationModel:
    ARCH = spec.ARCH_WEBP

    def __init__(self, device=None, dtype: str = "f32", attn_levels: str = "all", use_graph: bool = False,
                 fp8_min_channels: int | None = None):
        """attn_levels: 'all' = self-attention in every block (the reference); 'bottleneck' = only the three
        bottleneck blocks (what BASELINE.json's configs describe; no reference parity claim).
        fp8_min_channels (dtype 'fp8'): 3x3 convs with at least this many input and output channels take e4m3 operands
        (64, 128 or 256; None = the library's default)."""
        torch = _torch()
        self._lib = _lib.load()
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
        if device is None:
            raise RuntimeError("no HIP device: the restoration model has no CPU path")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("the restoration model runs on a HIP device only (got %s)" % self.device)
        self._cfg = _lib.Config(self.ARCH, {"f32": 0, "bf16": 1, "fp8": 3}[dtype], 0 if attn_levels == "all" else 5,
                                self.device.index or 0, int(use_graph))
        self._ctx = self._lib.ddpmir_create(C.byref(self._cfg))
        if not self._ctx:
            raise RuntimeError("ddpmir_create failed: " + self._lib.ddpmir_last_error(None).decode())
        self._loaded = False
        self._blob = None
        self.training = False
        if fp8_min_channels is not None:
            self.set_option("fp8_min_channels", fp8_min_channels)

    # ---- nn.Module look-alikes used by the reference call sites
    def to(self, device):
        torch = _torch()
        d = torch.device(device)
        if d.type != "cuda":
            raise RuntimeError("the restoration model runs on a HIP device only")
        if d.index is not None and d.index != (self.device.index or 0):
            raise RuntimeError(f"this model lives on {self.device}; create it with device={d} instead of moving it "
                               "(weights and workspace are bound to one GPU)")
        return self

    def eval(self):
        self.training = False
        return self

    def train(self, mode: bool = True):
        if mode:
            raise RuntimeError("inference-only model: training stays on stock PyTorch")
        return self

    def parameters(self):
        """Yields shape-only stand-ins so `sum(p.numel() for p in model.parameters())` prints the count."""
        class _P:
            def __init__(self, n): self._n = n
            def numel(self): return self._n
        for name, shp in spec.state_spec(self.ARCH).items():
            if not name.endswith("dct_matrix"):
                yield _P(int(np.prod(shp)))

    def load_state_dict(self, state, strict: bool = True):
        """Accepts name -> torch.Tensor | ndarray.  Every tensor is handed to ddpmir_load_tensor by name."""
        if self._loaded:
            raise RuntimeError("weights already finalized for this model object")
        want = spec.state_spec(self.ARCH)
        missing = [k for k in want if k not in state]
        unexpected = [k for k in state if k not in want]
        if strict and (missing or unexpected):
            raise RuntimeError(f"state_dict mismatch: missing {missing[:5]}... unexpected {unexpected[:5]}...")
        for name in want:
            if name in state:
                self._load_one(name, state[name])
        self.finalize()
        return self

    def _load_one(self, name, v):
        want = spec.state_spec(self.ARCH)
        arr, code = _host_array(v)
        if name not in want or tuple(arr.shape) != tuple(want[name]):
            raise RuntimeError(f"unexpected tensor or shape for {name}: {tuple(arr.shape)}")
        shape = (C.c_int64 * arr.ndim)(*arr.shape)
        rc = self._lib.ddpmir_load_tensor(self._ctx, name.encode(), arr.ctypes.data_as(C.c_void_p), code, shape, arr.ndim)
        _lib.check(self._lib, self._ctx, rc, f"load_tensor({name})")

    def finalize(self, blob=None, fill: bool = True):
        """blob: optional torch uint8 CUDA tensor of weight_blob_bytes() bytes that will hold the packed weights
        (used for the RCCL broadcast in dist.py)."""
        if self._loaded:
            raise RuntimeError("weights already finalized for this model object")
        ptr = None
        if blob is not None:
            if blob.device != self.device or blob.numel() < self.weight_blob_bytes():
                raise RuntimeError("finalize: the blob must be a uint8 tensor of weight_blob_bytes() bytes on the model's device")
            self._blob = blob
            ptr = C.c_void_p(blob.data_ptr())
        rc = self._lib.ddpmir_finalize_weights(self._ctx, ptr, int(fill))
        _lib.check(self._lib, self._ctx, rc, "finalize_weights")
        self._loaded = True

    def save_packed(self, path: str):
        """Checkpoint converter output: the packed, kernel-ready weight blob (OIHW -> [tap][Cout][Cin], bf16 copies,
        concatenated time projections) of a model whose state dict was loaded and validated by name."""
        torch = _torch()
        if not self._loaded:
            raise RuntimeError("load_state_dict() first")
        if self._blob is None:
            raise RuntimeError("save_packed needs the blob to be caller-owned: use convert_checkpoint()")
        np.save(path, np.concatenate([self._packed_header(), self._blob.cpu().numpy()]), allow_pickle=False)

    def _packed_header(self) -> np.ndarray:
        """64 bytes in front of a packed blob file: magic, blob layout version, model family, arithmetic mode,
        attn_min_level, payload byte count -- a blob only loads into the configuration it was packed for."""
        import struct
        h = PACKED_MAGIC + struct.pack("<4iq", self._lib.ddpmir_weight_layout_version(), self._cfg.arch, self._cfg.dtype,
                                       self._cfg.attn_min_level, self.weight_blob_bytes())
        return np.frombuffer(h.ljust(64, b"\0"), dtype=np.uint8)

    def load_packed(self, path: str):
        """Adopt a blob written by save_packed(): no per-tensor repacking.  The file's header must name this model's
        family, arithmetic mode, attention levels and the library's current blob layout version."""
        torch = _torch()
        arr = np.load(path if path.endswith(".npy") else path + ".npy", allow_pickle=False)
        want = self._packed_header()
        if arr.dtype != np.uint8 or arr.size < 64 or bytes(arr[:8]) != PACKED_MAGIC:
            raise RuntimeError("not a packed weight blob written by save_packed()")
        if not np.array_equal(arr[:64], want):
            import struct
            got = struct.unpack("<4iq", bytes(arr[8:32]))
            raise RuntimeError("packed blob does not match this model: file (layout, arch, dtype, attn_min_level, bytes) = "
                               f"{got}, model = {struct.unpack('<4iq', bytes(want[8:32]))}")
        if arr.size - 64 != self.weight_blob_bytes():
            raise RuntimeError("packed blob is truncated")
        blob = torch.from_numpy(arr[64:].copy()).to(self.device)
        self.finalize(blob=blob, fill=False)
        return self

    def weight_blob_bytes(self) -> int:
        return int(self._lib.ddpmir_weight_blob_bytes(self._ctx))

    def profile(self, enable, detail=False):
        """HIP-event spans: forwards + full-resolution attention launches; detail=True (or 2) brackets all 13 attention
        launches, detail=3 also every GEMM launch (classes "gemm" / "gemm_fp8")."""
        level = 0 if not enable else (int(detail) if int(detail) > 1 else (2 if detail else 1))
        _lib.check(self._lib, self._ctx, self._lib.ddpmir_profile_enable(self._ctx, level), "profile_enable")

    def set_option(self, name: str, value: float):
        """ddpmir_set_option: e.g. ("attention_no_max_bound", -1.0) forces the general attention kernel."""
        _lib.check(self._lib, self._ctx, self._lib.ddpmir_set_option(self._ctx, name.encode(), float(value)), f"set_option({name})")

    def version(self) -> str:
        return self._lib.ddpmir_version().decode()

    def profile_read(self, cls: str):
        """(device ms, launches, algorithmic FLOPs) accumulated for a kernel class since profile(True)."""
        ms, n, fl = C.c_double(0), C.c_int64(0), C.c_double(0)
        rc = self._lib.ddpmir_profile_read(self._ctx, cls.encode(), C.byref(ms), C.byref(n), C.byref(fl))
        _lib.check(self._lib, self._ctx, rc, "profile_read")
        return ms.value, n.value, fl.value

    def param_count(self) -> int:
        return int(self._lib.ddpmir_param_count(self._ctx))

    def __call__(self, x, t, compression_level=None):
        torch = _torch()
        if not self._loaded:
            raise RuntimeError("load_state_dict() first")
        x = x.to(self.device, torch.float32).contiguous()
        t = t.to(self.device, torch.float32).contiguous()
        cl = t if compression_level is None else compression_level.to(self.device, torch.float32).contiguous()
        b, c, h, w = x.shape
        if c != 3:
            raise RuntimeError("expected [B,3,H,W]")
        out = torch.empty_like(x)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        rc = self._lib.ddpmir_predict(self._ctx, x.data_ptr(), t.data_ptr(), cl.data_ptr(), out.data_ptr(), b, h, w,
                                      C.c_void_p(stream))
        _lib.check(self._lib, self._ctx, rc, "predict")
        return out

    forward = __call__
    predict_noise = __call__      # north_star wording; the DDRM models predict x0_hat, not eps

    def __del__(self):
        try:
            if getattr(self, "_ctx", None):
                self._lib.ddpmir_destroy(self._ctx)
                self._ctx = None
        except Exception:
            pass


class WebPDiffusionModel(_RestorationModel):
    """webp_inference.py:330-399"""
    ARCH = spec.ARCH_WEBP


class JPEGDiffusionModel(_RestorationModel):
    """svd.ipynb c0:297-366"""
    ARCH = spec.ARCH_JPEG


class AVIFDiffusionModel(_RestorationModel):
    """avif_inference.py:316-385"""
    ARCH = spec.ARCH_AVIF


MODEL_BY_ARCH = {spec.ARCH_WEBP: WebPDiffusionModel, spec.ARCH_JPEG: JPEGDiffusionModel,
                 spec.ARCH_AVIF: AVIFDiffusionModel}


def convert_checkpoint(model: _RestorationModel, ckpt_path: str, out_path: str):
    """best_ddrm_*_model.pth -> packed blob file: validates every state-dict name/shape, packs on the
    device into a caller-owned blob and writes it; `model.load_packed(out_path)` restores it on any rank."""
    torch = _torch()
    ckpt = torch.load(ckpt_path, map_location="cpu")
    sd = ckpt["model_state_dict"] if isinstance(ckpt, dict) and "model_state_dict" in ckpt else ckpt
    want = spec.state_spec(model.ARCH)
    missing = [k for k in want if k not in sd]
    unexpected = [k for k in sd if k not in want]
    if missing or unexpected:
        raise RuntimeError(f"state_dict mismatch: missing {missing[:5]} unexpected {unexpected[:5]}")
    blob = torch.empty(model.weight_blob_bytes(), dtype=torch.uint8, device=model.device)
    for k in want:
        model._load_one(k, sd[k])
    model.finalize(blob=blob, fill=True)
    model.save_packed(out_path)
    return model


def load_checkpoint(model: _RestorationModel, path: str):
    """webp_inference.py:621-627: accepts {'model_state_dict': ...} or a bare state_dict."""
    torch = _torch()
    ckpt = torch.load(path, map_location="cpu")
    sd = ckpt["model_state_dict"] if isinstance(ckpt, dict) and "model_state_dict" in ckpt else ckpt
    model.load_state_dict(OrderedDict(sd))
    return model


class UNetDenoiser:
    """eps-predict denoiser of the early experiments (experiments/code/conv.ipynb c0:55-101): `net(x, t)` returns
    the predicted noise; `ddpm_inference` / p_sample_loop is the deterministic posterior-mean loop (c1:117-125)."""

    def __init__(self, device=None):
        import torch
        self._lib = _lib.load()
        self.device = torch.device(device if device is not None else ("cuda", torch.cuda.current_device()))
        self._ctx = self._lib.ddpmir_create_eps_unet(self.device.index or 0)
        if not self._ctx:
            raise RuntimeError("ddpmir_create_eps_unet failed: " + self._lib.ddpmir_last_error(None).decode())
        self._loaded = False

    def eval(self):
        return self

    def to(self, device):
        return self

    def load_state_dict(self, state, strict=True):
        want = spec.eps_unet_spec()
        if strict and set(want) != set(state):
            raise RuntimeError("state_dict mismatch for UNetDenoiser")
        for name in want:
            v = state[name]
            arr = v.detach().cpu().float().numpy() if hasattr(v, "detach") else np.asarray(v, dtype=np.float32)
            arr = np.ascontiguousarray(arr, dtype=np.float32)
            if tuple(arr.shape) != tuple(want[name]):
                raise RuntimeError(f"shape mismatch for {name}")
            shape = (C.c_int64 * arr.ndim)(*arr.shape)
            rc = self._lib.ddpmir_load_tensor(self._ctx, name.encode(), arr.ctypes.data_as(C.c_void_p), 0, shape, arr.ndim)
            _lib.check(self._lib, self._ctx, rc, f"load_tensor({name})")
        _lib.check(self._lib, self._ctx, self._lib.ddpmir_finalize_weights(self._ctx, None, 1), "finalize_weights")
        self._loaded = True
        return self

    def __call__(self, x, t):
        """x [B,3,H,W] in [0,1]; t: python number or 1-element tensor (the reference passes [[t]])."""
        import torch
        x = x.to(self.device, torch.float32).contiguous()
        out = torch.empty_like(x)
        tv = float(t.reshape(-1)[0].item()) if hasattr(t, "reshape") else float(t)
        b, c, h, w = x.shape
        torch.cuda.current_stream(self.device).synchronize()
        rc = self._lib.ddpmir_predict_noise(self._ctx, x.data_ptr(), tv, out.data_ptr(), b, h, w)
        _lib.check(self._lib, self._ctx, rc, "predict_noise")
        return out

    predict_noise = __call__

    def p_sample_loop(self, x, T, betas):
        import torch
        x = x.to(self.device, torch.float32).contiguous()
        out = torch.empty_like(x)
        bt = np.ascontiguousarray(betas.detach().cpu().numpy() if hasattr(betas, "detach") else betas, dtype=np.float32)
        b, c, h, w = x.shape
        torch.cuda.current_stream(self.device).synchronize()
        rc = self._lib.ddpmir_p_sample_loop(self._ctx, x.data_ptr(), out.data_ptr(), b, h, w, int(T),
                                            bt.ctypes.data_as(C.c_void_p), int(bt.size))
        _lib.check(self._lib, self._ctx, rc, "p_sample_loop")
        return out

    def __del__(self):
        try:
            if getattr(self, "_ctx", None):
                self._lib.ddpmir_destroy(self._ctx)
                self._ctx = None
        except Exception:
            pass


def ddpm_inference(model: UNetDenoiser, x0, T=80, betas=None):
    """ddpm_inference(model, x0, T) of conv.ipynb c1:117-125 (beta = linspace(1e-4, 0.02, 100)); the caller supplies
    the already-degraded image (the reference jpeg-compresses x0 at q20 first)."""
    if betas is None:
        betas = np.linspace(1e-4, 0.02, 100, dtype=np.float64).astype(np.float32)
    return model.p_sample_loop(x0, T, betas)
