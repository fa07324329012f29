"""numpy wrappers of the single-operator entry points (include/ddpmir_ops.h); NHWC float32 in and out.
Used by the per-operator parity tests (each operator = one row of SURVEY.md section 8a)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

_P = C.c_void_p
_sigs = {
    "ddpmir_ops_last_error": (C.c_char_p, []),
    "ddpmir_op_conv": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, C.c_int, _P, _P, C.c_int,
                                 C.c_int, C.c_int, C.c_int, _P, C.c_int, _P, _P, C.c_int, _P, C.c_int, _P]),
    "ddpmir_op_set_bf16": (None, [C.c_int]),
    "ddpmir_op_set_gate_select": (None, [C.c_int]),
    "ddpmir_op_clock_sample_begin": (C.c_int, [C.c_double]),
    "ddpmir_op_clock_sample_end": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "ddpmir_op_set_tile256": (None, [C.c_int]),
    "ddpmir_op_set_epilogue_direct": (None, [C.c_int]),
    "ddpmir_op_gemm256_launches": (C.c_longlong, []),
    "ddpmir_op_conv_time": (C.c_int, [C.c_int] * 11 + [C.POINTER(C.c_float)]),
    "ddpmir_op_gn_scale_shift": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P]),
    "ddpmir_op_attention": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "ddpmir_op_attention_bf16": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "ddpmir_op_attention_bf16_ex": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _P, _P]),
    "ddpmir_op_conv_gnstats_bf16": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, _P, _P]),
    "ddpmir_op_conv_dual_bf16": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P, C.c_int, _P, _P, C.c_int, _P]),
    "ddpmir_op_conv_tile": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P]),
    "ddpmir_op_conv_dual_fp8": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, _P, _P, C.c_int, _P]),
    "ddpmir_op_qkv_attention_bf16": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, _P]),
    "ddpmir_op_attention_time": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "ddpmir_op_block_transform": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_float,
                                            C.c_float, _P]),
    "ddpmir_op_freq_gates_bf16": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, C.c_int, _P,
                                            _P, _P, _P]),
    "ddpmir_op_maxpool2": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "ddpmir_op_upsample2": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "ddpmir_op_stream16": (C.c_int, [C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, C.c_int]),
    "ddpmir_op_bilinear": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "ddpmir_op_adaptive_avgpool": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "ddpmir_op_time_embed": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, _P]),
    "ddpmir_op_quantize_u8": (C.c_int, [_P, C.c_int64, _P]),
    "ddpmir_op_ddrm_update": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int,
                                        C.c_int, _P]),
    "ddpmir_op_randn": (C.c_int, [C.c_int64, C.c_uint64, C.c_uint64, _P]),
}
OPS_SYMBOLS = _sigs
_bound = False


def _lib_ops():
    global _bound
    lib = _lib.load()
    if not _bound:
        for n, (r, a) in _sigs.items():
            f = getattr(lib, n)
            f.restype, f.argtypes = r, a
        _bound = True
    return lib


def _f(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return None if a is None else a.ctypes.data_as(_P)


def _chk(lib, rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed ({rc}): {lib.ddpmir_ops_last_error().decode()}")


ACT = {None: 0, "none": 0, "gelu": 1, "silu": 2, "lrelu": 3, "relu": 4, "sigmoid": 5, "tanh": 6}
MASK = {None: 0, "none": 0, "low": 1, "high": 2}


def conv(x, w, bias=None, scale=None, shift=None, pro_act=None, pro_mask=None, bs=4, low=3, tb=None, epi_act=None,
         mul=None, rowscale=None, epi_mask=None, res=None, nchw_out=False, bf16=False, gate_select=0):
    lib = _lib_ops()
    lib.ddpmir_op_set_gate_select(int(gate_select))
    lib.ddpmir_op_set_bf16({"dma": 2, "fp8": 3}.get(bf16, int(bf16) if not isinstance(bf16, str) else 0))
    x, w = _f(x), _f(w)
    if w.ndim == 2:
        w = w[:, :, None, None]
    b, h, wd, cin = x.shape
    cout, _, k, _ = w.shape
    bias, scale, shift, tb, mul, rowscale, res = map(_f, (bias, scale, shift, tb, mul, rowscale, res))
    co = gate_select if gate_select else cout
    out = np.empty((b, co, h, wd) if nchw_out else (b, h, wd, co), dtype=np.float32)
    rc = lib.ddpmir_op_conv(_p(x), b, h, wd, cin, _p(w), _p(bias), cout, k, _p(scale), _p(shift), ACT[pro_act],
                            MASK[pro_mask], bs, low, _p(tb), ACT[epi_act], _p(mul), _p(rowscale), MASK[epi_mask],
                            _p(res), int(nchw_out), _p(out))
    _chk(lib, rc, "op_conv")
    return out


def clock_sample_begin(ms: float):
    """start the one-wave shader-clock sampler (own stream) for `ms` of wall time"""
    lib = _lib_ops()
    _chk(lib, lib.ddpmir_op_clock_sample_begin(float(ms)), "clock_sample_begin")


def clock_sample_end():
    """-> (effective shader clock in GHz, sampled wall time in ms)"""
    lib = _lib_ops()
    g, m = C.c_double(0), C.c_double(0)
    _chk(lib, lib.ddpmir_op_clock_sample_end(C.byref(g), C.byref(m)), "clock_sample_end")
    return g.value, m.value


def set_tile256(mode: int):
    """IGemm::tile256 for the GEMM ops of this module: -1 auto (default), 0 never, 1 wherever gemm256_supported()."""
    _lib_ops().ddpmir_op_set_tile256(int(mode))


def set_epilogue_direct(on: int):
    """IGemm::epi_direct_opt for the GEMM ops of this module (1 default; 0 = LDS-transposed epilogue everywhere)."""
    _lib_ops().ddpmir_op_set_epilogue_direct(int(on))


def gemm256_launches() -> int:
    return int(_lib_ops().ddpmir_op_gemm256_launches())


def conv_time(b, h, w, cin, cout, k, bf16=True, gelu_norm=False, residual=False, dbg=0, iters=5):
    lib = _lib_ops()
    ms = C.c_float(0)
    _chk(lib, lib.ddpmir_op_conv_time(b, h, w, cin, cout, k, {"dma": 2, "fp8": 3}.get(bf16, int(bf16) if not isinstance(bf16, str) else 0), int(gelu_norm), int(residual), dbg, iters,
                                      C.byref(ms)), "op_conv_time")
    return ms.value


def gn_scale_shift(x, groups, gamma, beta):
    lib = _lib_ops()
    x = _f(x)
    b, c = x.shape[0], x.shape[-1]
    hw = int(np.prod(x.shape[1:-1]))
    sc = np.empty((b, c), np.float32)
    sh = np.empty((b, c), np.float32)
    g, be = _f(gamma), _f(beta)
    _chk(lib, lib.ddpmir_op_gn_scale_shift(_p(x), b, hw, c, groups, _p(g), _p(be), _p(sc), _p(sh)), "op_gn")
    return sc, sh


def attention(qkv, heads, bf16=False):
    lib = _lib_ops()
    qkv = _f(qkv)
    b, n, c3 = qkv.shape
    out = np.empty((b, n, c3 // 3), np.float32)
    fn = lib.ddpmir_op_attention_bf16 if bf16 else lib.ddpmir_op_attention
    _chk(lib, fn(_p(qkv), b, n, c3 // 3, heads, _p(out)), "op_attention")
    return out


def attention_bf16_routed(qkv, heads, nm_bound=80.0):
    """bf16 attention with an explicit routing bound; returns (out, general workgroups with work, general workgroups launched)."""
    lib = _lib_ops()
    qkv = _f(qkv)
    b, n, c3 = qkv.shape
    out = np.empty((b, n, c3 // 3), np.float32)
    stats = np.zeros(2, np.int64)
    _chk(lib, lib.ddpmir_op_attention_bf16_ex(_p(qkv), b, n, c3 // 3, heads, float(nm_bound), _p(out), stats.ctypes.data_as(_P)), "op_attention_ex")
    return out, int(stats[0]), int(stats[1])


def conv_gnstats_bf16(x, w, bias, groups):
    """conv (NHWC in/out) + per-(sample, group) sum / sum-of-squares of the output, from the same launch."""
    lib = _lib_ops()
    x, w, bias = _f(x), _f(w), _f(bias)
    b, h, wd, cin = x.shape
    cout, _, k, _ = w.shape
    out = np.empty((b, h, wd, cout), np.float32)
    sums = np.empty((b, groups, 2), np.float64)
    _chk(lib, lib.ddpmir_op_conv_gnstats_bf16(_p(x), b, h, wd, cin, _p(w), _p(bias), cout, k, groups, _p(out),
                                              sums.ctypes.data_as(C.c_void_p)), "op_conv_gnstats")
    return out, sums


def conv_dual_bf16(x1, w1, x2, w2, bias):
    """conv(x1, w1 [Cout,C1,k,k]) + conv1x1(x2, w2 [Cout,C2]) + bias in one launch; NHWC in/out."""
    lib = _lib_ops()
    x1, w1, x2, w2, bias = _f(x1), _f(w1), _f(x2), _f(w2), _f(bias)
    b, h, wd, c1 = x1.shape
    cout, _, k, _ = w1.shape
    out = np.empty((b, h, wd, cout), np.float32)
    _chk(lib, lib.ddpmir_op_conv_dual_bf16(_p(x1), b, h, wd, c1, _p(w1), k, _p(x2), x2.shape[-1], _p(w2), _p(bias), cout,
                                           _p(out)), "op_conv_dual")
    return out


def conv_tile(x, w, bias, scale=None, shift=None, act=None, tb=None, groups=0, in16=False, want32=True, want16=False, want_raw=False):
    """conv3x3(act(x * scale + shift)) + bias (+ tb) with the norm-apply pass fused into the conv (conv_tile.hip); NHWC in/out.
    Returns a dict with the requested outputs: out, out16, sums, raw16."""
    lib = _lib_ops()
    x, w, bias, scale, shift, tb = _f(x), _f(w), _f(bias), _f(scale), _f(shift), _f(tb)
    b, h, wd, cin = x.shape
    cout = w.shape[0]
    out = np.empty((b, h, wd, cout), np.float32) if want32 else None
    out16 = np.empty((b, h, wd, cout), np.float32) if want16 else None
    sums = np.empty((b, groups, 2), np.float64) if groups else None
    raw = np.empty((b, h, wd, cin), np.float32) if want_raw else None
    _chk(lib, lib.ddpmir_op_conv_tile(_p(x), b, h, wd, cin, int(in16), _p(scale), _p(shift), ACT[act], _p(w), _p(bias), _p(tb), cout, int(groups),
                                      int(want16), _p(out), _p(out16), sums.ctypes.data_as(C.c_void_p) if groups else None, _p(raw)), "op_conv_tile")
    return {"out": out, "out16": out16, "sums": sums, "raw16": raw}


def conv_dual_fp8(x1, w1, x2, w2, bias):
    """conv3x3(x1, w1) with e4m3 operands + conv1x1(x2, w2) in bf16 + bias in one launch (fp8 mode's conv_out); NHWC in/out."""
    lib = _lib_ops()
    x1, w1, x2, w2, bias = _f(x1), _f(w1), _f(x2), _f(w2), _f(bias)
    b, h, wd, c1 = x1.shape
    cout = w1.shape[0]
    out = np.empty((b, h, wd, cout), np.float32)
    _chk(lib, lib.ddpmir_op_conv_dual_fp8(_p(x1), b, h, wd, c1, _p(w1), _p(x2), x2.shape[-1], _p(w2), _p(bias), cout, _p(out)), "op_conv_dual_fp8")
    return out


def qkv_attention_bf16(x, w_in, b_in, heads, fused=True):
    """x [B,N,C] -> softmax(QK^T/sqrt(d))V per head, [B,N,C]; in_proj fused with the operand packing or not."""
    lib = _lib_ops()
    x, w_in, b_in = _f(x), _f(w_in), _f(b_in)
    b, n, c = x.shape
    out = np.empty((b, n, c), np.float32)
    _chk(lib, lib.ddpmir_op_qkv_attention_bf16(_p(x), b, n, c, heads, _p(w_in), _p(b_in), int(fused), _p(out)), "op_qkv_attention")
    return out


def attention_time(b, n, c, heads, mode, iters=3):
    lib = _lib_ops()
    ms = C.c_float(0)
    _chk(lib, lib.ddpmir_op_attention_time(b, n, c, heads, mode, iters, C.byref(ms)), "op_attention_time")
    return ms.value


def block_transform(x, mats, mix_in=0.0, mix_t=1.0):
    lib = _lib_ops()
    x, mats = _f(x), _f(mats)
    b, h, w, c = x.shape
    out = np.empty_like(x)
    _chk(lib, lib.ddpmir_op_block_transform(_p(x), b, h, w, c, mats.shape[-1], _p(mats), int(mats.ndim == 3), mix_in,
                                            mix_t, _p(out)), "op_block_transform")
    return out


def freq_gates_bf16(x, dct, w1, b1, w2, b2, boost, low, fused=True, att=None, wo=None, bo=None):
    """S = x + gate(DCT(x)) * DCT(x) of the DCT families' frequency block in bf16 mode (NHWC; bf16 values as float32);
    with att / wo / bo: x := wo att + bo + x first (the block attention's output projection and residual)."""
    lib = _lib_ops()
    x, dct, w1, b1, w2, b2, boost = _f(x), _f(dct), _f(w1), _f(b1), _f(w2), _f(b2), _f(boost)
    att, wo, bo = _f(att), _f(wo), _f(bo)
    b, h, w, c = x.shape
    out = np.empty_like(x)
    _chk(lib, lib.ddpmir_op_freq_gates_bf16(_p(x), b, h, w, c, dct.shape[-1], int(low), _p(dct), _p(w1), _p(b1), _p(w2), _p(b2),
                                            _p(boost), int(fused), _p(out), _p(att), _p(wo), _p(bo)), "op_freq_gates_bf16")
    return out


def maxpool2(x):
    lib = _lib_ops()
    x = _f(x)
    b, h, w, c = x.shape
    out = np.empty((b, h // 2, w // 2, c), np.float32)
    _chk(lib, lib.ddpmir_op_maxpool2(_p(x), b, h, w, c, _p(out)), "op_maxpool2")
    return out


def upsample2(x):
    lib = _lib_ops()
    x = _f(x)
    b, h, w, c = x.shape
    out = np.empty((b, 2 * h, 2 * w, c), np.float32)
    _chk(lib, lib.ddpmir_op_upsample2(_p(x), b, h, w, c, _p(out)), "op_upsample2")
    return out


def stream16(mode, x, groups=0, gamma=None, beta=None, f32=False):
    """inter-block stream pass on NHWC x (f32=False: the bf16 stream, x rounded to bf16 inside): mode 0 statistics, 1 max-pool 2x2, 2 bilinear x2.
    Returns (out or None, scale, shift) -- scale / shift [B, C] are the GroupNorm affine of the produced tensor (groups > 0)."""
    lib = _lib_ops()
    x = _f(x)
    b, h, w, c = x.shape
    ho, wo = (h // 2, w // 2) if mode == 1 else (2 * h, 2 * w) if mode == 2 else (h, w)
    out = np.empty((b, ho, wo, c), np.float32)
    sc, sh = np.empty((b, c), np.float32), np.empty((b, c), np.float32)
    g = _f(gamma) if gamma is not None else np.ones(c, np.float32)
    be = _f(beta) if beta is not None else np.zeros(c, np.float32)
    _chk(lib, lib.ddpmir_op_stream16(int(mode), _p(x), b, h, w, c, c, int(groups), _p(g), _p(be), _p(out), _p(sc), _p(sh), int(bool(f32))), "op_stream16")
    return (out if mode else None), sc, sh


def bilinear(x, ho, wo):
    lib = _lib_ops()
    x = _f(x)
    b, h, w, c = x.shape
    out = np.empty((b, ho, wo, c), np.float32)
    _chk(lib, lib.ddpmir_op_bilinear(_p(x), b, h, w, c, ho, wo, _p(out)), "op_bilinear")
    return out


def adaptive_avgpool(x, s):
    lib = _lib_ops()
    x = _f(x)
    b, h, w, c = x.shape
    out = np.empty((b, s, s, c), np.float32)
    _chk(lib, lib.ddpmir_op_adaptive_avgpool(_p(x), b, h, w, c, s, _p(out)), "op_adaptive_avgpool")
    return out


def time_embed(t, w0, b0, w2, b2):
    lib = _lib_ops()
    t, w0, b0, w2, b2 = map(_f, (t, w0, b0, w2, b2))
    out = np.empty((t.shape[0], 256), np.float32)
    _chk(lib, lib.ddpmir_op_time_embed(_p(t), t.shape[0], _p(w0), _p(b0), _p(w2), _p(b2), _p(out)), "op_time_embed")
    return out


def quantize_u8(x):
    lib = _lib_ops()
    x = _f(x)
    out = np.empty(x.shape, np.uint8)
    _chk(lib, lib.ddpmir_op_quantize_u8(_p(x), x.size, out.ctypes.data_as(_P)), "op_quantize_u8")
    return out


def ddrm_update(x_theta, dec_u8, y, noise, eta, eta_b, noise_coeff, step_i, steps):
    lib = _lib_ops()
    x_theta, y, noise = _f(x_theta), _f(y), _f(noise)
    dec_u8 = np.ascontiguousarray(dec_u8, dtype=np.uint8)
    b = x_theta.shape[0]
    out = np.empty_like(x_theta)
    _chk(lib, lib.ddpmir_op_ddrm_update(_p(x_theta), dec_u8.ctypes.data_as(_P), _p(y), _p(noise), x_theta.size // b, b,
                                        eta, eta_b, noise_coeff, step_i, steps, _p(out)), "op_ddrm_update")
    return out


def randn(n, seed, offset=0):
    lib = _lib_ops()
    out = np.empty((n,), np.float32)
    _chk(lib, lib.ddpmir_op_randn(n, seed, offset, _p(out)), "op_randn")
    return out
