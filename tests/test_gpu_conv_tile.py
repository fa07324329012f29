"""conv_tile.hip: GroupNorm-apply (+ GELU) fused into the full-resolution 3x3 convs (webp_inference.py:306-313), through the ops C
ABI, against the numpy oracle evaluated on the kernel's own operand (act(x * scale + shift) rounded to bf16, weights rounded to
bf16, exact products, fp32 accumulation) and against the two-launch path it replaces (gn_apply16 pass + DMA GEMM)."""
import numpy as np
import pytest

from oracle import ddpm_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from ddpm_image_restoration_amd import ops as _ops
    return _ops


def rnd(*shape, seed=0, scale=1.0):
    return (np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32)


def nhwc(x):
    return np.ascontiguousarray(x.transpose(0, 2, 3, 1))


def nchw(x):
    return np.ascontiguousarray(x.transpose(0, 3, 1, 2))


def bf(z):
    u = np.ascontiguousarray(z, dtype=np.float32).view(np.uint32)
    return ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32).view(np.float32)


def close(a, b, tol):
    scale = max(1.0, float(np.abs(b).max()))
    err = float(np.abs(a - b).max())
    assert err <= tol * scale, f"max err {err} (scale {scale})"


def gelu16(v):
    """gelu_bf16_grade of common.h: erf by Abramowitz & Stegun 7.1.26 (what gn_apply16_k and conv_tile_kernel evaluate)"""
    v = v.astype(np.float32)
    x = np.abs(v) * np.float32(0.70710678118654752440)
    t = np.float32(1.0) / (np.float32(0.3275911) * x + np.float32(1.0))
    poly = t * (t * (t * (t * (t * np.float32(1.061405429) + np.float32(-1.453152027)) + np.float32(1.421413741)) + np.float32(-0.284496736)) + np.float32(0.254829592))
    e = np.float32(1.0) - poly * np.exp(-x * x).astype(np.float32)
    return (np.float32(0.5) * v * (np.float32(1.0) + np.copysign(e, v))).astype(np.float32)


SHAPES = [(2, 8, 32, 64), (1, 16, 64, 64), (3, 24, 32, 64), (2, 32, 96, 64), (5, 64, 64, 64), (2, 256, 256, 64)]


@pytest.mark.parametrize("B,H,W,Cin", SHAPES)
@pytest.mark.parametrize("in16", [False, True])
def test_conv_tile_norm_gelu_conv(ops, B, H, W, Cin, in16):
    """norm2 -> GELU -> conv2: per-sample scale / shift, the bf16-grade GELU, zero padding of the ACTIVATED tensor at the image
    borders (one-tile images: every tile row touches a border), several tiles per image and per workgroup (the persistent walk:
    (2, 256, 256) is 512 tiles on 256 workgroups, (5, 64, 64) 80 tiles: fewer than CUs), fp32 and bf16 input; fp32 + bf16 outputs
    from one launch.  The oracle's operand is the numpy restatement of the bf16-grade GELU: an exp that differs in the last bit
    can flip a bf16 rounding of the operand, hence the 1e-3 bar here -- the tight bar is the comparison with the two-launch path
    below, which evaluates the same device functions."""
    Cout = 64
    x = rnd(B, Cin, H, W, seed=1, scale=1.5) + 0.3
    sc, sh = rnd(B, Cin, seed=2) * 0.3 + 1.0, rnd(B, Cin, seed=3) * 0.2
    w = rnd(Cout, Cin, 3, 3, seed=4, scale=1.0 / np.sqrt(Cin * 9))
    b = rnd(Cout, seed=5)
    xi = bf(x) if in16 else x
    a = bf(gelu16(xi * sc[:, :, None, None] + sh[:, :, None, None]))
    ref = O.conv2d(a, bf(w), b, 1)
    r = ops.conv_tile(nhwc(x), w, b, scale=sc, shift=sh, act="gelu", in16=in16, want32=True, want16=True)
    close(nchw(r["out"]), ref, 1e-3)
    assert np.array_equal(r["out16"], bf(r["out"]))                      # the bf16 copy is the rounded fp32 result
    again = ops.conv_tile(nhwc(x), w, b, scale=sc, shift=sh, act="gelu", in16=in16, want32=True)
    assert np.array_equal(again["out"], r["out"])                        # deterministic; the output set does not change the values


def test_conv_tile_norm1_conv1_with_time_bias_statistics_and_raw_copy(ops):
    """norm1 -> conv1 of a 64-channel block: no activation, time bias, the GroupNorm partial sums of the OUTPUT from the epilogue
    (statistics of the ROUNDED values when bf16 is the only copy), and the bf16 copy of the raw input that a 1x1 shortcut reads."""
    B, H, W, Cin, Cout, G = 2, 32, 64, 64, 64, 8
    x = rnd(B, Cin, H, W, seed=11, scale=2.0) + 0.5
    sc, sh = rnd(B, Cin, seed=12) * 0.3 + 1.0, rnd(B, Cin, seed=13) * 0.2
    w = rnd(Cout, Cin, 3, 3, seed=14, scale=1.0 / np.sqrt(Cin * 9))
    b, tb = rnd(Cout, seed=15), rnd(B, Cout, seed=16)
    a = bf(x * sc[:, :, None, None] + sh[:, :, None, None])
    ref = O.conv2d(a, bf(w), b, 1) + tb[:, :, None, None]
    r = ops.conv_tile(nhwc(x), w, b, scale=sc, shift=sh, act=None, tb=tb, groups=G, want32=False, want16=True, want_raw=True)
    close(nchw(r["out16"]), bf(ref), 1.2e-2)                              # bf16 output: one rounding step of the result's scale
    assert np.array_equal(r["raw16"], bf(nhwc(x)))
    o = r["out16"].astype(np.float64).reshape(B, H * W, G, Cout // G)
    want = np.stack([o.sum(axis=(1, 3)), (o * o).sum(axis=(1, 3))], axis=-1)
    assert np.allclose(r["sums"], want, rtol=2e-5, atol=1e-2), np.abs(r["sums"] - want).max()
    r32 = ops.conv_tile(nhwc(x), w, b, scale=sc, shift=sh, act=None, tb=tb, groups=G, want32=True)
    close(nchw(r32["out"]), ref, 1e-3)            # (numpy forms x * sc + sh with two roundings, the kernel with one fmaf: a few operand values round the other way)
    o = r32["out"].astype(np.float64).reshape(B, H * W, G, Cout // G)
    want = np.stack([o.sum(axis=(1, 3)), (o * o).sum(axis=(1, 3))], axis=-1)
    assert np.allclose(r32["sums"], want, rtol=2e-5, atol=1e-2)


@pytest.mark.parametrize("B,H,W", [(2, 64, 64), (3, 128, 96), (1, 8, 32)])
def test_conv_tile_equals_the_operand_pass_plus_dma_gemm(ops, B, H, W):
    """the path it replaces: gn_apply16 (scale / shift + GELU -> bf16 operand) + the strip kernel.  Same operand values, same
    products; only the accumulation order differs."""
    C = 64
    x = rnd(B, C, H, W, seed=21, scale=2.0) + 0.5
    gamma, beta = rnd(C, seed=22) * 0.2 + 1, rnd(C, seed=23) * 0.1
    w, b = rnd(C, C, 3, 3, seed=24, scale=0.05), rnd(C, seed=25)
    sc, sh = ops.gn_scale_shift(nhwc(x), 8, gamma, beta)
    two = ops.conv(nhwc(x), w, b, scale=sc, shift=sh, pro_act="gelu", bf16="dma")
    one = ops.conv_tile(nhwc(x), w, b, scale=sc, shift=sh, act="gelu")["out"]
    close(one, two, 2e-5)


def test_conv_tile_rejects_what_it_cannot_do(ops):
    x, w, b = rnd(1, 64, 12, 32, seed=1), rnd(64, 64, 3, 3, seed=2), rnd(64, seed=3)
    with pytest.raises(RuntimeError, match="unsupported"):
        ops.conv_tile(nhwc(x), w, b)                                      # H % 8 != 0
    with pytest.raises(RuntimeError, match="unsupported"):
        ops.conv_tile(nhwc(rnd(1, 64, 8, 32, seed=1)), rnd(128, 64, 3, 3, seed=2), rnd(128, seed=3))   # Cout != 64
    with pytest.raises(RuntimeError, match="unsupported"):
        ops.conv_tile(nhwc(rnd(1, 128, 8, 32, seed=1)), rnd(64, 128, 3, 3, seed=2), rnd(64, seed=3))   # Cin != 64 (144 KiB of weights)
