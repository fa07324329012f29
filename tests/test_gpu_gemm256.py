"""The 256 x 256 eight-wave implicit-GEMM tile (csrc/gemm256.hip) for the 3x3 convs with Cout >= 256
(webp_inference.py:282, 292, 229 at levels 2-5), through the ops C ABI, against the numpy oracle and against the 128-row
kernel it replaces.  Every case proves with the launch counter that the tile really ran (no silent fall-back).

Shapes: one tile, ragged M (rows past M read zeros through the buffer descriptor's range check), tiles that straddle several
images (every pixel's nine-tap validity mask), odd and even K-tile counts, several channel chunks, Cout 256 / 512 / 1024,
split-K (few tiles, linear epilogue) and the unsplit path (fused GroupNorm statistics), the second K segment, the fused
epilogue, e4m3 operands on exactly representable data."""
import numpy as np
import pytest

from oracle import ddpm_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ops():
    from ddpm_image_restoration_amd import ops as _ops
    yield _ops
    _ops.set_tile256(-1)


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


def both(ops, fn):
    """fn() with the tile forced and with the tile off; asserts the forced run took the tile and the other did not."""
    n0 = ops.gemm256_launches()
    ops.set_tile256(1)
    a = fn()
    n1 = ops.gemm256_launches()
    ops.set_tile256(0)
    b = fn()
    n2 = ops.gemm256_launches()
    ops.set_tile256(-1)
    assert n1 > n0, "the 256 x 256 tile did not take this launch"
    assert n2 == n1, "tile256 = 0 must keep the launch on the 128-row kernel"
    return a, b


SHAPES = [
    (1, 16, 16, 64, 256),      # one tile, 9 K tiles (odd), split in two
    (2, 12, 20, 128, 512),     # ragged M (480), two column tiles
    (3, 9, 13, 192, 256),      # ragged M (351), 27 K tiles, tiles straddling images with odd row length
    (5, 8, 8, 64, 256),        # a tile covers four images (borders inside the tile)
    (1, 32, 32, 256, 1024),    # four column tiles, 36 K tiles
    (2, 1, 160, 64, 256),      # one-row images: every dy = -1 / +1 tap is halo
    (1, 160, 2, 128, 256),     # two-column images: every pixel sits on a left or right border
    (1, 48, 48, 512, 512),     # 2304 pixels = 9 row tiles x 2, 72 K tiles
]


@pytest.mark.parametrize("B,H,W,Cin,Cout", SHAPES)
def test_conv_bf16_tile256(ops, B, H, W, Cin, Cout):
    x = rnd(B, Cin, H, W, seed=1)
    w = rnd(Cout, Cin, 3, 3, seed=2, scale=1.0 / np.sqrt(Cin * 9))
    b = rnd(Cout, seed=3)
    got, old = both(ops, lambda: ops.conv(nhwc(x), w, b, bf16="dma"))
    ref = O.conv2d(bf(x), bf(w), b, 1)
    close(nchw(got), ref, 3e-5)
    close(got, old, 2e-5)           # the 128-row kernel: same products, another summation order at most


def test_conv_bf16_tile256_auto_mode_takes_chip_filling_shapes_only(ops):
    """tile256 = -1 (measured policy, tools/gemm256_ab.py): 256 whole tiles take the launch unsplit (B=16, 64 x 64, 64 -> 256);
    64 tiles with a long K loop (B=8, 32 x 32, 512 -> 512: 72 K tiles, four splits of 18) too; 128 tiles with a short K loop
    (B=8, 64 x 64, 64 -> 256: 9 K tiles) and a single tile stay on the 128-row kernel."""
    ops.set_tile256(-1)
    w = rnd(256, 64, 3, 3, seed=5, scale=1.0 / 24)
    b = rnd(256, seed=6)
    x = rnd(16, 64, 64, 64, seed=4)
    n0 = ops.gemm256_launches()
    got = ops.conv(nhwc(x), w, b, bf16="dma")
    assert ops.gemm256_launches() == n0 + 1
    close(nchw(got)[:2], O.conv2d(bf(x[:2]), bf(w), b, 1), 3e-5)
    close(nchw(got)[15:], O.conv2d(bf(x[15:]), bf(w), b, 1), 3e-5)
    ops.conv(nhwc(x[:8]), w, b, bf16="dma")
    ops.conv(nhwc(x[:1, :, :16, :16]), w, b, bf16="dma")
    assert ops.gemm256_launches() == n0 + 1
    x5, w5, b5 = rnd(8, 512, 32, 32, seed=14), rnd(512, 512, 3, 3, seed=15, scale=1.0 / 68), rnd(512, seed=16)
    got5 = ops.conv(nhwc(x5), w5, b5, bf16="dma")
    assert ops.gemm256_launches() == n0 + 2
    close(nchw(got5)[:1], O.conv2d(bf(x5[:1]), bf(w5), b5, 1), 3e-5)


def test_conv_bf16_tile256_deterministic(ops):
    x, w, b = rnd(2, 128, 24, 24, seed=7), rnd(256, 128, 3, 3, seed=8, scale=0.03), rnd(256, seed=9)
    ops.set_tile256(1)
    first = ops.conv(nhwc(x), w, b, bf16="dma")
    for _ in range(3):
        assert np.array_equal(ops.conv(nhwc(x), w, b, bf16="dma"), first)


def test_conv_bf16_tile256_fused_epilogue(ops):
    """GroupNorm + GELU operand pass, then bias + time bias + activation + mul + rowscale + residual in the shared epilogue."""
    B, H, W, C, Co = 2, 16, 24, 128, 256
    x = rnd(B, C, H, W, seed=4, scale=2.0) + 0.5
    gamma, beta = rnd(C, seed=5) * 0.2 + 1, rnd(C, seed=6) * 0.1
    w, b = rnd(Co, C, 3, 3, seed=7, scale=0.05), rnd(Co, seed=8)
    tb, res, mul, rs = rnd(B, Co, seed=9), rnd(B, Co, H, W, seed=10), rnd(B, Co, H, W, seed=11), rnd(B, seed=12)
    ref = O.conv2d(O.gelu(O.group_norm(x, 8, gamma, beta)), w, b, 1) + tb[:, :, None, None]
    ref = 1 / (1 + np.exp(-ref)) * mul * rs[:, None, None, None] + res
    sc, sh = ops.gn_scale_shift(nhwc(x), 8, gamma, beta)
    got, old = both(ops, lambda: ops.conv(nhwc(x), w, b, scale=sc, shift=sh, pro_act="gelu", tb=tb, epi_act="sigmoid", mul=nhwc(mul),
                                          rowscale=rs, res=nhwc(res), bf16="dma"))
    close(nchw(got), ref, 2e-2)
    close(got, old, 2e-5)


@pytest.mark.parametrize("B,H,W,C1,C2,Cout", [(1, 16, 16, 256, 128, 256), (2, 12, 20, 512, 256, 512), (3, 9, 13, 256, 1024, 256),
                                              (1, 24, 24, 512, 64, 1024)])
def test_conv_bf16_tile256_second_k_segment(ops, B, H, W, C1, C2, Cout):
    """conv_out with the block's 1x1 shortcut as a second K segment (down3 / down4 / up2 / bottleneck.0 shapes)."""
    x1, x2 = rnd(B, C1, H, W, seed=30), rnd(B, C2, H, W, seed=31)
    w1 = rnd(Cout, C1, 3, 3, seed=32, scale=1.0 / np.sqrt(9 * C1))
    w2 = rnd(Cout, C2, seed=33, scale=1.0 / np.sqrt(C2))
    b = rnd(Cout, seed=34)
    got, old = both(ops, lambda: ops.conv_dual_bf16(nhwc(x1), w1, nhwc(x2), w2, b))
    ref = O.conv2d(bf(x1), bf(w1), b, 1) + O.conv2d(bf(x2), bf(w2)[:, :, None, None], np.zeros(Cout, np.float32), 0)
    close(nchw(got), ref, 4e-5)
    close(got, old, 3e-5)


@pytest.mark.parametrize("B,H,W,Cin,Cout,G", [(8, 64, 64, 64, 256, 8), (4, 64, 64, 128, 512, 8), (16, 32, 32, 64, 256, 4)])
def test_conv_bf16_tile256_fused_groupnorm_statistics(ops, B, H, W, Cin, Cout, G):
    """the unsplit path (statistics forbid the K split): the epilogue's per-(sample, group) sums are the statistics of the
    tensor it stored; one writer per (32-row block, group) also when a wave owns four such blocks."""
    x = rnd(B, Cin, H, W, seed=50)
    w = rnd(Cout, Cin, 3, 3, seed=51, scale=1.0 / np.sqrt(Cin * 9))
    b = rnd(Cout, seed=52) + 0.5
    (out, sums), (out0, sums0) = both(ops, lambda: ops.conv_gnstats_bf16(nhwc(x), w, b, G))
    o = out.astype(np.float64).reshape(B, H * W, G, Cout // G)
    want = np.stack([o.sum(axis=(1, 3)), (o * o).sum(axis=(1, 3))], axis=-1)
    assert np.allclose(sums, want, rtol=2e-5, atol=1e-2), np.abs(sums - want).max()
    close(out, out0, 2e-5)
    assert np.allclose(sums, sums0, rtol=1e-4, atol=1e-1)


def test_conv_fp8_tile256_is_exact_on_integer_data(ops):
    """e4m3 operands (v_mfma_scale_f32_32x32x64_f8f6f4) in the eight-wave tile: with exactly representable operands every
    product and partial sum is exact in fp32, so the result must EQUAL the fp32 convolution -- any wrong lane / k mapping,
    swizzle, tap order, border mask or split-K fold breaks it.  Ragged M, odd / even K-tile counts, several column tiles."""
    rng = np.random.default_rng(11)
    for (b, h, w, cin, cout) in ((2, 12, 16, 256, 256), (1, 16, 16, 128, 512), (3, 16, 16, 384, 256), (5, 8, 8, 128, 256), (1, 40, 40, 256, 1024)):
        x = rng.integers(-4, 5, (b, h, w, cin)).astype(np.float32)
        wt = (rng.integers(-3, 4, (cout, cin, 3, 3)) * 0.125).astype(np.float32)
        wt[0, 0, 0, 0] = 0.375                                   # |w| max fixed -> the tensor scale is a known power of two
        bias = rng.integers(-2, 3, cout).astype(np.float32)
        ref = O.conv2d(x.transpose(0, 3, 1, 2), wt, bias, 1).transpose(0, 2, 3, 1)
        got, old = both(ops, lambda: ops.conv(x, wt, bias, bf16="fp8"))
        assert np.array_equal(got, ref), (b, h, w, cin, cout, float(np.abs(got - ref).max()))
        assert np.array_equal(old, ref)


def test_conv_fp8_second_k_segment_is_exact_on_integer_data(ops):
    """fp8 mode's conv_out: e4m3 3x3 segment + bf16 1x1 shortcut segment under one accumulator scale, on the 128-row kernel and
    (shapes with whole tile pairs of each kind) on the eight-wave tile, whose e4m3 phases hand over to bf16 phases mid-stream.
    Exactly representable data -> equality with the fp32 convolution."""
    rng = np.random.default_rng(13)
    for (b, h, w, c1, c2, cout, tile) in ((2, 12, 16, 256, 128, 256, True), (1, 16, 16, 512, 256, 512, True), (3, 16, 16, 256, 1024, 256, True),
                                          (2, 8, 12, 128, 64, 128, False), (1, 16, 16, 128, 64, 256, False)):
        x1 = rng.integers(-4, 5, (b, h, w, c1)).astype(np.float32)
        x2 = rng.integers(-6, 7, (b, h, w, c2)).astype(np.float32)
        w1 = (rng.integers(-3, 4, (cout, c1, 3, 3)) * 0.125).astype(np.float32)
        w1[0, 0, 0, 0] = 0.375
        w2 = (rng.integers(-5, 6, (cout, c2)) * 0.0625).astype(np.float32)
        bias = rng.integers(-2, 3, cout).astype(np.float32)
        ref = (O.conv2d(x1.transpose(0, 3, 1, 2), w1, bias, 1) +
               O.conv2d(x2.transpose(0, 3, 1, 2), w2[:, :, None, None], np.zeros(cout, np.float32), 0)).transpose(0, 2, 3, 1)
        n0 = ops.gemm256_launches()
        ops.set_tile256(1)
        got = ops.conv_dual_fp8(x1, w1, x2, w2, bias)
        assert (ops.gemm256_launches() > n0) == tile, (c1, c2, cout)
        ops.set_tile256(0)
        old = ops.conv_dual_fp8(x1, w1, x2, w2, bias)
        assert np.array_equal(got, ref), (b, h, w, c1, c2, cout, float(np.abs(got - ref).max()))
        assert np.array_equal(old, ref)


def test_conv_tile256_timing_op_runs_both_kernels(ops):
    """ops.conv_time drives the same launcher (tools/ and bench use it for same-process A/B): both settings run and the forced
    one is counted."""
    n0 = ops.gemm256_launches()
    ops.set_tile256(1)
    t1 = ops.conv_time(2, 32, 32, 256, 256, 3, "dma", True, False, 0, 2)
    t8 = ops.conv_time(2, 32, 32, 256, 256, 3, "fp8", True, False, 0, 2)
    assert ops.gemm256_launches() >= n0 + 6
    ops.set_tile256(0)
    t0 = ops.conv_time(2, 32, 32, 256, 256, 3, "dma", True, False, 0, 2)
    assert t0 > 0 and t1 > 0 and t8 > 0
