"""Per-operator parity: every HIP kernel on the hot path against the numpy oracle (rows a1-a13 of
SURVEY.md section 8a).  fp32 tolerance 1e-4 relative to the output scale unless an op is bit-exact."""
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


def close(a, b, tol=1e-4):
    scale = max(1.0, float(np.abs(b).max()))
    err = float(np.abs(a - b).max())
    assert err <= tol * scale, f"max err {err} (scale {scale})"


@pytest.mark.parametrize("B,H,W,Cin,Cout,k", [
    (2, 16, 16, 64, 128, 3), (1, 32, 32, 3, 64, 3), (3, 8, 8, 128, 64, 3), (2, 4, 4, 512, 256, 3),
    (4, 1, 1, 512, 1024, 3), (2, 2, 2, 1024, 512, 3), (1, 12, 20, 64, 3, 3), (2, 16, 16, 128, 384, 1),
    (2, 16, 16, 64, 32, 1), (1, 6, 10, 256, 96, 1), (5, 7, 3, 36, 40, 3)])
def test_conv_plain(ops, B, H, W, Cin, Cout, k):
    x = rnd(B, Cin, H, W, seed=1)
    w = rnd(Cout, Cin, k, k, seed=2, scale=1.0 / np.sqrt(Cin * k * k))
    b = rnd(Cout, seed=3)
    ref = O.conv2d(x, w, b, k // 2)
    got = ops.conv(nhwc(x), w, b)
    close(nchw(got), ref)


@pytest.mark.parametrize("B,H,W,Cin,Cout,k", [
    (2, 16, 16, 64, 128, 3), (1, 32, 32, 3, 64, 3), (2, 4, 4, 512, 256, 3), (4, 1, 1, 512, 1024, 3),
    (1, 12, 20, 64, 3, 3), (2, 16, 16, 128, 384, 1), (2, 16, 16, 64, 32, 1), (5, 7, 3, 36, 40, 3), (1, 6, 10, 68, 96, 1)])
def test_conv_bf16(ops, B, H, W, Cin, Cout, k):
    """bf16 matrix-core variant against the fp32 oracle evaluated on bf16-rounded operands (exact products,
    fp32 accumulation): only the summation order differs."""
    def bf(z):
        u = np.ascontiguousarray(z, dtype=np.float32).view(np.uint32)
        return ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32).view(np.float32)
    x = rnd(B, Cin, H, W, seed=1)
    w = rnd(Cout, Cin, k, k, seed=2, scale=1.0 / np.sqrt(Cin * k * k))
    b = rnd(Cout, seed=3)
    ref = O.conv2d(bf(x), bf(w), b, k // 2)
    got = ops.conv(nhwc(x), w, b, bf16=True)
    close(nchw(got), ref, 2e-5)
    close(nchw(got), O.conv2d(x, w, b, k // 2), 2e-2)


@pytest.mark.parametrize("H,W,bs,low", [(16, 16, 4, 3), (24, 40, 8, 4), (2, 2, 8, 4), (5, 13, 4, 3)])
def test_conv_bf16_dma_gate_select(ops, H, W, bs, low):
    """frequency gates as one block-diagonal GEMM: per pixel keep sigmoid(low half) or boost * sigmoid(high half)
    according to the static in-block mask (webp_inference.py:241-252), times the DCT tensor, plus the residual."""
    B, C = 2, 64
    hid = rnd(B, C, H, W, seed=20)
    w = rnd(2 * C, C, 1, 1, seed=21, scale=0.2)
    w[:C, C // 2:] = 0; w[C:, :C // 2] = 0            # block diagonal: low gate reads hidden[:C/2], high gate the rest
    b = rnd(2 * C, seed=22, scale=0.3)
    xd, hb, boost = rnd(B, C, H, W, seed=23), rnd(B, C, H, W, seed=24), np.abs(rnd(B, seed=25)) + 0.5
    got = ops.conv(nhwc(hid), w, b, bs=bs, low=low, epi_act="sigmoid", mul=nhwc(xd), rowscale=boost, res=nhwc(hb),
                   bf16="dma", gate_select=C)
    def bf(z):
        u = np.ascontiguousarray(z, dtype=np.float32).view(np.uint32)
        return ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32).view(np.float32)
    z = O.conv2d(bf(hid), bf(w), b, 0)
    sig = 1 / (1 + np.exp(-z.astype(np.float64)))
    m = O.low_mask(H, W, bs, low)[None, None]
    ref = hb + xd * np.where(m > 0, sig[:, :C], sig[:, C:] * boost[:, None, None, None])
    close(nchw(got), ref.astype(np.float32), 2e-5)


@pytest.mark.parametrize("B,H,W,C1,C2,Cout", [(2, 16, 16, 64, 128, 64), (1, 8, 8, 512, 1024, 512), (2, 32, 32, 64, 3, 64),
                                              (1, 4, 4, 1024, 512, 1024), (3, 5, 9, 40, 24, 36), (2, 128, 128, 64, 128, 64),
                                              (1, 192, 176, 64, 8, 48), (2, 256, 256, 64, 3, 64), (3, 200, 232, 64, 24, 64)])
def test_conv_bf16_dma_second_k_segment(ops, B, H, W, C1, C2, Cout):
    """3x3 conv_out with the block's 1x1 shortcut conv folded in as a second K segment (incl. split-K shapes; the last two:
    64 -> 64 at full-resolution sizes with the 3-channel image operand / a 24-channel operand and a ragged last tile)."""
    def bf(z):
        u = np.ascontiguousarray(z, dtype=np.float32).view(np.uint32)
        return ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32).view(np.float32)
    x1, x2 = rnd(B, C1, H, W, seed=30), rnd(B, C2, H, W, seed=31)
    w1 = rnd(Cout, C1, 3, 3, seed=32, scale=1.0 / np.sqrt(9 * C1))
    w2 = rnd(Cout, C2, seed=33, scale=1.0 / np.sqrt(C2))
    b = rnd(Cout, seed=34)
    got = ops.conv_dual_bf16(nhwc(x1), w1, nhwc(x2), w2, b)
    ref = O.conv2d(bf(x1), bf(w1), b, 1) + O.conv2d(bf(x2), bf(w2)[:, :, None, None], np.zeros(Cout, np.float32), 0)
    close(nchw(got), ref, 4e-5)


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,G", [(2, 128, 128, 64, 64, 3, 8), (8, 64, 64, 32, 128, 1, 8), (8, 64, 64, 64, 256, 3, 8),
                                                (2, 128, 128, 8, 32, 3, 8), (16, 64, 32, 16, 512, 1, 8)])
def test_conv_bf16_dma_fused_groupnorm_statistics(ops, B, H, W, Cin, Cout, k, G):
    """the epilogue's per-(sample, group) sum / sum of squares must be the statistics of the tensor it stored."""
    x = rnd(B, Cin, H, W, seed=50)
    w = rnd(Cout, Cin, k, k, seed=51, scale=1.0 / np.sqrt(Cin * k * k))
    b = rnd(Cout, seed=52) + 0.5
    out, sums = ops.conv_gnstats_bf16(nhwc(x), w, b, G)
    o = out.astype(np.float64).reshape(B, H * W, G, Cout // G)
    want = np.stack([o.sum(axis=(1, 3)), (o * o).sum(axis=(1, 3))], axis=-1)
    assert np.allclose(sums, want, rtol=2e-5, atol=1e-2), np.abs(sums - want).max()
    n = H * W * (Cout // G)                                  # what nn.GroupNorm derives from them
    mean, mean_w = sums[..., 0] / n, want[..., 0] / n
    var, var_w = sums[..., 1] / n - mean ** 2, want[..., 1] / n - mean_w ** 2
    assert np.abs(mean - mean_w).max() < 1e-5 and np.abs(var - var_w).max() < 1e-5


def test_conv_bf16_fused_prologue_epilogue(ops):
    B, H, W, C, Co = 2, 16, 16, 64, 128
    x = rnd(B, C, H, W, seed=4, scale=2.0) + 0.5
    gamma, beta = rnd(C, seed=5) * 0.2 + 1, rnd(C, seed=6) * 0.1
    w, b = rnd(Co, C, 3, 3, seed=7, scale=0.05), rnd(Co, seed=8)
    tb, res = rnd(B, Co, seed=9), rnd(B, Co, H, W, seed=10)
    ref = O.conv2d(O.gelu(O.group_norm(x, 8, gamma, beta)), w, b, 1) + tb[:, :, None, None] + res
    sc, sh = ops.gn_scale_shift(nhwc(x), 8, gamma, beta)
    got = ops.conv(nhwc(x), w, b, scale=sc, shift=sh, pro_act="gelu", tb=tb, res=nhwc(res), bf16=True)
    close(nchw(got), ref, 2e-2)


def test_conv_fused_norm_gelu_timebias_residual(ops):
    """conv2 of a block: GroupNorm -> GELU applied in the loader, bias + time bias + residual in the epilogue."""
    B, H, W, C, Co = 2, 16, 16, 64, 128
    x = rnd(B, C, H, W, seed=4, scale=2.0) + 0.5
    gamma, beta = rnd(C, seed=5) * 0.2 + 1, rnd(C, seed=6) * 0.1
    w, b = rnd(Co, C, 3, 3, seed=7, scale=0.05), rnd(Co, seed=8)
    tb, res = rnd(B, Co, seed=9), rnd(B, Co, H, W, seed=10)
    ref = O.conv2d(O.gelu(O.group_norm(x, 8, gamma, beta)), w, b, 1) + tb[:, :, None, None] + res
    sc, sh = ops.gn_scale_shift(nhwc(x), 8, gamma, beta)
    got = ops.conv(nhwc(x), w, b, scale=sc, shift=sh, pro_act="gelu", tb=tb, res=nhwc(res))
    close(nchw(got), ref)


@pytest.mark.parametrize("C,G,HW", [(3, 3, 1024), (64, 8, 4096), (128, 8, 256), (1024, 8, 1), (512, 8, 4), (64, 8, 9000)])
def test_group_norm_scale_shift(ops, C, G, HW):
    x = rnd(2, C, HW, 1, seed=11, scale=3.0) + 1.5
    gamma, beta = rnd(C, seed=12) * 0.2 + 1, rnd(C, seed=13) * 0.1
    ref = O.group_norm(x, G, gamma, beta)
    sc, sh = ops.gn_scale_shift(nhwc(x), G, gamma, beta)
    got = x * sc[:, :, None, None] + sh[:, :, None, None]
    close(got, ref, 2e-5)


@pytest.mark.parametrize("act", ["silu", "gelu"])
def test_conv_head_nchw_tanh(ops, act):
    B, H, W, C = 2, 8, 8, 64
    x = rnd(B, C, H, W, seed=14)
    gamma, beta = np.ones(C, np.float32), np.zeros(C, np.float32)
    w, b = rnd(3, C, 3, 3, seed=15, scale=0.05), rnd(3, seed=16)
    f = O.silu if act == "silu" else O.gelu
    ref = np.tanh(O.conv2d(f(O.group_norm(x, 8, gamma, beta)), w, b, 1))
    sc, sh = ops.gn_scale_shift(nhwc(x), 8, gamma, beta)
    got = ops.conv(nhwc(x), w, b, scale=sc, shift=sh, pro_act=act, epi_act="tanh", nchw_out=True)
    close(got, ref, 2e-5)


@pytest.mark.parametrize("H,W,bs,low", [(16, 16, 4, 3), (16, 16, 8, 4), (2, 2, 4, 3), (1, 1, 8, 4), (4, 4, 8, 4), (6, 10, 4, 3)])
def test_frequency_gates(ops, H, W, bs, low):
    """low/high gate pair of the frequency block (a8): masks in the GEMM loader and epilogue."""
    B, C = 2, 64
    xd = rnd(B, C, H, W, seed=17)
    base = rnd(B, C, H, W, seed=18)
    w1, b1 = rnd(C // 2, C, 1, 1, seed=19, scale=0.1), rnd(C // 2, seed=20)
    w2, b2 = rnd(C, C // 2, 1, 1, seed=21, scale=0.1), rnd(C, seed=22)
    boost = np.array([0.3, 1.7], np.float32)
    m = O.low_mask(H, W, bs, low)[None, None]
    for which, mask in (("low", m), ("high", ~m)):
        z = np.where(mask, xd, 0).astype(np.float32)
        hmid = O.conv2d(z, w1, b1, 0)
        hmid = np.where(hmid > 0, hmid, 0.2 * hmid).astype(np.float32)
        gate = O.sigmoid(O.conv2d(hmid, w2, b2, 0))
        if which == "high":
            gate = gate * boost[:, None, None, None]
        ref = base + gate * z
        hid = ops.conv(nhwc(xd), w1, b1, pro_mask=which, bs=bs, low=low, epi_act="lrelu")
        got = ops.conv(hid, w2, b2, epi_act="sigmoid", mul=nhwc(xd), epi_mask=which, bs=bs, low=low,
                       rowscale=boost if which == "high" else None, res=nhwc(base))
        close(nchw(got), ref, 2e-5)


@pytest.mark.parametrize("N,C,heads", [(256, 64, 4), (1024, 64, 4), (256, 128, 4), (64, 256, 4), (64, 512, 4),
                                       (16, 512, 4), (4, 1024, 4), (1, 1024, 4), (96, 64, 8), (200, 128, 8),
                                       (64, 1024, 8), (4, 1024, 8), (64, 1024, 4), (256, 1024, 4), (100, 1024, 4)])
def test_attention(ops, N, C, heads):
    B = 2
    qkv = rnd(B, N, 3 * C, seed=23, scale=1.5)
    d = C // heads
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    sp = lambda z: z.reshape(B, N, heads, d).transpose(0, 2, 1, 3)
    s = (sp(q) / np.sqrt(d)) @ sp(k).transpose(0, 1, 3, 2)
    s = s - s.max(-1, keepdims=True)
    p = np.exp(s)
    p /= p.sum(-1, keepdims=True)
    ref = (p @ sp(v)).transpose(0, 2, 1, 3).reshape(B, N, C)
    got = ops.attention(qkv, heads)
    close(got, ref, 2e-5)


@pytest.mark.parametrize("N,C,heads", [(256, 64, 4), (1024, 64, 4), (4096, 64, 4), (256, 128, 4), (1024, 128, 4),
                                       (64, 256, 4), (576, 256, 4), (64, 512, 4), (256, 512, 4), (96, 64, 8),
                                       (200, 128, 8), (1000, 64, 4), (2048, 64, 8)])
def test_attention_bf16(ops, N, C, heads):
    """bf16 matrix-core kernel: operands rounded to bf16, fp32 softmax statistics and accumulation."""
    B = 2
    qkv = rnd(B, N, 3 * C, seed=40, scale=1.5)
    d = C // heads
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    sp = lambda z: z.reshape(B, N, heads, d).transpose(0, 2, 1, 3).astype(np.float64)
    s = (sp(q) / np.sqrt(d)) @ sp(k).transpose(0, 1, 3, 2)
    s = s - s.max(-1, keepdims=True)
    p = np.exp(s)
    p /= p.sum(-1, keepdims=True)
    ref = (p @ sp(v)).transpose(0, 2, 1, 3).reshape(B, N, C).astype(np.float32)
    got = ops.attention(qkv, heads, bf16=True)
    err = np.abs(got - ref).max()
    assert err < 8e-2, err            # bf16 operands: 2^-8 relative on q, k, v, p
    assert np.abs(got - ref).mean() < 4e-3


@pytest.mark.parametrize("B,N,C,heads", [(2, 256, 64, 4), (1, 4096, 64, 4), (3, 96 * 32, 64, 4), (5, 64, 64, 4),
                                         (2, 256, 128, 4), (1, 4096, 128, 4), (3, 40 * 32, 128, 4), (4, 64, 128, 4),
                                         (2, 256, 64, 8), (1, 4096, 64, 8), (3, 40 * 32, 128, 8), (2, 1024, 256, 8), (5, 64, 64, 8)])
def test_qkv_projection_fused_with_attention_packing(ops, B, N, C, heads):
    """in_proj whose epilogue writes the attention operand images (Q / K / V^T tiles, |k| bounds) must give exactly
    what the separate fp32-qkv + pack_qkv path gives (same GEMM, same roundings), and match nn.MultiheadAttention's
    core (webp_inference.py:317-321; 8 heads: avif_inference.py:281) within bf16 operand precision.  head_dim 8, 16 and 32."""
    x = rnd(B, N, C, seed=60, scale=1.2)
    w = rnd(3 * C, C, seed=61, scale=1.0 / np.sqrt(C)) * 1.5
    b = rnd(3 * C, seed=62, scale=0.3)
    fused = ops.qkv_attention_bf16(x, w, b, heads, fused=True)
    plain = ops.qkv_attention_bf16(x, w, b, heads, fused=False)
    dd = np.abs(fused - plain)
    assert np.array_equal(fused, plain), (dd.max(), int((dd > 0).sum()), np.argwhere(dd > 0)[:6].tolist())
    def bf(z):   # the GEMM consumes bf16-rounded x and W (exact products, fp32 accumulation)
        u = np.ascontiguousarray(z, dtype=np.float32).view(np.uint32)
        return ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32).view(np.float32)
    qkv = (bf(x).astype(np.float64) @ bf(w).T.astype(np.float64) + b).astype(np.float32)
    d = C // heads
    sp = lambda z: z.reshape(B, N, heads, d).transpose(0, 2, 1, 3).astype(np.float64)
    s = (sp(qkv[..., :C]) / np.sqrt(d)) @ sp(qkv[..., C:2 * C]).transpose(0, 1, 3, 2)
    pr = np.exp(s - s.max(-1, keepdims=True))
    pr /= pr.sum(-1, keepdims=True)
    ref = (pr @ sp(qkv[..., 2 * C:])).transpose(0, 2, 1, 3).reshape(B, N, C).astype(np.float32)
    assert np.abs(fused - ref).max() < 0.15          # bf16 rounding of q, k, v, p under a fairly sharp softmax
    assert np.abs(fused - ref).mean() < 5e-3


def test_attention_bf16_spiky_and_shifted(ops):
    """forces the deferred-rescale branch mid-sequence and a large negative common offset (first-tile init)."""
    B, N, C, heads = 1, 2048, 64, 4
    qkv = rnd(B, N, 3 * C, seed=41)
    qkv[0, 7, :16] *= 25
    qkv[0, 1500, C:C + 16] = qkv[0, 7, :16] * 2      # score ~ +2*|q|^2/4 far above everything before it
    qkv[0, :, 16:32] = 6.0                           # head 1: q constant
    qkv[0, :, C + 16:C + 32] = -6.0                  # head 1: k constant negative -> all scores ~ -144
    d = C // heads
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    bf = lambda z: (z.astype(np.float32).view(np.uint32) + 0x7FFF + ((z.astype(np.float32).view(np.uint32) >> 16) & 1) & 0xFFFF0000).view(np.float32)
    sp = lambda z: z.reshape(B, N, heads, d).transpose(0, 2, 1, 3).astype(np.float64)
    s = (sp(q) / np.sqrt(d)) @ sp(k).transpose(0, 1, 3, 2)
    s = s - s.max(-1, keepdims=True)
    p = np.exp(s)
    p /= p.sum(-1, keepdims=True)
    ref = (p @ sp(v)).transpose(0, 2, 1, 3).reshape(B, N, C).astype(np.float32)
    got = ops.attention(qkv, heads, bf16=True)
    assert np.isfinite(got).all()
    # row 7 of head 0 is a one-hot on key 1500 (bf16 rounding of v only)
    assert np.abs(got[0, 7, :16] - ref[0, 7, :16]).max() < 3e-2
    assert np.abs(got - ref).mean() < 6e-3


def test_attention_spiky_softmax(ops):
    """a query/key pair with a huge score forces the online-softmax rescale path mid-sequence."""
    B, N, C, heads = 1, 512, 64, 4
    qkv = rnd(B, N, 3 * C, seed=24)
    qkv[0, 5, :16] *= 30
    qkv[0, 300, C:C + 16] = qkv[0, 5, :16]
    d = C // heads
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    sp = lambda z: z.reshape(B, N, heads, d).transpose(0, 2, 1, 3).astype(np.float64)
    s = (sp(q) / np.sqrt(d)) @ sp(k).transpose(0, 1, 3, 2)
    s = s - s.max(-1, keepdims=True)
    p = np.exp(s)
    p /= p.sum(-1, keepdims=True)
    ref = (p @ sp(v)).transpose(0, 2, 1, 3).reshape(B, N, C)
    close(ops.attention(qkv, heads), ref.astype(np.float32), 2e-5)


@pytest.mark.parametrize("H,W,bs", [(16, 16, 4), (16, 16, 8), (2, 2, 4), (1, 1, 8), (4, 4, 8), (6, 10, 4), (12, 20, 8)])
def test_block_dct(ops, H, W, bs):
    from ddpm_image_restoration_amd.synth import dct_matrix
    x = rnd(2, 64, H, W, seed=25)
    D = dct_matrix(bs)
    close(nchw(ops.block_transform(nhwc(x), D)), O.blockwise_transform(x, D), 1e-5)
    close(nchw(ops.block_transform(nhwc(x), D, 1.0, 0.1)), x + np.float32(0.1) * O.blockwise_transform(x, D), 1e-5)


def test_block_transform_per_channel(ops):
    x = rnd(2, 32, 12, 20, seed=26)
    T = rnd(32, 8, 8, seed=27, scale=0.35)
    close(nchw(ops.block_transform(nhwc(x), T)), O.blockwise_transform(x, T), 1e-5)


def test_pool_and_resample(ops):
    x = rnd(2, 64, 12, 20, seed=28)
    assert np.array_equal(nchw(ops.maxpool2(nhwc(x))), O.max_pool2(x))
    close(nchw(ops.upsample2(nhwc(x))), O.upsample2(x), 1e-6)
    for s in (1, 2, 4, 8):
        close(nchw(ops.adaptive_avgpool(nhwc(x), s)), O.adaptive_avg_pool(x, s), 1e-6)
    for shape in ((3, 64, 32, 48), (1, 1024, 16, 16), (2, 256, 64, 64)):   # nested bins: one pass feeds all four scales
        xb = rnd(*shape, seed=27)
        for s in (1, 2, 4, 8):
            close(nchw(ops.adaptive_avgpool(nhwc(xb), s)), O.adaptive_avg_pool(xb, s), 2e-6)
    small = rnd(2, 64, 2, 2, seed=29)
    for s in (1, 2, 4, 8):
        a = O.adaptive_avg_pool(small, s)
        close(nchw(ops.adaptive_avgpool(nhwc(small), s)), a, 1e-6)
        close(nchw(ops.bilinear(nhwc(a), 2, 2)), O.bilinear(a, 2, 2), 1e-6)
    a8 = O.adaptive_avg_pool(x, 8)
    close(nchw(ops.bilinear(nhwc(a8), 12, 20)), O.bilinear(a8, 12, 20), 1e-6)


def test_bf16_activation_stream_passes(ops):
    """bf16 / fp8 mode: what flows between residual blocks is bf16 (as under the reference's autocast), and the max-pool /
    bilinear x2 passes also leave the GroupNorm sums of what they wrote.  Max-pool of bf16 values is exact; bilinear x2 equals the
    oracle's interpolation of the rounded input rounded once more; the folded scale / shift equal GroupNorm statistics of the
    produced bf16 tensor (oracle group_norm: webp_inference.py:277-279)."""
    def bf16(a):
        u = np.ascontiguousarray(a, np.float32).view(np.uint32)
        return ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32).view(np.float32)

    def affine(t_nchw, G, gamma, beta):   # per (b, c) scale / shift such that group_norm(t) = t * scale + shift
        B, C = t_nchw.shape[:2]
        g = t_nchw.astype(np.float64).reshape(B, G, -1)
        mean, var = g.mean(-1), g.var(-1)
        rstd = 1.0 / np.sqrt(var + 1e-5)
        sc = np.repeat(rstd, C // G, axis=1) * gamma[None]
        sh = beta[None] - np.repeat(mean, C // G, axis=1) * sc
        return sc, sh
    for (B, C, H, W, G) in ((2, 64, 16, 24, 8), (3, 128, 8, 8, 8), (1, 512, 4, 6, 4), (2, 1024, 2, 2, 8), (2, 64, 64, 32, 4)):
        x = rnd(B, C, H, W, seed=40 + C, scale=1.5) + 0.3
        xr = bf16(x)
        gamma, beta = rnd(C, seed=41) * 0.2 + 1.0, rnd(C, seed=42) * 0.1
        for mode, ref in ((0, xr), (1, O.max_pool2(xr)), (2, bf16(O.upsample2(xr)))):
            out, sc, sh = ops.stream16(mode, nhwc(x), G, gamma, beta)
            if mode == 1:
                assert np.array_equal(nchw(out), ref)
            elif mode == 2:   # fp32 interpolation order differs from numpy's in the last bit: at most one bf16 ulp
                d = np.abs(nchw(out) - ref)
                assert d.max() <= np.abs(ref).max() * 2.0 ** -7 and (d > 0).mean() < 0.02, (d.max(), (d > 0).mean())
                ref = nchw(out)   # statistics are those of the tensor the pass wrote
            esc, esh = affine(ref, G, gamma.astype(np.float64), beta.astype(np.float64))
            close(sc, esc, 2e-5)
            close(sh, esh, 2e-5)
        # the fp32 stream (bf16 mode's default): same passes on fp32 tensors, statistics of the fp32 result
        for mode, ref in ((1, O.max_pool2(x)), (2, O.upsample2(x))):
            out, sc, sh = ops.stream16(mode, nhwc(x), G, gamma, beta, f32=True)
            if mode == 1:
                assert np.array_equal(nchw(out), ref)
            else:
                close(nchw(out), ref, 1e-6)
            esc, esh = affine(nchw(out), G, gamma.astype(np.float64), beta.astype(np.float64))
            close(sc, esc, 2e-5)
            close(sh, esh, 2e-5)


def test_time_embedding(ops, synth_state):
    sd = synth_state(0)
    t = np.array([0.0, 0.37, 0.999, 0.5], np.float32)
    m = O.Model(0, sd)
    ref = m.time_embedding(t)
    got = ops.time_embed(t, sd["time_embed.proj.0.weight"], sd["time_embed.proj.0.bias"],
                         sd["time_embed.proj.2.weight"], sd["time_embed.proj.2.bias"])
    close(got, ref, 1e-5)


def test_quantize_bit_exact(ops):
    x = np.concatenate([rnd(100000, seed=30, scale=0.7), np.linspace(-1.2, 1.2, 50001, dtype=np.float32),
                        ((np.arange(256, dtype=np.float32) - 127.5) / 127.5).astype(np.float32)])
    assert np.array_equal(ops.quantize_u8(x), O.to_u8(x))


def test_ddrm_update_bit_exact(ops):
    B, n = 3, 3 * 32 * 32
    th, y, nz = rnd(B, n, seed=31, scale=0.5), rnd(B, n, seed=32, scale=0.5), rnd(B, n, seed=33)
    dec = np.random.default_rng(34).integers(0, 256, (B, n), dtype=np.uint8)
    for eta_b in (1.0, 0.6):
        for i, steps in ((7, 50), (0, 50), (999, 1000)):
            t = np.float32(i) / np.float32(steps)
            xp = (th - O.from_u8(dec)) + y
            if i > 0:
                rn = nz * (t * np.float32(0.2))
                ref = (np.float32(eta_b) * xp + np.float32(1 - eta_b) * th) + np.float32(0.85) * rn
            else:
                ref = xp
            got = ops.ddrm_update(th, dec, y, nz, 0.85, eta_b, 0.2, i, steps)
            assert np.array_equal(got, ref.astype(np.float32)), (eta_b, i)


def test_randn_moments_and_shard_independence(ops):
    a = ops.randn(1 << 20, seed=7)
    assert abs(float(a.mean())) < 5e-3 and abs(float(a.std()) - 1) < 5e-3
    assert abs(float((a ** 4).mean()) - 3) < 0.05
    b = ops.randn(1 << 19, seed=7, offset=(1 << 19) // 4)
    assert np.array_equal(a[1 << 19:], b)       # second half drawn as its own shard: same numbers


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,Cin,Cout,k", [
    (1, 12, 20, 64, 64, 3), (2, 16, 16, 128, 384, 1), (2, 16, 16, 64, 32, 1), (5, 7, 3, 36, 40, 3), (1, 6, 10, 68, 96, 1),
    (1, 4, 4, 1024, 1024, 3), (3, 2, 2, 200, 132, 3), (1, 1, 1, 16, 8, 3), (2, 32, 32, 3, 64, 3), (1, 9, 130, 72, 128, 3),
    (3, 5, 9, 3, 64, 3), (1, 1, 1, 3, 16, 3), (2, 16, 24, 8, 40, 3),
    (2, 128, 128, 64, 64, 3), (1, 160, 224, 32, 40, 3), (4, 96, 96, 128, 64, 3), (3, 128, 96, 64, 16, 3), (1, 256, 130, 72, 64, 3),
    (2, 256, 256, 64, 64, 3), (3, 200, 232, 64, 64, 3), (9, 128, 128, 64, 64, 3)])
def test_conv_bf16_dma(ops, B, H, W, Cin, Cout, k):
    """LDS-DMA bf16 GEMM (gemm16.hip), including the split-K atomics path (deep, few output tiles), channel tails
    (Cin % 64 != 0, Cin = 3 padded to 8: the tap-packed K axis), ragged M and Cout tiles, the row-strip kernel of the large
    3x3 convs with <= 64 output channels (tiles crossing image rows, channel tails), 1x1 maps whose every tap but the centre is
    halo, and (last three shapes, >= 131 072 pixels of 64 -> 64) the descriptor-addressed strip kernel at the sizes of the
    full-resolution levels: tiles that straddle two images, a ragged last tile (also the shapes of the experiments build's
    weights-stationary kernel)."""
    def bf(z):
        u = np.ascontiguousarray(z, dtype=np.float32).view(np.uint32)
        return ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32).view(np.float32)
    x = rnd(B, Cin, H, W, seed=1)
    w = rnd(Cout, Cin, k, k, seed=2, scale=1.0 / np.sqrt(Cin * k * k))
    b = rnd(Cout, seed=3)
    got = ops.conv(nhwc(x), w, b, bf16="dma")
    close(nchw(got), O.conv2d(bf(x), bf(w), b, k // 2), 3e-5)


@pytest.mark.gpu
def test_conv_bf16_dma_fused_epilogue(ops):
    """GroupNorm+GELU operand (gn_apply16) and bias + time bias + activation + mul + rowscale + residual epilogue."""
    B, H, W, C, Co = 2, 16, 16, 64, 128
    x = rnd(B, C, H, W, seed=4, scale=2.0) + 0.5
    gamma, beta = rnd(C, seed=5) * 0.2 + 1, rnd(C, seed=6) * 0.1
    w, b = rnd(Co, C, 3, 3, seed=7, scale=0.05), rnd(Co, seed=8)
    tb, res, mul, rs = rnd(B, Co, seed=9), rnd(B, Co, H, W, seed=10), rnd(B, Co, H, W, seed=11), rnd(B, seed=12)
    ref = O.conv2d(O.gelu(O.group_norm(x, 8, gamma, beta)), w, b, 1) + tb[:, :, None, None]
    ref = 1 / (1 + np.exp(-ref)) * mul * rs[:, None, None, None] + res
    sc, sh = ops.gn_scale_shift(nhwc(x), 8, gamma, beta)
    got = ops.conv(nhwc(x), w, b, scale=sc, shift=sh, pro_act="gelu", tb=tb, epi_act="sigmoid", mul=nhwc(mul),
                   rowscale=rs, res=nhwc(res), bf16="dma")
    close(nchw(got), ref, 2e-2)





def test_conv_fp8_operand_layout_is_exact_on_integer_data(ops):
    """DDPMIR_FP8 convs (v_mfma_scale_f32_32x32x64_f8f6f4, e4m3 operands by LDS-DMA): with operands that are exactly
    representable in e4m3 (small integers; weights = small integers * 2^-k, undone by the power-of-two tensor scale) every
    product and partial sum is exact in fp32, so the result must EQUAL the fp32 convolution -- an asymmetric pattern that
    any wrong lane / k mapping, swizzle or tap order breaks.  3x3 and 1x1, ragged M, two K-chunk counts."""
    rng = np.random.default_rng(11)
    # (Cout 64 = the two-column-tile instantiation of the full-resolution convs; 64 x 64 pixels: more than 256 tiles, BK = 32)
    for (b, h, w, cin, cout, k) in ((2, 8, 12, 256, 128, 3), (1, 5, 7, 320, 96, 3), (3, 16, 16, 512, 256, 1), (2, 16, 24, 64, 64, 3),
                                    (2, 12, 16, 128, 64, 3), (9, 64, 64, 64, 64, 3)):
        x = rng.integers(-4, 5, (b, h, w, cin)).astype(np.float32)
        wt = (rng.integers(-3, 4, (cout, cin, k, k)) * 0.125).astype(np.float32)
        wt[0, 0, 0, 0] = 0.375                                   # |w| max fixed -> the tensor scale is a known power of two
        bias = rng.integers(-2, 3, cout).astype(np.float32)
        ref = O.conv2d(x.transpose(0, 3, 1, 2), wt, bias, k // 2).transpose(0, 2, 3, 1)
        got = ops.conv(x, wt, bias, bf16="fp8")
        assert np.array_equal(got, ref), (b, h, w, cin, cout, k, float(np.abs(got - ref).max()))


def test_conv_fp8_against_quantised_oracle(ops):
    """Random data: the fp8 conv equals the fp32 conv of the e4m3-rounded operands to accumulation-order accuracy, and is
    within e4m3's 2^-4 relative operand error of the unquantised conv."""
    rng = np.random.default_rng(12)
    x = rng.standard_normal((2, 16, 16, 256)).astype(np.float32)
    wt = (rng.standard_normal((256, 256, 3, 3)) * 0.02).astype(np.float32)
    bias = rng.standard_normal(256).astype(np.float32) * 0.1

    def e4m3(a):   # round to nearest even onto the e4m3 grid (3 mantissa bits, min normal 2^-6, subnormal step 2^-9, max 448)
        a = np.clip(a.astype(np.float64), -448, 448)
        e = np.floor(np.log2(np.maximum(np.abs(a), 2.0 ** -20)))
        q = np.where(e < -6, 2.0 ** -9, 2.0 ** (e - 3))
        return (np.rint(a / q) * q).astype(np.float32)
    kexp = np.floor(np.log2(448.0 / np.abs(wt).max()))
    wq = e4m3(wt * 2.0 ** kexp) * 2.0 ** -kexp
    ref_q = O.conv2d(e4m3(x).transpose(0, 3, 1, 2), wq.astype(np.float32), bias, 1).transpose(0, 2, 3, 1)
    ref = O.conv2d(x.transpose(0, 3, 1, 2), wt, bias, 1).transpose(0, 2, 3, 1)
    got = ops.conv(x, wt, bias, bf16="fp8")
    assert np.abs(got - ref_q).max() < 2e-4 * np.abs(ref).max()
    assert np.abs(got - ref).mean() < 0.05 * np.abs(ref).mean()


def test_attention_bf16_no_max_and_general_kernels(ops):
    """head_dim <= 16: pairs whose scores are bounded (|q||k| <= 80 in log2 units) run in the kernel without a running
    max, the others in the general kernel with the deferred rescale; a flag per (batch, head) routes them.  A batch that
    mixes both -- image 0 ordinary activations, image 1 with q, k scaled up until scores reach +-150 (a nearly one-hot
    softmax) -- must agree for every pair with softmax attention evaluated on the SAME bf16-rounded operands (the rounding
    of q and k moves such logits by ~0.3, so the unrounded softmax is not the yardstick there); also head_dim 8."""
    rng = np.random.default_rng(21)

    def bf16(a):
        u = np.ascontiguousarray(a, np.float32).view(np.uint32)
        return ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32).view(np.float32)
    for heads, C in ((4, 64), (8, 64)):
        B, N, d = 2, 2048, C // heads
        qkv = rng.standard_normal((B, N, 3 * C)).astype(np.float32)
        qkv[1, :, :2 * C] *= 6.0
        qkv[1, 7, C:C + d] = qkv[1, 3, :d] * 2.0                          # one dominant key for query 3 of head 0
        q, k, v = (qkv[..., i * C:(i + 1) * C].reshape(B, N, heads, d).transpose(0, 2, 1, 3) for i in range(3))
        qr = bf16(q * np.float32(1.4426950408889634 / np.sqrt(d))).astype(np.float64)
        kr, vr = bf16(k).astype(np.float64), bf16(v).astype(np.float64)
        s2 = qr @ kr.transpose(0, 1, 3, 2)                                # log2 units
        assert np.abs(s2[0]).max() < 80 < np.abs(s2[1]).max()            # image 0 is bounded, image 1 is not
        p = np.exp2(s2 - s2.max(-1, keepdims=True))
        ref = ((p / p.sum(-1, keepdims=True)) @ vr).transpose(0, 2, 1, 3).reshape(B, N, C).astype(np.float32)
        got = ops.attention(qkv, heads, bf16=True)
        assert np.isfinite(got).all()
        for b in range(B):
            e = np.abs(got[b] - ref[b])
            assert e.max() < 3e-2 and e.mean() < 2e-3, (heads, b, e.max(), e.mean())


@pytest.mark.parametrize("heads,C,N,B", [(4, 64, 65536, 1), (8, 64, 8192, 2), (4, 128, 4096, 2), (4, 256, 4096, 2)])
def test_attention_outlier_token_costs_one_query_block(ops, heads, C, N, B):
    """Routing between the no-max and the general attention kernel is per (batch, head, block of queries): ONE query token
    with a huge norm per image (what a trained checkpoint may produce) must send only its own block of 128..512 queries to
    the general kernel -- < 1 % of the workgroups at these sizes -- and the result must still be softmax attention on the
    bf16-rounded operands, for the outlier row too.  nm_bound < 0 (general kernel everywhere) and the default bound must
    agree to rounding, and an outlier KEY (which bounds every query of its pair) must fall back pair-wide."""
    rng = np.random.default_rng(5)
    d = C // heads
    qkv = (rng.standard_normal((B, N, 3 * C)) * 0.7).astype(np.float32)
    out_tok = (37, N - 5)
    for b in range(B):
        qkv[b, out_tok[b], :C] *= 400.0                                   # one outlier query token per image, every head

    def bf16(a):
        u = np.ascontiguousarray(a, np.float32).view(np.uint32)
        return ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32).view(np.float32)
    got, taken, total = ops.attention_bf16_routed(qkv, heads)
    # one routing block (= one no-max workgroup = 2 general workgroups at head_dim <= 16, 1 above) per (image, head)
    assert total > 0 and taken == B * heads * (2 if d <= 16 else 1), (taken, total)
    if N == 65536:
        assert taken < 0.01 * total, (taken, total)                       # the full-resolution level of a 256 x 256 forward
    gen, taken_g, total_g = ops.attention_bf16_routed(qkv, heads, nm_bound=-1.0)
    assert taken_g == total_g == total
    assert np.isfinite(got).all() and np.isfinite(gen).all()
    e = np.abs(got - gen)
    assert e.max() < 2e-2 and e.mean() < 1e-3, (e.max(), e.mean())
    # reference on a subset of query rows (the outlier rows, their block neighbours and a few ordinary rows), float64
    q, k, v = (qkv[..., i * C:(i + 1) * C].reshape(B, N, heads, d).transpose(0, 2, 1, 3) for i in range(3))
    kr, vr = bf16(k).astype(np.float64), bf16(v).astype(np.float64)
    for b in range(B):
        rows = np.array(sorted({out_tok[b], max(out_tok[b] - 1, 0), min(out_tok[b] + 1, N - 1), 0, N // 2, N - 1, 1000}))
        qr = bf16(q[b][:, rows] * np.float32(1.4426950408889634 / np.sqrt(d))).astype(np.float64)
        s2 = qr @ kr[b].transpose(0, 2, 1)
        p = np.exp2(s2 - s2.max(-1, keepdims=True))
        ref = ((p / p.sum(-1, keepdims=True)) @ vr[b]).transpose(1, 0, 2).reshape(len(rows), C)
        er = np.abs(got[b, rows] - ref)
        assert er.max() < 3e-2, (b, er.max())
    # an outlier key raises the bound of every query of its pair: those pairs run in the general kernel as a whole
    qkv2 = (rng.standard_normal((1, N, 3 * C)) * 0.7).astype(np.float32)
    qkv2[0, 11, C:C + d] *= 400.0                                         # head 0 only
    _, taken2, total2 = ops.attention_bf16_routed(qkv2, heads)
    assert taken2 == total2 // heads, (taken2, total2)


@pytest.mark.parametrize("B,H,W,bs,low,C", [(2, 16, 32, 8, 4, 64), (3, 32, 32, 4, 3, 64), (1, 64, 64, 8, 4, 64), (2, 8, 16, 4, 3, 64),
                                             (2, 16, 24, 8, 4, 128), (3, 16, 16, 4, 3, 128), (1, 32, 64, 8, 4, 128)])
def test_frequency_block_one_pass_kernel(ops, B, H, W, bs, low, C):
    """DCT + both gate layers + S = x + g * DCT(x) in one kernel (freq_fused.hip, the 64- and 128-channel blocks) against the three
    kernels it replaces (bit for bit: same FMA / MFMA order, same bf16 roundings) and against a float64 restatement of
    webp_inference.py:231-270 on the bf16-rounded operands."""
    from ddpm_image_restoration_amd.synth import dct_matrix
    x = rnd(B, C, H, W, seed=50)
    D = dct_matrix(bs)
    w1 = rnd(C, C, seed=51, scale=0.2)
    b1 = rnd(C, seed=52, scale=0.3)
    w2 = rnd(2 * C, C, seed=53, scale=0.3)
    w2[:C, C // 2:] = 0; w2[C:, :C // 2] = 0
    b2 = rnd(2 * C, seed=54, scale=0.3)
    boost = np.abs(rnd(B, seed=55)) + 0.5
    fused = ops.freq_gates_bf16(nhwc(x), D, w1, b1, w2, b2, boost, low, fused=True)
    three = ops.freq_gates_bf16(nhwc(x), D, w1, b1, w2, b2, boost, low, fused=False)
    ndiff = int((fused != three).sum())
    print(f"fused vs three kernels: {ndiff} of {fused.size} bf16 values differ, max |d| {np.abs(fused - three).max():.2e}")
    assert ndiff == 0

    def bf(z):
        u = np.ascontiguousarray(z, dtype=np.float32).view(np.uint32)
        return ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32).view(np.float32)
    xd = O.blockwise_transform(x, D)
    hid = O.conv2d(bf(xd), bf(w1)[:, :, None, None], b1, 0)
    hid = np.where(hid > 0, hid, 0.2 * hid).astype(np.float32)
    z = O.conv2d(bf(hid), bf(w2)[:, :, None, None], b2, 0).astype(np.float64)
    sig = 1 / (1 + np.exp(-z))
    m = O.low_mask(H, W, bs, low)[None, None]
    ref = x + xd * np.where(m > 0, sig[:, :C], sig[:, C:] * boost[:, None, None, None])
    err = np.abs(nchw(fused) - ref)
    assert err.max() < 4e-2 and err.mean() < 2e-3, (err.max(), err.mean())     # the final bf16 rounding of S (|S| up to ~4)
    # with the attention's output projection and residual formed inside (x := W_o att + b_o + x)
    att, wo, bo = rnd(B, C, H, W, seed=56), rnd(C, C, seed=57, scale=0.15), rnd(C, seed=58, scale=0.2)
    f2 = ops.freq_gates_bf16(nhwc(x), D, w1, b1, w2, b2, boost, low, fused=True, att=nhwc(att), wo=wo, bo=bo)
    t2 = ops.freq_gates_bf16(nhwc(x), D, w1, b1, w2, b2, boost, low, fused=False, att=nhwc(att), wo=wo, bo=bo)
    assert int((f2 != t2).sum()) == 0
    x2 = x + O.conv2d(bf(att), bf(wo)[:, :, None, None], bo, 0)
    assert np.abs(nchw(f2) - nchw(fused)).max() > 0.05 and np.isfinite(f2).all() and np.abs(x2).max() < 50


@pytest.mark.parametrize("B,H,W,bs,low,C", [(2, 256, 256, 8, 4, 64), (4, 128, 128, 8, 4, 128), (2, 256, 256, 4, 3, 64)])
def test_frequency_block_one_pass_kernel_with_several_workgroups_per_cu(ops, B, H, W, bs, low, C):
    """The shapes of the test above launch at most 64 tiles -- one workgroup per CU.  A v_pk_fma_f32 form of the kernel's DCT
    passed them bit for bit and produced thousands of wrong values per launch, different ones each time, once two workgroups
    shared a CU.  Here: 1024 tiles (three workgroups per CU), equality with the three-kernel path, with and
    without the attention's output projection formed inside, and the same bits on every call."""
    from ddpm_image_restoration_amd.synth import dct_matrix
    x = rnd(B, C, H, W, seed=60)
    D = dct_matrix(bs)
    w1, b1 = rnd(C, C, seed=61, scale=0.2), rnd(C, seed=62, scale=0.3)
    w2 = rnd(2 * C, C, seed=63, scale=0.3)
    w2[:C, C // 2:] = 0; w2[C:, :C // 2] = 0
    b2 = rnd(2 * C, seed=64, scale=0.3)
    boost = np.abs(rnd(B, seed=65)) + 0.5
    att, wo, bo = rnd(B, C, H, W, seed=66), rnd(C, C, seed=67, scale=0.15), rnd(C, seed=68, scale=0.2)
    three = ops.freq_gates_bf16(nhwc(x), D, w1, b1, w2, b2, boost, low, fused=False)
    three_op = ops.freq_gates_bf16(nhwc(x), D, w1, b1, w2, b2, boost, low, fused=False, att=nhwc(att), wo=wo, bo=bo)
    for rep in range(3):
        f = ops.freq_gates_bf16(nhwc(x), D, w1, b1, w2, b2, boost, low, fused=True)
        f_op = ops.freq_gates_bf16(nhwc(x), D, w1, b1, w2, b2, boost, low, fused=True, att=nhwc(att), wo=wo, bo=bo)
        assert int((f != three).sum()) == 0, (rep, int((f != three).sum()), float(np.abs(f - three).max()))
        assert int((f_op != three_op).sum()) == 0, (rep, int((f_op != three_op).sum()), float(np.abs(f_op - three_op).max()))


@pytest.mark.parametrize("B,H,W,Cin,Cout,k", [(2, 128, 128, 64, 64, 3), (3, 192, 224, 64, 64, 3), (1, 160, 224, 32, 32, 3), (2, 32, 32, 3, 64, 3),
                                              (2, 16, 16, 64, 32, 1), (4, 96, 96, 128, 64, 3), (3, 128, 96, 64, 32, 3), (9, 128, 128, 64, 64, 3),
                                              (2, 64, 64, 64, 64, 1), (5, 7, 3, 36, 40, 3), (3, 200, 232, 64, 64, 3)])
def test_conv_bf16_register_direct_epilogue_equals_the_lds_epilogue(ops, B, H, W, Cin, Cout, k):
    """The 64-column GEMM kernels (gemm16_kernel<1|2>, conv3x3_strip_kernel: full-resolution convs, tile walks, the tap-packed
    image conv) store their accumulators straight from registers for the plain option sets (gemm_epilogue.h:
    gemm16_epilogue_direct; bias, time bias, fp32 / bf16 outputs).  Same arithmetic in the same order as the LDS-transposed
    epilogue: bit-identical results.  The last two shapes are NOT eligible (ragged column tile; H*W % 32 != 0) and must simply
    keep working; an epilogue with factors / residual stays on the LDS path under either setting."""
    x = rnd(B, Cin, H, W, seed=61)
    w = rnd(Cout, Cin, k, k, seed=62, scale=1.0 / np.sqrt(Cin * k * k))
    b = rnd(Cout, seed=63)
    tb, res = rnd(B, Cout, seed=64), rnd(B, Cout, H, W, seed=65)
    try:
        for kw in (dict(), dict(tb=tb), dict(tb=tb, res=nhwc(res))):
            ops.set_epilogue_direct(1)
            a = ops.conv(nhwc(x), w, b, bf16="dma", **kw)
            ops.set_epilogue_direct(0)
            c = ops.conv(nhwc(x), w, b, bf16="dma", **kw)
            assert np.array_equal(a, c), (list(kw), float(np.abs(a - c).max()))
    finally:
        ops.set_epilogue_direct(1)
    def bf(z):
        u = np.ascontiguousarray(z, dtype=np.float32).view(np.uint32)
        return ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32).view(np.float32)
    close(nchw(a)[:1], O.conv2d(bf(x[:1]), bf(w), b, k // 2) + tb[:1, :, None, None] + res[:1], 3e-5)


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,G", [(2, 128, 128, 64, 64, 3, 8), (2, 128, 128, 8, 32, 3, 8), (4, 64, 128, 64, 64, 3, 1), (2, 128, 128, 32, 64, 3, 4),
                                                (8, 64, 64, 16, 32, 1, 2), (2, 128, 128, 64, 64, 3, 2)])
def test_conv_bf16_register_direct_epilogue_groupnorm_statistics(ops, B, H, W, Cin, Cout, k, G):
    """the direct epilogue's per-(sample, group) sums (one writer per 32-row block and group; groups of 4 ... 64 channels, a
    64-channel group spanning both column tiles) are the statistics of the tensor it stored and equal the LDS epilogue's up to
    the summation order."""
    x = rnd(B, Cin, H, W, seed=70)
    w = rnd(Cout, Cin, k, k, seed=71, scale=1.0 / np.sqrt(Cin * k * k))
    b = rnd(Cout, seed=72) + 0.5
    try:
        ops.set_epilogue_direct(1)
        out, sums = ops.conv_gnstats_bf16(nhwc(x), w, b, G)
        ops.set_epilogue_direct(0)
        out0, sums0 = ops.conv_gnstats_bf16(nhwc(x), w, b, G)
    finally:
        ops.set_epilogue_direct(1)
    assert np.array_equal(out, out0)
    o = out.astype(np.float64).reshape(B, H * W, G, Cout // G)
    want = np.stack([o.sum(axis=(1, 3)), (o * o).sum(axis=(1, 3))], axis=-1)
    assert np.allclose(sums, want, rtol=2e-5, atol=1e-2), np.abs(sums - want).max()
    assert np.allclose(sums, sums0, rtol=2e-5, atol=1e-2)
