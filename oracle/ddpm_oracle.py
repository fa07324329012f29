"""CPU oracle: a numpy restatement of the reference's restoration path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's ``cpu_baseline`` leg may import this module; the product
(ddpm_image_restoration_amd/) never does and fails loudly when its HIP library is missing.

What it restates (reference file:line next to each function):
  * the three U-Nets' forward  (webp_inference.py:135-399, svd.ipynb c0:102-366, avif_inference.py:102-385)
  * the codec round trips      (webp_inference.py:506-528, svd.ipynb c0:45-69, avif_inference.py:64-98)
  * phase_consistency          (webp_inference.py:531-550)
  * the DDRM sampling loops    (webp_inference.py:557-602, svd.ipynb c0:395-440, avif_inference.py:414-459)
  * the eps-predict DDPM loop  (experiments/code/conv.ipynb c1:113-125)

Pinning: tests/test_oracle_golden.py checks every function here against tests/golden/*.npz, which were
produced by executing the reference's own source in the build container (tests/golden/make_golden.py; the
reference publishes no fixtures of its own -- SURVEY.md section 4).  The codecs are third-party libraries
(Pillow -> libjpeg-turbo / libwebp / libavif) that the reference does not pin; the goldens pin them to the
versions of this image.

Layout is NCHW float32 like the reference; arithmetic is float32 (numpy BLAS for the contractions).
"""
from __future__ import annotations

import io
import math

import numpy as np
from scipy.special import erf as _erf

F32 = np.float32

# ------------------------------------------------------------------------------------------------ constants
ARCH_WEBP, ARCH_JPEG, ARCH_AVIF = 0, 1, 2

BLOCKS = [  # name, in_c, out_c   (webp_inference.py:337-356)
    ("down1", 3, 64), ("down2", 64, 128), ("down3", 128, 256), ("down4", 256, 512), ("down5", 512, 512),
    ("bottleneck.0", 512, 1024), ("bottleneck.1", 1024, 1024), ("bottleneck.2", 1024, 512),
    ("up1", 1024, 512), ("up2", 1024, 256), ("up3", 512, 128), ("up4", 256, 64), ("up5", 128, 64),
]

HEADS = {ARCH_WEBP: 4, ARCH_JPEG: 4, ARCH_AVIF: 8}
DCT_BLOCK = {ARCH_WEBP: 4, ARCH_JPEG: 8}
LOW_SIZE = {ARCH_WEBP: 3, ARCH_JPEG: 4}
BOOST = {ARCH_WEBP: (0.15, 1.9), ARCH_JPEG: (0.2, 2.0)}
# sampler constants: noise coeff, phase-fix quality threshold, phase-fix period, alpha
SAMPLER = {ARCH_WEBP: (0.2, 15, 5, 0.7), ARCH_JPEG: (0.2, 20, 5, 0.7), ARCH_AVIF: (0.15, 30, 3, 0.8)}


# ------------------------------------------------------------------------------------------------ primitives
def _groups(c):  # webp_inference.py:277-279
    g = min(8, c)
    while c % g != 0 and g > 1:
        g -= 1
    return g


def group_norm(x, g, w, b, eps=1e-5):
    """nn.GroupNorm: biased variance over (C/g, H, W) per sample and group, then per-channel affine."""
    n, c, h, wd = x.shape
    xg = x.reshape(n, g, -1).astype(np.float64)
    mean = xg.mean(axis=2, keepdims=True)
    var = xg.var(axis=2, keepdims=True)
    y = ((xg - mean) / np.sqrt(var + eps)).astype(F32).reshape(n, c, h, wd)
    return y * w.reshape(1, c, 1, 1) + b.reshape(1, c, 1, 1)


def conv2d(x, w, b, pad):
    """nn.Conv2d stride 1, zero padding `pad`, weight OIHW."""
    n, c, h, wd = x.shape
    o, _, kh, kw = w.shape
    if kh == 1 and kw == 1:
        y = np.einsum("oc,nchw->nohw", w[:, :, 0, 0], x, optimize=True)
    else:
        xp = np.pad(x, ((0, 0), (0, 0), (pad, pad), (pad, pad)))
        win = np.lib.stride_tricks.sliding_window_view(xp, (kh, kw), axis=(2, 3))  # n c h w kh kw
        cols = win.transpose(0, 2, 3, 1, 4, 5).reshape(n * h * wd, c * kh * kw)
        y = (cols @ w.reshape(o, -1).T).reshape(n, h, wd, o).transpose(0, 3, 1, 2)
    return (y + b.reshape(1, o, 1, 1)).astype(F32)


def linear(x, w, b):
    return (x @ w.T + b).astype(F32)


def gelu(x):  # F.gelu default = exact erf form (webp_inference.py:312)
    return (x * F32(0.5) * (F32(1.0) + _erf(x / F32(math.sqrt(2.0))).astype(F32))).astype(F32)


def silu(x):
    return (x / (F32(1.0) + np.exp(-x))).astype(F32)


def sigmoid(x):
    return (F32(1.0) / (F32(1.0) + np.exp(-x))).astype(F32)


def max_pool2(x):  # nn.MaxPool2d(2): stride 2, floor
    n, c, h, w = x.shape
    x = x[:, :, : h // 2 * 2, : w // 2 * 2]
    return x.reshape(n, c, h // 2, 2, w // 2, 2).max(axis=(3, 5))


def _lin_idx(out_size, in_size, scale):
    """torch area_pixel_compute_source_index (align_corners=False) + linear weights."""
    dst = np.arange(out_size, dtype=F32)
    src = np.maximum(F32(scale) * (dst + F32(0.5)) - F32(0.5), F32(0.0)).astype(F32)
    i0 = np.minimum(np.floor(src).astype(np.int64), in_size - 1)
    i1 = np.minimum(i0 + 1, in_size - 1)
    l1 = (src - i0.astype(F32)).astype(F32)
    return i0, i1, (F32(1.0) - l1).astype(F32), l1


def bilinear(x, out_h, out_w, scale_h=None, scale_w=None):
    """F.interpolate(mode='bilinear', align_corners=False).  With scale_factor=s torch uses 1/s as the
    coordinate scale; with size= it uses in/out."""
    n, c, h, w = x.shape
    sh = scale_h if scale_h is not None else h / out_h
    sw = scale_w if scale_w is not None else w / out_w
    y0, y1, wy0, wy1 = _lin_idx(out_h, h, sh)
    x0, x1, wx0, wx1 = _lin_idx(out_w, w, sw)
    top = x[:, :, y0][:, :, :, x0] * wx0 + x[:, :, y0][:, :, :, x1] * wx1
    bot = x[:, :, y1][:, :, :, x0] * wx0 + x[:, :, y1][:, :, :, x1] * wx1
    return (top * wy0[:, None] + bot * wy1[:, None]).astype(F32)


def upsample2(x):  # F.interpolate(scale_factor=2, bilinear, align_corners=False) webp_inference.py:389
    return bilinear(x, x.shape[2] * 2, x.shape[3] * 2, 0.5, 0.5)


def adaptive_avg_pool(x, s):
    """nn.AdaptiveAvgPool2d(s): window [floor(i*H/s), ceil((i+1)*H/s))."""
    n, c, h, w = x.shape
    out = np.empty((n, c, s, s), dtype=F32)
    for i in range(s):
        ys, ye = (i * h) // s, -((-(i + 1) * h) // s)
        for j in range(s):
            xs, xe = (j * w) // s, -((-(j + 1) * w) // s)
            out[:, :, i, j] = x[:, :, ys:ye, xs:xe].mean(axis=(2, 3))
    return out


def _attn_chunk(q, k, v, scale):
    sc = (q * scale) @ k.T
    sc -= sc.max(axis=1, keepdims=True)
    np.exp(sc, out=sc)
    sc /= sc.sum(axis=1, keepdims=True)
    return sc @ v


def mha(tokens, in_w, in_b, out_w, out_b, heads):
    """nn.MultiheadAttention(batch_first=True) self-attention, eval, no mask (webp_inference.py:317-321).
    Large token counts are processed in query chunks on a thread pool (numpy releases the GIL), so the
    N x N scores are never held at once and all host cores are used when this is timed as the CPU baseline."""
    b, n, c = tokens.shape
    d = c // heads
    qkv = linear(tokens, in_w, in_b)
    q, k, v = qkv[..., :c], qkv[..., c:2 * c], qkv[..., 2 * c:]
    q = np.ascontiguousarray(q.reshape(b, n, heads, d).transpose(0, 2, 1, 3))
    k = np.ascontiguousarray(k.reshape(b, n, heads, d).transpose(0, 2, 1, 3))
    v = np.ascontiguousarray(v.reshape(b, n, heads, d).transpose(0, 2, 1, 3))
    out = np.empty((b, heads, n, d), dtype=F32)
    scale = F32(1.0 / math.sqrt(d))
    chunk = max(1, min(n, (1 << 22) // max(n, 1)))
    jobs = [(bi, hi, s) for bi in range(b) for hi in range(heads) for s in range(0, n, chunk)]

    def run(job):
        bi, hi, s = job
        out[bi, hi, s:s + chunk] = _attn_chunk(q[bi, hi, s:s + chunk], k[bi, hi], v[bi, hi], scale)

    if len(jobs) <= 2 or n < 4096:
        for j in jobs:
            run(j)
    else:
        import os
        from concurrent.futures import ThreadPoolExecutor
        try:
            from threadpoolctl import threadpool_limits
        except ImportError:
            threadpool_limits = None
        # at most 32 worker threads: OpenBLAS keeps a fixed table of per-caller buffers (128 in the numpy wheel) and aborts
        # the process when more threads than that call into it -- seen with 256 threads on the MI355X host
        nthreads = oracle_threads()
        with ThreadPoolExecutor(nthreads) as ex:
            if threadpool_limits is not None:
                with threadpool_limits(limits=1):
                    list(ex.map(run, jobs))
            else:
                list(ex.map(run, jobs))
    out = out.transpose(0, 2, 1, 3).reshape(b, n, c)
    return linear(out, out_w, out_b)


def oracle_threads():
    """Threads the oracle's attention uses (also what bench.py reports as `cpu_baseline.cores`)."""
    import os
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(32, n))


def blockwise_transform(x, mats):
    """Zero-pad to a multiple of the block, per block Y = T X T^T with coefficients left in place, crop.
    `mats` is one (b,b) matrix (DCTLayer, webp_inference.py:161-192) or (C,b,b) per-channel matrices
    (AVIFAdaptiveTransform, avif_inference.py:140-177)."""
    n, c, h, w = x.shape
    bs = mats.shape[-1]
    hp, wp = (bs - h % bs) % bs, (bs - w % bs) % bs
    xp = np.pad(x, ((0, 0), (0, 0), (0, hp), (0, wp)))
    H, W = h + hp, w + wp
    blk = xp.reshape(n, c, H // bs, bs, W // bs, bs).transpose(0, 1, 2, 4, 3, 5)  # n c by bx i j
    if mats.ndim == 2:
        y = np.einsum("ui,ncyxij,vj->ncyxuv", mats, blk, mats, optimize=True)
    else:
        y = np.einsum("cui,ncyxij,cvj->ncyxuv", mats, blk, mats, optimize=True)
    y = y.transpose(0, 1, 2, 4, 3, 5).reshape(n, c, H, W)
    return np.ascontiguousarray(y[:, :, :h, :w]).astype(F32)


def low_mask(h, w, bs, low):
    """Static in-block low-frequency mask of webp_inference.py:241-252 (incl. the partial-block rule)."""
    m = np.zeros((h, w), dtype=bool)
    for i in range(0, h, bs):
        ie = min(i + bs, h)
        for j in range(0, w, bs):
            je = min(j + bs, w)
            ls = max(1, min(low, min(ie - i, je - j)))
            m[i:i + ls, j:j + ls] = True
    return m


# ------------------------------------------------------------------------------------------------ model
class Model:
    """Forward of {WebP,JPEG,AVIF}DiffusionModel on a name->ndarray state dict."""

    def __init__(self, arch, state, attn_min_level=0):
        self.arch = arch
        self.p = {k: np.asarray(v, dtype=F32) for k, v in state.items()}
        self.attn_min_level = attn_min_level   # 0 = attention in every block (the reference)
        self.taps = None                       # optional dict collecting intermediates

    # -- webp_inference.py:145-151
    def time_embedding(self, t):
        half = 128
        emb = math.log(10000) / (half - 1)
        freq = np.exp(np.arange(half, dtype=F32) * F32(-emb)).astype(F32)
        e = t.astype(F32)[:, None] * freq[None, :]
        e = np.concatenate([np.sin(e), np.cos(e)], axis=-1).astype(F32)
        h = silu(linear(e, self.p["time_embed.proj.0.weight"], self.p["time_embed.proj.0.bias"]))
        return linear(h, self.p["time_embed.proj.2.weight"], self.p["time_embed.proj.2.bias"])

    def _gate(self, pfx, z, act, i0=0, i1=2):
        """sigmoid(conv1x1_{i1}(act(conv1x1_{i0}(z)))) -- the Sequential gates of the frequency blocks."""
        h = conv2d(z, self.p[f"{pfx}.{i0}.weight"], self.p[f"{pfx}.{i0}.bias"], 0)
        h = np.where(h > 0, h, h * F32(0.2)).astype(F32) if act == "lrelu" else np.maximum(h, 0).astype(F32)
        return sigmoid(conv2d(h, self.p[f"{pfx}.{i1}.weight"], self.p[f"{pfx}.{i1}.bias"], 0))

    # -- webp_inference.py:231-270 / svd.ipynb c0:198-237
    def freq_block_dct(self, pfx, x, cl):
        bs, low = DCT_BLOCK[self.arch], LOW_SIZE[self.arch]
        xd = blockwise_transform(x, self.p[pfx + ".dct.dct_matrix"])
        m = low_mask(x.shape[2], x.shape[3], bs, low)[None, None]
        lowf = np.where(m, xd, F32(0)).astype(F32)
        highf = np.where(m, F32(0), xd).astype(F32)
        ga = self._gate(pfx + ".low_freq_attn", lowf, "lrelu")
        gb = self._gate(pfx + ".high_freq_attn", highf, "lrelu")
        lo, hi = BOOST[self.arch]
        boost = np.clip(F32(1.0) - cl.astype(F32), F32(lo), F32(hi)).reshape(-1, 1, 1, 1)
        gb = gb * boost
        comb = ga * lowf + gb * highf
        return conv2d(x + comb, self.p[pfx + ".conv_out.weight"], self.p[pfx + ".conv_out.bias"], 1)

    # -- avif_inference.py:140-182
    def avif_transform(self, pfx, x):
        tr = blockwise_transform(x, self.p[pfx + ".transform_weights"])
        h = np.maximum(conv2d(tr, self.p[pfx + ".quantization.0.weight"], self.p[pfx + ".quantization.0.bias"], 0), 0)
        q = sigmoid(conv2d(h.astype(F32), self.p[pfx + ".quantization.2.weight"], self.p[pfx + ".quantization.2.bias"], 0))
        return tr * q

    # -- avif_inference.py:222-256
    def freq_block_avif(self, pfx, x, cl):
        xt = self.avif_transform(pfx + ".adaptive_transform", x)
        acc = np.zeros_like(x)
        for si, s in enumerate((1, 2, 4, 8)):
            a = adaptive_avg_pool(x, s)
            a = self._gate(f"{pfx}.multi_scale_attn.{si}", a, "relu", 1, 3)
            if a.shape != x.shape:
                a = bilinear(a, x.shape[2], x.shape[3])
            acc = acc + a
        avg = acc / F32(4.0)
        color = self._gate(pfx + ".color_consistency", x, "relu")
        e = np.maximum(conv2d(x, self.p[pfx + ".edge_preserve.0.weight"], self.p[pfx + ".edge_preserve.0.bias"], 1), 0)
        edge = sigmoid(conv2d(e.astype(F32), self.p[pfx + ".edge_preserve.2.weight"], self.p[pfx + ".edge_preserve.2.bias"], 1))
        clf = cl.astype(F32).reshape(-1, 1, 1, 1)
        color = color * np.clip(F32(0.5) + F32(0.5) * (F32(1.0) - clf), F32(0.3), F32(1.5))
        edge = edge * np.clip(F32(0.7) + F32(0.3) * (F32(1.0) - clf), F32(0.5), F32(1.3))
        enh = xt * avg * color * edge
        return conv2d((x + enh).astype(F32), self.p[pfx + ".conv_out.weight"], self.p[pfx + ".conv_out.bias"], 1)

    # -- webp_inference.py:303-327
    def block(self, name, x, temb, cl, level):
        p = self.p
        cin, cout = x.shape[1], p[name + ".conv1.weight"].shape[0]
        h = group_norm(x, _groups(cin), p[name + ".norm1.weight"], p[name + ".norm1.bias"])
        h = conv2d(h, p[name + ".conv1.weight"], p[name + ".conv1.bias"], 1)
        h = h + linear(temb, p[name + ".time_proj.weight"], p[name + ".time_proj.bias"])[:, :, None, None]
        h = group_norm(h, _groups(cout), p[name + ".norm2.weight"], p[name + ".norm2.bias"])
        h = gelu(h)
        h = conv2d(h, p[name + ".conv2.weight"], p[name + ".conv2.bias"], 1)
        self._tap(name + ".conv2", h)
        if level >= self.attn_min_level:
            b, c, hh, ww = h.shape
            tok = h.reshape(b, c, hh * ww).transpose(0, 2, 1)
            a = mha(tok, p[name + ".attn.in_proj_weight"], p[name + ".attn.in_proj_bias"],
                    p[name + ".attn.out_proj.weight"], p[name + ".attn.out_proj.bias"], HEADS[self.arch])
            h = h + a.transpose(0, 2, 1).reshape(b, c, hh, ww)
        self._tap(name + ".attn", h)
        if self.arch == ARCH_AVIF:
            h = self.freq_block_avif(name + ".freq_guide", h, cl)
        else:
            h = self.freq_block_dct(name + ".freq_guide", h, cl)
        sc = x if cin == cout else conv2d(x, p[name + ".shortcut.weight"], p[name + ".shortcut.bias"], 0)
        out = (sc + h).astype(F32)
        self._tap(name, out)
        return out

    def _tap(self, key, val):
        if self.taps is not None:
            self.taps[key] = val.copy()

    # -- webp_inference.py:369-399
    def forward(self, x, t, cl=None):
        x = np.asarray(x, dtype=F32)
        t = np.asarray(t, dtype=F32)
        cl = t.copy() if cl is None else np.asarray(cl, dtype=F32)
        temb = self.time_embedding(t)
        self._tap("t_emb", temb)
        blk = lambda nm, z, lvl: self.block(nm, z, temb, cl, lvl)
        d1 = blk("down1", x, 0)
        d2 = blk("down2", max_pool2(d1), 1)
        d3 = blk("down3", max_pool2(d2), 2)
        d4 = blk("down4", max_pool2(d3), 3)
        d5 = blk("down5", max_pool2(d4), 4)
        b = blk("bottleneck.0", max_pool2(d5), 5)
        b = blk("bottleneck.1", b, 5)
        b = blk("bottleneck.2", b, 5)
        cat = lambda u, s: np.concatenate([upsample2(u), s], axis=1)
        u1 = blk("up1", cat(b, d5), 4)
        u2 = blk("up2", cat(u1, d4), 3)
        u3 = blk("up3", cat(u2, d3), 2)
        u4 = blk("up4", cat(u3, d2), 1)
        u5 = blk("up5", cat(u4, d1), 0)
        if self.arch == ARCH_AVIF:
            comb = u5 + F32(0.15) * self.avif_transform("avif_layer", u5)
        else:
            comb = u5 + F32(0.1) * blockwise_transform(u5, self.p["dct_layer.dct_matrix"])
        h = group_norm(comb.astype(F32), 8, self.p["out_conv.0.weight"], self.p["out_conv.0.bias"])
        h = silu(h)
        return np.tanh(conv2d(h, self.p["out_conv.2.weight"], self.p["out_conv.2.bias"], 1)).astype(F32)

    __call__ = forward


# ------------------------------------------------------------------------------------------------ codecs
def to_u8(x):
    """(x*127.5+127.5).clamp(0,255).to(uint8): two fp32 roundings, then truncation (webp_inference.py:509)."""
    v = (np.asarray(x, dtype=F32) * F32(127.5)).astype(F32) + F32(127.5)
    return np.clip(v, F32(0), F32(255)).astype(np.uint8)


def from_u8(u):
    """ToTensor then .sub(0.5).mul(2.0) (webp_inference.py:524-528)."""
    return ((u.astype(F32) / F32(255.0)).astype(F32) - F32(0.5)) * F32(2.0)


def codec_u8(u8_nchw, quality, kind):
    """PIL encode->decode of every image of a uint8 NCHW batch.  kind: 'webp' | 'jpeg' | 'avif'."""
    from PIL import Image
    out = np.empty_like(u8_nchw)
    for i, img in enumerate(u8_nchw):
        pil = Image.fromarray(np.ascontiguousarray(img.transpose(1, 2, 0)))
        buf = io.BytesIO()
        if kind == "webp":
            q = max(0, min(100, int(quality)))
            pil.save(buf, format="WEBP", quality=q)
        elif kind == "jpeg":
            q = max(1, min(100, int(quality)))
            pil.save(buf, format="JPEG", quality=q, subsampling="4:4:4" if q > 30 else "4:2:0")
        elif kind == "avif":
            q = max(1, min(100, int(quality)))
            try:
                pil.save(buf, format="AVIF", quality=q)
            except Exception:  # avif_inference.py:88-95: silent JPEG fallback
                buf = io.BytesIO()
                pil.save(buf, format="JPEG", quality=q)
        else:
            raise ValueError(kind)
        buf.seek(0)
        out[i] = np.asarray(Image.open(buf).convert("RGB"), dtype=np.uint8).transpose(2, 0, 1)
    return out


def compress(x, quality, kind):
    """webp_compress / jpeg_compress / avif_compress: [-1,1] float NCHW -> decoded [-1,1] float NCHW."""
    return from_u8(codec_u8(to_u8(x), quality, kind)).astype(F32)


KIND = {ARCH_WEBP: "webp", ARCH_JPEG: "jpeg", ARCH_AVIF: "avif"}


# ------------------------------------------------------------------------------------------------ sampler
def phase_consistency(x, ref, alpha):
    """webp_inference.py:531-550."""
    xf = np.fft.fft2(x.astype(F32)).astype(np.complex64)
    rf = np.fft.fft2(ref.astype(F32)).astype(np.complex64)
    mag = np.abs(xf).astype(F32)
    ph = np.angle(rf).astype(F32)
    adj = np.fft.ifft2((mag * np.cos(ph) + 1j * mag * np.sin(ph)).astype(np.complex64)).real.astype(F32)
    return (F32(alpha) * x + F32(1 - alpha) * adj).astype(F32)


def ddrm_sample(model, x_t, quality, steps=100, eta=0.85, eta_b=1.0, noise=None, codec=None, trace=None):
    """DDRM{WebP,JPEG,AVIF}Sampler.sample (webp_inference.py:557-602).

    noise[i] replaces torch.randn_like at loop index i; codec(x, quality) replaces the PIL round trip
    (default: this module's compress for the model's codec); trace collects (i, x_theta, x_t)."""
    coeff, phase_q, every, alpha = SAMPLER[model.arch]
    kind = KIND[model.arch]
    codec = codec or (lambda z, q: compress(z, q, kind))
    x_t = np.asarray(x_t, dtype=F32).copy()
    y = x_t.copy()
    b = x_t.shape[0]
    for i in range(steps - 1, -1, -1):
        t = (np.full((b,), i, dtype=F32) / F32(steps)).astype(F32)
        x_theta = model(x_t, t, t.copy())
        dec = codec(x_theta, quality)
        x_prime = (x_theta - dec) + y
        if i > 0:
            eps = noise[i] if noise is not None else np.random.standard_normal(x_t.shape).astype(F32)
            rnd = eps * (t * F32(coeff)).reshape(-1, 1, 1, 1)
            x_t = (F32(eta_b) * x_prime + F32(1 - eta_b) * x_theta) + F32(eta) * rnd
            if quality < phase_q and i % every == 0:
                x_t = phase_consistency(x_t, y, alpha)
        else:
            x_t = x_prime
        x_t = x_t.astype(F32)
        if trace is not None:
            trace.append((i, x_theta.copy(), x_t.copy()))
    return x_t


# ------------------------------------------------------------------------------------------------ eps-DDPM
def ddpm_betas(T=100):
    return np.linspace(1e-4, 0.02, T, dtype=np.float64).astype(F32)   # torch.linspace fp32


def p_sample_loop(eps_model, x, T, betas):
    """ddpm_inference (experiments/code/conv.ipynb c1:117-125): deterministic posterior-mean loop."""
    alphas = (F32(1.0) - betas).astype(F32)
    x = np.asarray(x, dtype=F32).copy()
    for t in range(T, 0, -1):
        eps = eps_model(x, np.full((x.shape[0],), t, dtype=F32))
        a = alphas[t - 1]
        abar = np.prod(alphas[:t], dtype=F32)
        x = (F32(1.0) / np.sqrt(a)) * (x - ((F32(1.0) - a) / np.sqrt(F32(1.0) - abar)) * eps)
        x = x.astype(F32)
    return x


class EpsUnet:
    """UNetDenoiser.forward (experiments/code/conv.ipynb c0:55-101): eps = net(x, t)."""

    def __init__(self, state):
        self.p = {k: np.asarray(v, dtype=F32) for k, v in state.items()}

    def _dc(self, name, x):
        x = np.maximum(conv2d(x, self.p[name + ".conv.0.weight"], self.p[name + ".conv.0.bias"], 1), 0).astype(F32)
        return np.maximum(conv2d(x, self.p[name + ".conv.2.weight"], self.p[name + ".conv.2.bias"], 1), 0).astype(F32)

    def _up(self, name, x):
        """nn.ConvTranspose2d(k=2, s=2): out[n,o,2y+i,2x+j] = sum_c x[n,c,y,x] w[c,o,i,j] + b[o]."""
        w, b = self.p[name + ".weight"], self.p[name + ".bias"]
        n, c, h, wd = x.shape
        y = np.einsum("nchw,coij->nohiwj", x, w, optimize=True).reshape(n, w.shape[1], 2 * h, 2 * wd)
        return (y + b.reshape(1, -1, 1, 1)).astype(F32)

    def __call__(self, x, t):
        x = np.asarray(x, dtype=F32)
        tv = np.asarray(t, dtype=F32).reshape(-1, 1)
        h = np.maximum(tv @ self.p["time_embed.0.weight"].T + self.p["time_embed.0.bias"], 0).astype(F32)
        temb = (h @ self.p["time_embed.2.weight"].T + self.p["time_embed.2.bias"]).astype(F32)[:, :, None, None]
        x1 = self._dc("encoder1", x)
        x2 = self._dc("encoder2", max_pool2(x1))
        x3 = self._dc("encoder3", max_pool2(x2))
        z = self._dc("bottleneck", x3) + temb
        z = self._up("up3", z)
        z = self._dc("decoder3", np.concatenate([bilinear(x3, z.shape[2], z.shape[3]), z], axis=1))
        z = self._up("up2", z)
        z = self._dc("decoder2", np.concatenate([bilinear(x2, z.shape[2], z.shape[3]), z], axis=1))
        z = self._up("up1", z)
        z = self._dc("decoder1", np.concatenate([bilinear(x1, z.shape[2], z.shape[3]), z], axis=1))
        z = conv2d(z, self.p["out_conv.weight"], self.p["out_conv.bias"], 0)
        return np.tanh(bilinear(z, x.shape[2], x.shape[3])).astype(F32)   # reference hard-codes size=(32, 32)


def psnr(a, b):
    """webp_inference.py:691-697: on clamp(x*0.5+0.5, 0, 1); -10 log10(mse + 1e-8)."""
    a01 = np.clip(a * 0.5 + 0.5, 0, 1)
    b01 = np.clip(b * 0.5 + 0.5, 0, 1)
    return float(-10.0 * np.log10(np.mean((a01 - b01) ** 2, dtype=np.float64) + 1e-8))
