"""State-dict layout of the three restoration U-Nets (names, shapes) and their per-codec constants.

The names/shapes mirror what ``torch.load(...)['model_state_dict']`` holds for the reference models
(reference: webp_inference.py:330-367 WebPDiffusionModel, avif_inference.py:316-353 AVIFDiffusionModel,
svd.ipynb c0:297-334 JPEGDiffusionModel), so a checkpoint trained with the reference loads by name.
``tests/golden/make_golden.py`` asserts this table against the reference's own ``state_dict()``.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass

ARCH_WEBP, ARCH_JPEG, ARCH_AVIF = 0, 1, 2
ARCH_NAMES = {"webp": ARCH_WEBP, "jpeg": ARCH_JPEG, "avif": ARCH_AVIF}

TIME_DIM = 256

# (name, in_c, out_c, level)   level = number of 2x2 pools above the block (spatial = R / 2**level)
BLOCKS = [
    ("down1", 3, 64, 0),
    ("down2", 64, 128, 1),
    ("down3", 128, 256, 2),
    ("down4", 256, 512, 3),
    ("down5", 512, 512, 4),
    ("bottleneck.0", 512, 1024, 5),
    ("bottleneck.1", 1024, 1024, 5),
    ("bottleneck.2", 1024, 512, 5),
    ("up1", 1024, 512, 4),
    ("up2", 1024, 256, 3),
    ("up3", 512, 128, 2),
    ("up4", 256, 64, 1),
    ("up5", 128, 64, 0),
]


@dataclass(frozen=True)
class CodecConsts:
    """Per-codec constants of model + sampler (reference lines in the field comments)."""
    arch: int
    heads: int            # webp_inference.py:295 (4) / avif_inference.py:281 (8)
    dct_block: int        # webp_inference.py:156 (4) / svd.ipynb c0:123 (8) / avif_inference.py:123 (8)
    low_size: int         # webp_inference.py:247 (3) / svd.ipynb c0:214 (4); unused for AVIF
    boost_lo: float       # webp_inference.py:263 clamp(1-cl, 0.15, 1.9) / svd.ipynb c0:230 (0.2, 2.0)
    boost_hi: float
    head_mix: float       # webp_inference.py:397 (0.1) / avif_inference.py:383 (0.15)
    noise_coeff: float    # webp_inference.py:588 (0.2) / avif_inference.py:445 (0.15)
    phase_q: int          # webp_inference.py:595 (15) / svd.ipynb c0:433 (20) / avif_inference.py:452 (30)
    phase_every: int      # 5 / 5 / 3
    phase_alpha: float    # 0.7 / 0.7 / 0.8
    q_min: int            # quality clamp low end: webp 0 (webp_inference.py:518), jpeg/avif 1
    steps_lo: int         # harness steps clamp: webp_inference.py:678-679 (20,80); avif_inference.py:534-535 (15,75)
    steps_hi: int


CONSTS = {
    ARCH_WEBP: CodecConsts(ARCH_WEBP, 4, 4, 3, 0.15, 1.9, 0.1, 0.2, 15, 5, 0.7, 0, 20, 80),
    ARCH_JPEG: CodecConsts(ARCH_JPEG, 4, 8, 4, 0.2, 2.0, 0.1, 0.2, 20, 5, 0.7, 1, 20, 80),
    ARCH_AVIF: CodecConsts(ARCH_AVIF, 8, 8, 0, 0.0, 0.0, 0.15, 0.15, 30, 3, 0.8, 1, 15, 75),
}


def gn_groups(c: int) -> int:
    """webp_inference.py:277-279: min(8, c) decremented until it divides c."""
    g = min(8, c)
    while c % g != 0 and g > 1:
        g -= 1
    return g


def _conv(d, name, cout, cin, k):
    d[name + ".weight"] = (cout, cin, k, k)
    d[name + ".bias"] = (cout,)


def _lin(d, name, cout, cin):
    d[name + ".weight"] = (cout, cin)
    d[name + ".bias"] = (cout,)


def _avif_transform(d, p, c):
    d[p + ".transform_weights"] = (c, 8, 8)
    d[p + ".inverse_weights"] = (c, 8, 8)      # a parameter the forward never reads (avif_inference.py:130)
    _conv(d, p + ".quantization.0", c, c, 1)
    _conv(d, p + ".quantization.2", c, c, 1)


def state_spec(arch: int) -> "OrderedDict[str, tuple]":
    """Ordered name -> shape of every state_dict entry (parameters and buffers) of the reference model."""
    k = CONSTS[arch]
    d: "OrderedDict[str, tuple]" = OrderedDict()
    _lin(d, "time_embed.proj.0", TIME_DIM * 4, TIME_DIM)
    _lin(d, "time_embed.proj.2", TIME_DIM, TIME_DIM * 4)
    for name, cin, cout, _lvl in BLOCKS:
        d[f"{name}.norm1.weight"] = (cin,)
        d[f"{name}.norm1.bias"] = (cin,)
        _conv(d, f"{name}.conv1", cout, cin, 3)
        _lin(d, f"{name}.time_proj", cout, TIME_DIM)
        d[f"{name}.norm2.weight"] = (cout,)
        d[f"{name}.norm2.bias"] = (cout,)
        _conv(d, f"{name}.conv2", cout, cout, 3)
        d[f"{name}.attn.in_proj_weight"] = (3 * cout, cout)
        d[f"{name}.attn.in_proj_bias"] = (3 * cout,)
        _lin(d, f"{name}.attn.out_proj", cout, cout)
        fg = f"{name}.freq_guide"
        if arch == ARCH_AVIF:
            _avif_transform(d, fg + ".adaptive_transform", cout)
            for s in range(4):
                _conv(d, f"{fg}.multi_scale_attn.{s}.1", cout // 4, cout, 1)
                _conv(d, f"{fg}.multi_scale_attn.{s}.3", cout, cout // 4, 1)
            _conv(d, fg + ".color_consistency.0", cout, cout, 1)
            _conv(d, fg + ".color_consistency.2", cout, cout, 1)
            _conv(d, fg + ".edge_preserve.0", cout // 2, cout, 3)
            _conv(d, fg + ".edge_preserve.2", cout, cout // 2, 3)
            _conv(d, fg + ".conv_out", cout, cout, 3)
        else:
            d[fg + ".dct.dct_matrix"] = (k.dct_block, k.dct_block)
            _conv(d, fg + ".low_freq_attn.0", cout // 2, cout, 1)
            _conv(d, fg + ".low_freq_attn.2", cout, cout // 2, 1)
            _conv(d, fg + ".high_freq_attn.0", cout // 2, cout, 1)
            _conv(d, fg + ".high_freq_attn.2", cout, cout // 2, 1)
            _conv(d, fg + ".conv_out", cout, cout, 3)
        if cin != cout:
            _conv(d, f"{name}.shortcut", cout, cin, 1)
    if arch == ARCH_AVIF:
        _avif_transform(d, "avif_layer", 64)
    else:
        d["dct_layer.dct_matrix"] = (k.dct_block, k.dct_block)
    d["out_conv.0.weight"] = (64,)
    d["out_conv.0.bias"] = (64,)
    _conv(d, "out_conv.2", 3, 64, 3)
    return d


def param_count(arch: int) -> int:
    """Learnable parameters (buffers excluded) -- 114 398 409 WebP/JPEG, 158 284 137 AVIF."""
    n = 0
    for name, shp in state_spec(arch).items():
        if name.endswith("dct_matrix"):
            continue
        m = 1
        for s in shp:
            m *= s
        n += m
    return n


def eps_unet_spec() -> "OrderedDict[str, tuple]":
    """State dict of the early eps-predict ``UNetDenoiser`` (experiments/code/conv.ipynb c0:55-76), the only
    model in the reference with literal predict_noise / p_sample semantics."""
    d: "OrderedDict[str, tuple]" = OrderedDict()

    def dc(name, cin, cout):
        _conv(d, name + ".conv.0", cout, cin, 3)
        _conv(d, name + ".conv.2", cout, cout, 3)

    dc("encoder1", 3, 64)
    dc("encoder2", 64, 128)
    dc("encoder3", 128, 256)
    dc("bottleneck", 256, 512)
    d["up3.weight"] = (512, 256, 2, 2); d["up3.bias"] = (256,)
    dc("decoder3", 512, 256)
    d["up2.weight"] = (256, 128, 2, 2); d["up2.bias"] = (128,)
    dc("decoder2", 256, 128)
    d["up1.weight"] = (128, 64, 2, 2); d["up1.bias"] = (64,)
    dc("decoder1", 128, 64)
    _conv(d, "out_conv", 3, 64, 1)
    _lin(d, "time_embed.0", 512, 1)
    _lin(d, "time_embed.2", 512, 512)
    return d
