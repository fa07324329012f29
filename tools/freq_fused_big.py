"""The one-pass frequency kernel against the three kernels it replaces at sizes with several workgroups per CU (the unit test's
shapes have at most 64 tiles), with and without the attention's output projection formed inside; repeated for determinism."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ddpm_image_restoration_amd import ops
from ddpm_image_restoration_amd.synth import dct_matrix
rnd = lambda *s, seed=0, scale=1.0: (np.random.default_rng(seed).standard_normal(s) * scale).astype(np.float32)
nhwc = lambda a: np.ascontiguousarray(a.transpose(0, 2, 3, 1))
for (B, H, W, bs, low, C) in [(1, 256, 256, 8, 4, 64), (2, 256, 256, 8, 4, 64), (1, 128, 128, 8, 4, 128), (4, 128, 128, 8, 4, 128), (2, 256, 256, 4, 3, 64)]:
    x = rnd(B, C, H, W, seed=50); D = dct_matrix(bs)
    w1 = rnd(C, C, seed=51, scale=0.2); b1 = rnd(C, seed=52, scale=0.3)
    w2 = rnd(2 * C, C, seed=53, scale=0.3); w2[:C, C // 2:] = 0; w2[C:, :C // 2] = 0
    b2 = rnd(2 * C, seed=54, scale=0.3); boost = np.abs(rnd(B, seed=55)) + 0.5
    att, wo, bo = rnd(B, C, H, W, seed=56), rnd(C, C, seed=57, scale=0.15), rnd(C, seed=58, scale=0.2)
    three = ops.freq_gates_bf16(nhwc(x), D, w1, b1, w2, b2, boost, low, fused=False)
    t2 = ops.freq_gates_bf16(nhwc(x), D, w1, b1, w2, b2, boost, low, fused=False, att=nhwc(att), wo=wo, bo=bo)
    for rep in range(3):
        f = ops.freq_gates_bf16(nhwc(x), D, w1, b1, w2, b2, boost, low, fused=True)
        f2 = ops.freq_gates_bf16(nhwc(x), D, w1, b1, w2, b2, boost, low, fused=True, att=nhwc(att), wo=wo, bo=bo)
        d, d2 = (f != three), (f2 != t2)
        print((B, H, W, bs, low, C), "rep", rep, "plain: differing", int(d.sum()), "max", float(np.abs(f - three).max()),
              "| with out_proj: differing", int(d2.sum()), "max", float(np.abs(f2 - t2).max()), flush=True)
        if d.any():
            idx = np.argwhere(d)[:5]; print("   first plain diffs (b, y, x, c):", idx.tolist())
        if d2.any():
            idx = np.argwhere(d2)[:5]; print("   first out_proj diffs (b, y, x, c):", idx.tolist())
