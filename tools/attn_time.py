import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddpm_image_restoration_amd import ops
shapes = [(4, 65536, 64, 4)] if len(sys.argv) > 1 else [(4, 65536, 64, 4), (4, 16384, 128, 4), (4, 16384, 64, 4), (4, 4096, 256, 4), (4, 1024, 512, 4), (4, 65536, 64, 8)]
for (b, n, c, h) in shapes:
    for mode in ((1,) if len(sys.argv) > 1 else (0, 1)):
        if mode == 0 and n == 65536 and h == 8: continue
        ms = ops.attention_time(b, n, c, h, mode, 2)
        fl = 4.0 * n * n * c * b
        print(f"var={os.environ.get('DDPMIR_ATTN_VARIANT','0')} B={b} N={n} C={c} heads={h} mode={'bf16' if mode else 'f32'}: {ms:.2f} ms  {fl/ms/1e9:.1f} TFLOP/s  {b*h*n*n/ms/1e9:.2f} Tscores/s", flush=True)
