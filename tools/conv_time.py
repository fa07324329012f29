import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddpm_image_restoration_amd import ops
B = 16
cases = [  # (H, Cin, Cout, k) at B=16, 256^2 pyramid
    (256, 64, 64, 3), (256, 128, 64, 3), (256, 64, 192, 1), (256, 64, 32, 1), (256, 32, 64, 1),
    (128, 128, 128, 3), (64, 256, 256, 3), (32, 512, 512, 3), (16, 512, 512, 3), (16, 1024, 512, 3), (8, 1024, 1024, 3), (8, 512, 1024, 3),
    (8, 1024, 3072, 1)]
dbgs = [0] if len(sys.argv) < 2 else [int(v) for v in sys.argv[1].split(",")]
for (h, ci, co, k) in cases:
    fl = 2.0 * B * h * h * ci * co * k * k
    byts = B * h * h * (ci + co) * 4 + ci * co * k * k * 2
    line = f"H={h:3d} {ci:4d}->{co:4d} k{k}: "
    for dbg in dbgs:
        ms = ops.conv_time(B, h, h, ci, co, k, True, False, False, dbg, 5)
        line += f"[dbg{dbg}: {ms*1e3:8.1f} us {fl/ms/1e9:7.1f} TF/s {byts/ms/1e6:7.1f} GB/s] "
    print(line, flush=True)
