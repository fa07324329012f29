import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddpm_image_restoration_amd import ops
for dbg in (0, 13, 13 | 32, 13 | 64, 13 | 32 | 64, 8, 8 | 32, 1 | 8, 4 | 8):
    ts = []
    for b in (2, 8):
        ts.append(min(ops.conv_time(b, 512, 512, 64, 64, 3, "dma", True, False, dbg, 3) for _ in range(2)) * 1e3)
    print(f"dbg={dbg:3d}: B=2 {ts[0]:7.1f} us  B=8 {ts[1]:7.1f} us  -> {(ts[1] - ts[0]) / 24:5.2f} us per tile", flush=True)
