"""Dissection of the weights-stationary 64->64 conv kernel (experiments build: IGemm::dbg bits 1 = no activation loads,
4 = no fragment reads / MFMAs, 8 = no epilogue) at the configs[4] level-0 shape, with the strip kernel beside it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddpm_image_restoration_amd import ops
for tag, env in (("ws", {}), ("strip", {"DDPMIR_NO_WS64": "1"})):
    for k, v in env.items():
        os.environ[k] = v
    for dbg in (0, 8, 1, 4, 1 | 8, 4 | 8, 1 | 4 | 8):
        t = min(ops.conv_time(8, 512, 512, 64, 64, 3, "dma", True, False, dbg, 3) for _ in range(2))
        print(f"{tag:6s} dbg={dbg:2d} ({'+'.join(n for b, n in ((1, 'noA'), (4, 'noMFMA'), (8, 'noEPI')) if dbg & b) or 'full'}): {t * 1e3:7.1f} us", flush=True)
    for k in env:
        os.environ.pop(k)
