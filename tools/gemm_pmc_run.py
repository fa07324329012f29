"""One launch set of the dominant conv shapes (for rocprofv3 --pmc runs): 64->64 and 128->128 3x3 at configs[4] sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddpm_image_restoration_amd import ops
if len(sys.argv) > 1 and sys.argv[1] == "r04":   # round 4: the strip conv with bf16 output only (conv2 of a block: dbg bit 7) and with both copies, then the others
    for dbg in (128, 0):
        print(512, 64, 64, dbg, ops.conv_time(8, 512, 512, 64, 64, 3, "dma", True, False, dbg, 3), flush=True)
    for (b, h, ci, co) in ((8, 256, 128, 128), (8, 128, 256, 256)):
        print(h, ci, co, ops.conv_time(b, h, h, ci, co, 3, "dma", True, False, 0, 3), flush=True)
    sys.exit(0)
for (b, h, ci, co) in ((8, 512, 64, 64), (8, 256, 128, 128), (8, 128, 256, 256)):
    print(h, ci, co, ops.conv_time(b, h, h, ci, co, 3, "dma", True, False, 0, 3), flush=True)
