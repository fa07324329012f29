"""One launch set of the dominant conv shapes (for rocprofv3 --pmc runs): 64->64 and 128->128 3x3 at configs[4] sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddpm_image_restoration_amd import ops
for (b, h, ci, co) in ((8, 512, 64, 64), (8, 256, 128, 128), (8, 128, 256, 256)):
    print(h, ci, co, ops.conv_time(b, h, h, ci, co, 3, "dma", True, False, 0, 3), flush=True)
