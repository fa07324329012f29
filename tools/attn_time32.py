"""One launch set of the level-1 (head_dim 32, N = 16 384) and level-0 (head_dim 16, N = 65 536) attention shapes, for PMC runs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddpm_image_restoration_amd import ops
print("d32", ops.attention_time(16, 16384, 128, 4, 1, 2))
print("d16", ops.attention_time(4, 65536, 64, 4, 1, 2))
