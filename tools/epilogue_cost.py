"""How much of a full-resolution conv launch is its epilogue?  Experiments build (DDPMIR_LIB_PATH=.../libddpmir_exp.so): dbg bit 3
skips the whole epilogue (stores included), bit 7 (host side) makes the bf16 copy the only output.
  python ddpm_image_restoration_amd/build.py --experiments && DDPMIR_LIB_PATH=ddpm_image_restoration_amd/libddpmir_exp.so python tools/epilogue_cost.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddpm_image_restoration_amd import ops  # noqa: E402

for (b, h, ci, co, k) in ((8, 512, 8, 64, 3), (8, 512, 64, 64, 3), (8, 512, 128, 64, 3), (8, 256, 64, 128, 3), (8, 256, 128, 128, 3), (8, 512, 64, 64, 1), (8, 512, 64, 192, 1)):
    t = {}
    for name, dbg, gelu in (("fp32 + bf16 out", 0, True), ("bf16 out only", 128, True), ("fp32 out only", 0, False), ("no epilogue", 8, True)):
        t[name] = min(ops.conv_time(b, h, h, ci, co, k, "dma", gelu, False, dbg, 5) for _ in range(3)) * 1e3
    fl = 2.0 * b * h * h * ci * co * k * k
    mb16 = b * h * h * co * 2 / 1e6
    print(f"B={b} {h}x{h} {ci:3d}->{co:3d} k{k}: " + " | ".join(f"{n} {v:6.1f} us" for n, v in t.items()) +
          f"   (K loop at {fl / t['no epilogue'] / 1e6:.0f} TF/s; bf16 output {mb16:.0f} MB = {mb16 / 5e3 * 1e3:.0f} us at 5 TB/s)", flush=True)
