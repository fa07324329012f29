"""Is the level-0 attention launch quantised in rounds of resident workgroups?  flash_bf16_nm<16, 4, 8, 0>: one workgroup = 512
queries of one (batch, head), three workgroups per CU = 768 resident; B = 16 gives 8192 workgroups = 10.67 rounds.  If the launch
costs 11 rounds, B = 15 (7680 = 10 rounds) costs 10/11 of it, not 15/16."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddpm_image_restoration_amd import ops  # noqa: E402

n, c, h = 65536, 64, 4
res = {}
for rep in range(2):
    for b in (3, 6, 9, 12, 13, 14, 15, 16, 17, 18):
        ms = ops.attention_time(b, n, c, h, 1, 2)
        res[b] = min(res.get(b, 1e9), ms)
for b, ms in sorted(res.items()):
    wgs = b * h * n // 512
    print(f"B={b:2d}: {wgs:5d} workgroups = {wgs / 768:6.2f} rounds of 768: {ms:7.3f} ms = {ms / (wgs / 768):.4f} ms per round-equivalent, {ms / -(-wgs // 768):.4f} per whole round, {ms / b:.4f} ms per image", flush=True)
