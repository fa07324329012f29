"""Times every GEMM shape of one JPEG/WebP U-Net forward (B=16, 256x256) through the production conv path and
prints where the GEMM milliseconds go.  usage: python tools/gemm_inventory.py [B] [R]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddpm_image_restoration_amd import ops, spec
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
R = int(sys.argv[2]) if len(sys.argv) > 2 else 256
DBG = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rows = []
ONLY = os.environ.get("ONLY")
for name, cin, cout, lvl in spec.BLOCKS:
    h = R >> lvl
    cin8 = max(cin, 8)
    shapes = [("conv1", cin8, cout, 3), ("conv2", cout, cout, 3), ("in_proj", cout, 3 * cout, 1), ("out_proj", cout, cout, 1),
              ("gate_a", cout, cout // 2, 1), ("gate_b", cout // 2, cout, 1), ("gate_a", cout, cout // 2, 1),
              ("gate_b", cout // 2, cout, 1), ("conv_out", cout, cout, 3)]
    if cin != cout:
        shapes.append(("shortcut", cin8, cout, 1))
    for nm, ci, co, k in shapes:
        if ONLY and f"{h}:{ci}:{co}:{k}" not in ONLY.split(","): continue
        rows.append((name, nm, h, ci, co, k))
cache = {}
tot = 0.0
per_level = {}
for name, nm, h, ci, co, k in rows:
    key = (h, ci, co, k)
    if key not in cache:
        cache[key] = min(ops.conv_time(B, h, h, ci, co, k, "dma", nm in ("conv1","conv2","in_proj","gate_a"), False, DBG, 5) for _ in range(2))
    ms = cache[key]
    fl = 2.0 * B * h * h * ci * co * k * k
    byts = B * h * h * (ci * 2 + co * 4) + ci * co * k * k * 2
    tot += ms
    per_level[h] = per_level.get(h, 0.0) + ms
    print(f"{name:13s} {nm:9s} H={h:3d} {ci:4d}->{co:4d} k{k}: {ms*1e3:7.1f} us {fl/ms/1e9:7.1f} TF/s {byts/ms/1e6:7.1f} GB/s", flush=True)
print(f"total {tot:.3f} ms; per level:", {h: round(v, 3) for h, v in per_level.items()})
