"""Round 2's final tree (a copy under .ab_r2/, built from commit a6e9c77) against HEAD on ONE device: interleaved child processes of
each tree's own bench.py.  usage: python tools/r2_vs_r3.py ["configs[1]"]   (inside a gpurun call; .ab_r2/ is git-ignored)
.ab_r2/ is made with:  git worktree add /tmp/r2tree a6e9c77; mkdir .ab_r2; cp -r /tmp/r2tree/{ddpm_image_restoration_amd,bench.py,
__graft_entry__.py,oracle,include,profiles} .ab_r2/; git worktree remove --force /tmp/r2tree; (cd .ab_r2 && python -c
"from ddpm_image_restoration_amd import build; build.build(force=True)")   -- 50 s of hipcc; it travels to the GPU box with the snapshot."""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
trees = {"r02 (a6e9c77)": os.path.join(root, ".ab_r2"), "r03 (HEAD)": root}
WL = {"configs[1] JPEG 256^2 B=16": [],
      "configs[4] shape bf16": ["--size", "512", "--batch", "8", "--attn", "bottleneck", "--quality", "20"],
      "configs[4] shape fp8": ["--size", "512", "--batch", "8", "--attn", "bottleneck", "--quality", "20", "--dtype", "fp8"]}
if len(sys.argv) > 1:   # e.g. "configs[1]": only the workloads whose name starts with it
    WL = {k: v for k, v in WL.items() if k.startswith(sys.argv[1])}
res = {}
for rnd in range(2):
    for wname, wargs in WL.items():
        for tname, tdir in trees.items():
            extra = ["--no-workloads"] if tdir == root else []
            out = subprocess.run([sys.executable, "bench.py", "--steps", "10", "--warmup", "3", "--no-cpu-baseline"] + wargs + extra,
                                 capture_output=True, text=True, cwd=tdir)
            try:
                d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
                r = d.get("roofline") or {}
                rest = (r["forward_ms"] - 2 * r["avg_launch_ms"]) if r.get("avg_launch_ms") and r.get("forward_ms") else None
                res.setdefault((wname, tname), []).append((d["ms_per_step"], r.get("avg_launch_ms"), rest))
            except Exception:
                print("failed", wname, tname, out.stderr[-400:], flush=True)
for (wname, tname), v in res.items():
    print(f"{wname:28s} {tname:14s} ms/step {['%.3f' % a for a, _, _ in v]}" +
          (f"  level-0 attention launch {['%.3f' % b for _, b, _ in v]}  forward minus level-0 attention {['%.3f' % c for _, _, c in v]}" if v[0][2] else ""), flush=True)
