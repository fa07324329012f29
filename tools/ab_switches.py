"""A/B of experiment switches (experiments build, DDPMIR_LIB_PATH) on ONE device: bench slices in interleaved child processes.
usage: ab_switches.py "SW1,SW2+SW3,SW4=3,..." [c5|head|both] [rounds]   ('' = no switch; SW2+SW3 = both set; SW=value)"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sets = [s for s in sys.argv[1].split(",")]
which = sys.argv[2] if len(sys.argv) > 2 else "both"
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
WL = {"c5": ["--size", "512", "--batch", "8", "--attn", "bottleneck", "--quality", "20", "--steps", "20", "--warmup", "4"],
      "head": ["--steps", "10", "--warmup", "2"]}
names = ["c5", "head"] if which == "both" else [which]
res = {(w, s): [] for w in names for s in sets}
for rnd in range(rounds):
    for w in names:
        for s in sets:
            env = dict(os.environ, DDPMIR_LIB_PATH=os.path.join(root, "ddpm_image_restoration_amd", "libddpmir_exp.so"))
            for v in [x for x in s.split("+") if x]:
                k, _, val = v.partition("=")       # SW or SW=value
                env[k] = val or "1"
            out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--no-workloads"] + WL[w], env=env, capture_output=True, text=True, cwd=root)
            try:
                d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
                r = d.get("roofline") or {}
                rest = (r["forward_ms"] - 2 * r["avg_launch_ms"]) if (w == "head" and r) else d["ms_per_step"]
                res[(w, s)].append((d["ms_per_step"], rest))
            except Exception as e:
                res[(w, s)].append((float("nan"), float("nan")))
                print("failed", w, s, out.stderr[-300:], flush=True)
for w in names:
    for s in sets:
        v = res[(w, s)]
        print(f"{w:5s} [{s or 'default':40s}] ms/step {['%.3f' % a for a, _ in v]}  {'forward minus level-0 attention' if w == 'head' else 'step'} {['%.3f' % b for _, b in v]}", flush=True)
