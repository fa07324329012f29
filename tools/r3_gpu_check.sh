# One GPU call of the round's routine check: the whole -m gpu suite, then the default bench line (with its workloads).
#   gpurun --timeout 1150 -- 'bash tools/r3_gpu_check.sh'
# Every step runs under its own timeout; a step that is killed stops the call (tools/r3_step.sh).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
source tools/r3_step.sh
O=gpurun_out/r3; mkdir -p $O
step 700 $O/gputests.log python -m pytest tests -x -q -m gpu --durations=6
tail -4 $O/gputests.log
step 400 $O/bench.json python bench.py --steps 20 --warmup 5
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r3/bench.json') if l.startswith('{')][-1])
r = d['roofline']
print('headline', round(d['ms_per_step'], 3), 'ms/step', round(d['value'], 4), 'img/s | attn', round(r['avg_launch_ms'], 3), 'frac', round(r['frac'], 4),
      'rest', round(r['forward_ms'] - 2 * r['avg_launch_ms'], 3), 'traffic stale', (r.get('traffic_source') or {}).get('stale'))
for w in d.get('workloads', []):
    rr = w.get('roofline') or {}
    print(' -', (w.get('name') or '')[:66], w.get('ms_per_step') and round(w['ms_per_step'], 3), 'frac', rr.get('frac') and round(rr['frac'], 3), w.get('error') or '')
PY
