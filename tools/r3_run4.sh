cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
source tools/r3_step.sh
O=gpurun_out/r3; mkdir -p $O
step 700 $O/gputests2.log python -m pytest tests -x -q -m gpu --durations=8
tail -4 $O/gputests2.log
step 300 $O/bench2.json python bench.py --steps 10 --warmup 3 --no-cpu-baseline --workloads 0,2,3
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r3/bench2.json') if l.startswith('{')][-1])
print('headline', round(d['ms_per_step'],3), 'fwd', round(d['roofline']['forward_ms'],3), 'attn', round(d['roofline']['avg_launch_ms'],3))
for w in d['workloads']:
    print(w.get('name','')[:60], w.get('ms_per_step'), w.get('error'))
PY
