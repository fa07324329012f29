cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
source tools/r3_step.sh
O=gpurun_out/r3; mkdir -p $O
step 700 $O/gputests6.log python -m pytest tests -x -q -m gpu
tail -3 $O/gputests6.log
step 400 $O/bench4.json python bench.py --steps 20 --warmup 5 --no-cpu-baseline
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r3/bench4.json') if l.startswith('{')][-1])
r=d['roofline']; print('headline', round(d['ms_per_step'],3), 'attn', round(r['avg_launch_ms'],3), 'rest', round(r['forward_ms']-2*r['avg_launch_ms'],3))
for w in d['workloads']:
    print(w.get('name','')[:70], w.get('ms_per_step'), w.get('error'))
PY
