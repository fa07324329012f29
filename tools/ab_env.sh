# usage: bash tools/ab_env.sh VAR   -> runs bench with and without VAR=1 in the same box, prints rest-of-forward ms
for v in "" "1"; do
  if [ -n "$v" ]; then export $1=1; else unset $1; fi
  python3 bench.py --steps 10 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1=$v', 'ms/step', round(d['ms_per_step'],2), 'attnL0', round(r['avg_launch_ms'],2), 'fwd', round(r['forward_ms'],2), 'rest', round(r['forward_ms']-2*r['avg_launch_ms'],2))"
done
