# usage: bash tools/ab_env.sh VAR   -> runs bench with and without VAR=1 in the same box, prints rest-of-forward ms
# The A/B switches only exist in the experiments build: python -m ddpm_image_restoration_amd.build --experiments
# (-> ddpm_image_restoration_amd/libddpmir_exp.so, selected here through DDPMIR_LIB_PATH)
export DDPMIR_LIB_PATH=${DDPMIR_LIB_PATH:-$(dirname $0)/../ddpm_image_restoration_amd/libddpmir_exp.so}
for v in "" "1"; do
  if [ -n "$v" ]; then export $1=1; else unset $1; fi
  python3 bench.py --steps 10 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1=$v', 'ms/step', round(d['ms_per_step'],2), 'attnL0', round(r['avg_launch_ms'],2), 'fwd', round(r['forward_ms'],2), 'rest', round(r['forward_ms']-2*r['avg_launch_ms'],2))"
done
