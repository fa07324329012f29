cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
source tools/r3_step.sh
O=gpurun_out/r3; mkdir -p $O
step 700 $O/gputests5.log python -m pytest tests -x -q -m gpu --durations=6
tail -12 $O/gputests5.log
step 400 $O/bench3.json python bench.py --steps 20 --warmup 5
tail -c 400 $O/bench3.json
