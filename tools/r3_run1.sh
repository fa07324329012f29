cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
source tools/r3_step.sh
O=gpurun_out/r3; mkdir -p $O
step 600 $O/gputests1.log python -m pytest tests -x -q -m gpu --durations=15
tail -4 $O/gputests1.log
step 500 $O/bench1.json python bench.py --steps 20 --warmup 5
tail -c 600 $O/bench1.json
step 300 $O/pmc_attn.log bash tools/pmc_attn32.sh
tail -8 $O/pmc_attn.log
step 500 $O/fp8_eval.log python tools/fp8_eval.py
tail -12 $O/fp8_eval.log
