cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
source tools/r3_step.sh
mkdir -p gpurun_out/r3
step 700 gpurun_out/r3/gputests.log python -m pytest tests -x -q -m gpu --durations=4
tail -3 gpurun_out/r3/gputests.log
bash tools/prof_all_r03_a.sh
