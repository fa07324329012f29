cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
source tools/r3_step.sh
O=gpurun_out/r3; mkdir -p $O
step 700 $O/gputests3.log python -m pytest tests -x -q -m gpu --durations=5
tail -4 $O/gputests3.log
step 300 $O/bf16_err.log python tools/bf16_err.py
tail -8 $O/bf16_err.log
step 600 $O/ab_switches1.log python tools/ab_switches.py ",DDPMIR_HA_FP32,DDPMIR_HB_FP32,DDPMIR_HA_FP32+DDPMIR_HB_FP32,DDPMIR_NO_SKIP_STATS,DDPMIR_NO_HEAD_GN_FUSE" both 2
cat $O/ab_switches1.log
