cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
source tools/r3_step.sh
O=gpurun_out/r3; mkdir -p $O
step 700 $O/gputests4.log python -m pytest tests -x -q -m gpu
tail -3 $O/gputests4.log
step 200 $O/pmc_gemm2.log bash tools/pmc_gemm.sh
tail -9 $O/pmc_gemm2.log
step 500 $O/ab_switches2.log python tools/ab_switches.py ",DDPMIR_NO_BUFLOAD" both 2
cat $O/ab_switches2.log
