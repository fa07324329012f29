cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
source tools/r3_step.sh
O=gpurun_out/r3; mkdir -p $O
step 300 $O/persist_tests.log python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv"
tail -3 $O/persist_tests.log
step 300 $O/persist_ab.log python tools/strip_persist_ab.py
cat $O/persist_ab.log
