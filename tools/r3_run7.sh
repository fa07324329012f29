cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
source tools/r3_step.sh
O=gpurun_out/r3; mkdir -p $O
step 300 $O/ws_tests.log python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv_bf16_dma"
tail -5 $O/ws_tests.log
step 120 $O/ws_time.log python tools/gemm_pmc_run.py
cat $O/ws_time.log
