cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -s -k "frequency_block_one_pass" > gpurun_out/r2_ff_tests.log 2>&1 || { tail -30 gpurun_out/r2_ff_tests.log; exit 1; }
tail -3 gpurun_out/r2_ff_tests.log
rm -rf gpurun_out/prof_tl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_tl -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2_tl_bench.log 2>&1 || exit 1
python3 tools/forward_timeline.py gpurun_out/prof_tl > gpurun_out/r02_jpeg_forward_timeline.txt
rm -rf gpurun_out/prof_tl
grep "freq_gate\|dispatches" gpurun_out/r02_jpeg_forward_timeline.txt
