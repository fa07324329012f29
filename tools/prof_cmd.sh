cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_g16 -- python3 bench.py --steps 2 --warmup 1 > gpurun_out/prof_g16.log 2>&1
python3 tools/prof_summary.py gpurun_out/prof_g16 3 gemm16 > gpurun_out/prof_g16_summary.txt 2>&1
