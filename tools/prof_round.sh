# Refreshes the committed profile evidence: default bench line + rocprofv3 kernel stats of the same command.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python3 bench.py > gpurun_out/r01_bench_bf16_default.json 2> gpurun_out/r01_bench.err || exit 1
rm -rf gpurun_out/prof_r01
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r01 -- python3 bench.py > gpurun_out/prof_r01.log 2>&1 || exit 1
python3 tools/prof_summary.py gpurun_out/prof_r01 6 > gpurun_out/r01_bench_bf16_per_forward_summary.txt 2>&1
cp "$(ls -t gpurun_out/prof_r01/*/*kernel_stats.csv | head -1)" gpurun_out/r01_bench_bf16_kernel_stats.csv
