cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_f32
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_f32 -- python3 bench.py --dtype f32 --no-cpu-baseline > gpurun_out/prof_f32.log 2>&1 || exit 1
cp "$(ls -t gpurun_out/prof_f32/*/*kernel_stats.csv | head -1)" gpurun_out/r01_bench_f32_kernel_stats.csv
