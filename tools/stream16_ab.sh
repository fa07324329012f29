# bf16 activation stream on / off at the configs[4] shape in bf16 mode: per-kernel time per forward of both (rocprofv3 --stats)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
C5="--size 512 --attn bottleneck --batch 8 --quality 20 --no-cpu-baseline --no-workloads --steps 10 --warmup 3"
for v in 0 1; do
  rm -rf gpurun_out/prof_s16_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_s16_$v -- python3 bench.py $C5 --stream16 $v > gpurun_out/prof_s16_$v.log 2>&1 || exit 1
  python3 tools/prof_summary.py gpurun_out/prof_s16_$v 0 > gpurun_out/r03_stream16_${v}_per_forward_summary.txt 2>&1
  python3 tools/forward_timeline.py gpurun_out/prof_s16_$v > gpurun_out/r03_stream16_${v}_forward_timeline.txt 2>&1
  rm -rf gpurun_out/prof_s16_$v
done
head -30 gpurun_out/r03_stream16_1_per_forward_summary.txt
