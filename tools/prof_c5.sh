cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_c5
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_c5 -- python3 bench.py --size 512 --batch 8 --quality 20 --attn bottleneck --no-cpu-baseline --steps 4 > gpurun_out/prof_c5.log 2>&1
python3 tools/prof_summary.py gpurun_out/prof_c5 5 > gpurun_out/prof_c5_summary.txt 2>&1
