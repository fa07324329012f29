cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_avif -- python3 tools/avif_breakdown.py > gpurun_out/prof_avif.log 2>&1
python3 tools/prof_summary.py gpurun_out/prof_avif 6 > gpurun_out/prof_avif_summary.txt 2>&1
