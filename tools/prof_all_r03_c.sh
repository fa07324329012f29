set -x
export ROUND=r03
C5="--size 512 --attn bottleneck --batch 8 --quality 20"
CLASS=gemm bash tools/prof_round.sh r03_c5_fp8_64 jpeg_fp8@64_512_B8_bottleneck $C5 --dtype fp8 --fp8-min-channels 64 || echo FAIL c5 fp8@64
bash tools/stream16_ab.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 tools/graph_vs_eager.py > gpurun_out/r03_timestep_graph_vs_eager.txt 2>&1
tail -n 4 gpurun_out/r03_timestep_graph_vs_eager.txt
