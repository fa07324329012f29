set -x
export ROUND=r03
C5="--size 512 --attn bottleneck --batch 8 --quality 20"
CLASS=gemm bash tools/prof_round.sh r03_c5_bf16 jpeg_bf16_512_B8_bottleneck $C5 || echo FAIL c5 bf16
CLASS=gemm bash tools/prof_round.sh r03_c5_fp8 jpeg_fp8_512_B8_bottleneck $C5 --dtype fp8 || echo FAIL c5 fp8
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py --full --no-cpu-baseline --no-workloads > gpurun_out/r03_jpeg_full_T1000.json 2> gpurun_out/r03_full.err
python3 bench.py --full --no-cpu-baseline --no-workloads --arch avif --T 250 > gpurun_out/r03_avif_full_T250.json 2>> gpurun_out/r03_full.err
python3 bench.py --full --no-cpu-baseline --no-workloads --arch webp > gpurun_out/r03_webp_full_T1000.json 2>> gpurun_out/r03_full.err
tail -n 1 gpurun_out/r03_c5_bf16_forward_timeline.txt
