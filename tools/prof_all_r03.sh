# Regenerates the committed r03 profile artefacts at HEAD (copy gpurun_out/r03_* into profiles/ afterwards): about 12 GPU-minutes
set -x
export ROUND=r03
bash tools/prof_round.sh r03_jpeg jpeg_bf16_256_B16_all || echo FAIL jpeg
bash tools/prof_round.sh r03_webp webp_bf16_256_B16_all --arch webp || echo FAIL webp
bash tools/prof_round.sh r03_avif avif_bf16_256_B16_all --arch avif --T 250 || echo FAIL avif
C5="--size 512 --attn bottleneck --batch 8 --quality 20"
CLASS=gemm bash tools/prof_round.sh r03_c5_bf16 jpeg_bf16_512_B8_bottleneck $C5 || echo FAIL c5 bf16
CLASS=gemm bash tools/prof_round.sh r03_c5_fp8 jpeg_fp8_512_B8_bottleneck $C5 --dtype fp8 || echo FAIL c5 fp8
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# complete restorations
python3 bench.py --full --no-cpu-baseline --no-workloads > gpurun_out/r03_jpeg_full_T1000.json 2> gpurun_out/r03_full.err
python3 bench.py --full --no-cpu-baseline --no-workloads --arch avif --T 250 > gpurun_out/r03_avif_full_T250.json 2>> gpurun_out/r03_full.err
python3 bench.py --full --no-cpu-baseline --no-workloads --arch webp > gpurun_out/r03_webp_full_T1000.json 2>> gpurun_out/r03_full.err
python3 tools/graph_vs_eager.py > gpurun_out/r03_timestep_graph_vs_eager.txt 2>&1
tail -n 1 gpurun_out/r03_jpeg_forward_timeline.txt; tail -n 1 gpurun_out/r03_c5_bf16_forward_timeline.txt
