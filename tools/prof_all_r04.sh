# Round-4 profile evidence at HEAD (each line: bench json, rocprofv3 kernel stats, per-forward summary, forward timeline, PMC HBM traffic).
#   gpurun --timeout 1200 -- 'bash tools/prof_all_r04.sh a'   (a: the three families at 256 x 256;  b: configs[4]'s shape in three modes)
set -x
export ROUND=r04
part=${1:-a}
if [ "$part" = a ]; then
  bash tools/prof_round.sh r04_jpeg jpeg_bf16_256_B16_all || echo FAIL jpeg
  bash tools/prof_round.sh r04_webp webp_bf16_256_B16_all --arch webp || echo FAIL webp
  bash tools/prof_round.sh r04_avif avif_bf16_256_B16_all --arch avif --T 250 || echo FAIL avif
else
  C5="--size 512 --attn bottleneck --batch 8 --quality 20"
  CLASS=gemm bash tools/prof_round.sh r04_c5_bf16 jpeg_bf16_512_B8_bottleneck $C5 || echo FAIL c5 bf16
  CLASS=gemm bash tools/prof_round.sh r04_c5_fp8 jpeg_fp8_512_B8_bottleneck $C5 --dtype fp8 || echo FAIL c5 fp8
  CLASS=gemm bash tools/prof_round.sh r04_c5_fp8_64 "jpeg_fp8@64_512_B8_bottleneck" $C5 --dtype fp8 --fp8-min-channels 64 || echo FAIL c5 fp8@64
fi
ls gpurun_out | grep r04_ | head -40
