set -x
export ROUND=r03
bash tools/prof_round.sh r03_jpeg jpeg_bf16_256_B16_all || echo FAIL jpeg
bash tools/prof_round.sh r03_webp webp_bf16_256_B16_all --arch webp || echo FAIL webp
bash tools/prof_round.sh r03_avif avif_bf16_256_B16_all --arch avif --T 250 || echo FAIL avif
