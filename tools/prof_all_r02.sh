set -x
bash tools/prof_round.sh r02_jpeg jpeg_bf16_256_B16_all || echo FAIL jpeg
bash tools/prof_round.sh r02_webp webp_bf16_256_B16_all --arch webp || echo FAIL webp
bash tools/prof_round.sh r02_avif avif_bf16_256_B16_all --arch avif --T 250 || echo FAIL avif
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py --size 512 --attn bottleneck --batch 8 --quality 20 --no-cpu-baseline > gpurun_out/r02_c5_bf16_bench.json 2> gpurun_out/r02_c5.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c5 -- python3 bench.py --size 512 --attn bottleneck --batch 8 --quality 20 --no-cpu-baseline > gpurun_out/prof_c5.log 2>&1
python3 tools/prof_summary.py gpurun_out/prof_c5 0 > gpurun_out/r02_c5_bf16_per_forward_summary.txt 2>&1
rm -rf gpurun_out/prof_c5
cat gpurun_out/r02_c5_bf16_per_forward_summary.txt
