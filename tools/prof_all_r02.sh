# Regenerates every committed r02 profile artefact at HEAD (copy gpurun_out/r02_* into profiles/ afterwards):
#   bash tools/prof_all_r02.sh            (about 10 GPU-minutes)
set -x
bash tools/prof_round.sh r02_jpeg jpeg_bf16_256_B16_all || echo FAIL jpeg
bash tools/prof_round.sh r02_webp webp_bf16_256_B16_all --arch webp || echo FAIL webp
bash tools/prof_round.sh r02_avif avif_bf16_256_B16_all --arch avif --T 250 || echo FAIL avif
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
C5="--size 512 --attn bottleneck --batch 8 --quality 20 --no-cpu-baseline"
python3 bench.py $C5 > gpurun_out/r02_c5_bf16_bench.json 2> gpurun_out/r02_c5.err
python3 bench.py $C5 --dtype fp8 > gpurun_out/r02_c5_fp8_bench.json 2>> gpurun_out/r02_c5.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c5 -- python3 bench.py $C5 > gpurun_out/prof_c5.log 2>&1
python3 tools/prof_summary.py gpurun_out/prof_c5 0 > gpurun_out/r02_c5_bf16_per_forward_summary.txt 2>&1
python3 tools/forward_timeline.py gpurun_out/prof_c5 > gpurun_out/r02_c5_forward_timeline.txt 2>&1
rm -rf gpurun_out/prof_c5
# per-dispatch timelines of one timestep
for a in jpeg webp avif; do
  T=1000; [ $a = avif ] && T=250
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_tl -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --arch $a --T $T > gpurun_out/prof_tl.log 2>&1
  python3 tools/forward_timeline.py gpurun_out/prof_tl > gpurun_out/r02_${a}_forward_timeline.txt 2>&1
  rm -rf gpurun_out/prof_tl
done
# complete restorations
python3 bench.py --full --no-cpu-baseline > gpurun_out/r02_jpeg_full_T1000.json 2>> gpurun_out/r02_c5.err
python3 bench.py --full --no-cpu-baseline --arch avif --T 250 > gpurun_out/r02_avif_full_T250.json 2>> gpurun_out/r02_c5.err
python3 bench.py --full --no-cpu-baseline --arch webp > gpurun_out/r02_webp_full_T1000.json 2>> gpurun_out/r02_c5.err
tail -n 1 gpurun_out/r02_jpeg_forward_timeline.txt; tail -n 1 gpurun_out/r02_c5_forward_timeline.txt
