# Round-4 closing evidence on one box: default bench line, full-T runs of the three families, the frequency kernel's phase probe,
# its many-workgroups check, and the library A/B (tools/ab/head.so = the build before DESIGN.md 4.10's work, new.so = HEAD).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err
cp gpurun_out/bench_workloads.json gpurun_out/r04_bench_workloads.json
tail -c 600 gpurun_out/r04_bench_default.json; echo
python3 bench.py --full --no-cpu-baseline --no-workloads > gpurun_out/r04_jpeg_full_T1000.json 2> gpurun_out/r04_full.err
python3 bench.py --full --no-cpu-baseline --no-workloads --arch avif --T 250 > gpurun_out/r04_avif_full_T250.json 2>> gpurun_out/r04_full.err
python3 bench.py --full --no-cpu-baseline --no-workloads --arch webp > gpurun_out/r04_webp_full_T1000.json 2>> gpurun_out/r04_full.err
(cd tools/micro && hipcc -O3 -std=c++17 --offload-arch=gfx950 -I ../../ddpm_image_restoration_amd/csrc -o freq_phase freq_phase.hip 2>/dev/null; (./freq_phase 8 1; ./freq_phase 8 0; ./freq_phase 1 1) > ../../gpurun_out/r04_freq_phase.txt 2>&1)
python3 tools/freq_fused_big.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_freq_fused_many_workgroups.txt
[ -f tools/ab/head.so ] && python3 tools/lib_ab_model.py head.so new.so > gpurun_out/r04_freq_fused_ab.txt 2>&1
python3 tools/bf16_err.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_bf16_mode_forward_error.txt
python3 tools/bf16_err256.py 2>&1 | grep 256 >> gpurun_out/r04_bf16_mode_forward_error.txt
tail -n 3 gpurun_out/r04_freq_fused_ab.txt gpurun_out/r04_bf16_mode_forward_error.txt
