cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py --full --no-cpu-baseline --no-workloads > gpurun_out/r03_jpeg_full_T1000.json 2> gpurun_out/r03_full.err
python3 bench.py --full --no-cpu-baseline --no-workloads --arch avif --T 250 > gpurun_out/r03_avif_full_T250.json 2>> gpurun_out/r03_full.err
python3 bench.py --full --no-cpu-baseline --no-workloads --arch webp > gpurun_out/r03_webp_full_T1000.json 2>> gpurun_out/r03_full.err
python3 tools/graph_vs_eager.py > gpurun_out/r03_timestep_graph_vs_eager.txt 2>&1
tail -n 7 gpurun_out/r03_timestep_graph_vs_eager.txt
