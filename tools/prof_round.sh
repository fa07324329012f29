# Refreshes the committed profile evidence of one bench workload at HEAD:
#   bash tools/prof_round.sh <tag> <traffic key> [bench.py args ...]          (ROUND=r03 by default; CLASS=gemm for the
#   e.g.  bash tools/prof_round.sh r03_jpeg jpeg_bf16_256_B16_all              bottleneck-attention workloads)
#         bash tools/prof_round.sh r03_avif avif_bf16_256_B16_all --arch avif --T 250
# writes under gpurun_out/ (copy into profiles/ afterwards):
#   <tag>_bench.json                 the bench line (un-profiled run, this workload only)
#   <tag>_kernel_stats.csv           rocprofv3 --kernel-trace --stats of the same command
#   <tag>_per_forward_summary.txt    per-kernel ms per forward + the level-0 attention launches
#   <tag>_hbm_traffic.txt            per-kernel HBM bytes from two PMC passes (FETCH_SIZE x2, WRITE_SIZE)
#   ${ROUND}_traffic.json            bytes per launch of the roofline kernel -> bench.py's roofline.traffic (with the source hash)
tag=$1; key=$2; shift 2
ROUND=${ROUND:-r04}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TJ=gpurun_out/${ROUND}_traffic.json
[ -f profiles/${ROUND}_traffic.json ] && [ ! -f $TJ ] && cp profiles/${ROUND}_traffic.json $TJ
python3 bench.py --no-workloads "$@" > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || exit 1
rm -rf gpurun_out/prof_${tag} gpurun_out/pmcf_${tag} gpurun_out/pmcw_${tag}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag} -- python3 bench.py --no-cpu-baseline --no-workloads "$@" > gpurun_out/prof_${tag}.log 2>&1 || exit 1
python3 tools/prof_summary.py gpurun_out/prof_${tag} 0 > gpurun_out/${tag}_per_forward_summary.txt 2>&1
python3 tools/forward_timeline.py gpurun_out/prof_${tag} > gpurun_out/${tag}_forward_timeline.txt 2>&1
cp "$(ls -t gpurun_out/prof_${tag}/*/*kernel_stats.csv | head -1)" gpurun_out/${tag}_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcf_${tag} -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-workloads "$@" > gpurun_out/pmcf_${tag}.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmcw_${tag} -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-workloads "$@" > gpurun_out/pmcw_${tag}.log 2>&1 || exit 1
python3 tools/pmc_summary.py gpurun_out/pmcf_${tag} gpurun_out/pmcw_${tag} 0 --json $TJ --key ${key} ${CLASS:+--class $CLASS} \
        --command "python3 bench.py --steps 2 --warmup 1 $*" > gpurun_out/${tag}_hbm_traffic.txt 2>&1
rm -rf gpurun_out/prof_${tag} gpurun_out/pmcf_${tag} gpurun_out/pmcw_${tag}
tail -3 gpurun_out/${tag}_per_forward_summary.txt
