# lean refresh of the GEMM-class traffic keys of the configs[4] workloads (two PMC passes each, no kernel trace): for source changes
# that do not touch their kernels (they share the library's source hash with everything else)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TJ=gpurun_out/r03_traffic.json
[ -f profiles/r03_traffic.json ] && [ ! -f $TJ ] && cp profiles/r03_traffic.json $TJ
C5="--size 512 --attn bottleneck --batch 8 --quality 20"
run() {  # tag key args...
  tag=$1; key=$2; shift 2
  rm -rf gpurun_out/pmcf_$tag gpurun_out/pmcw_$tag
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcf_$tag -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-workloads "$@" > gpurun_out/pmcf_$tag.log 2>&1 || return 1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmcw_$tag -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-workloads "$@" > gpurun_out/pmcw_$tag.log 2>&1 || return 1
  python3 tools/pmc_summary.py gpurun_out/pmcf_$tag gpurun_out/pmcw_$tag 0 --json $TJ --key $key --class gemm --command "python3 bench.py --steps 2 --warmup 1 $*" > gpurun_out/${tag}_hbm_traffic.txt 2>&1
  rm -rf gpurun_out/pmcf_$tag gpurun_out/pmcw_$tag
  tail -n 1 gpurun_out/${tag}_hbm_traffic.txt
}
run r03_c5_bf16 jpeg_bf16_512_B8_bottleneck $C5
run r03_c5_fp8 jpeg_fp8_512_B8_bottleneck $C5 --dtype fp8
run r03_c5_fp8_64 "jpeg_fp8@64_512_B8_bottleneck" $C5 --dtype fp8 --fp8-min-channels 64
cp $TJ profiles/r03_traffic.json   # (the box's copy: the default line below reads it)
python3 bench.py --steps 20 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
