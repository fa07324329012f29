# HBM-side traffic per kernel of the default bench workload: two separate PMC passes (FETCH_SIZE, WRITE_SIZE), no
# trace domains besides the counters (MI355X_MICROARCH.md, HBM section: FETCH_SIZE x2 on gfx950).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_write.log 2>&1 || exit 1
python3 tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write 3 > gpurun_out/r01_bench_bf16_hbm_traffic.txt 2>&1
