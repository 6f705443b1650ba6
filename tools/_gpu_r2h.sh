set -o pipefail
mkdir -p gpurun_out/r2h
python -m pytest tests -m gpu -q -x > gpurun_out/r2h/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r2h/pytest_gpu.log
python bench.py > gpurun_out/r2h/bench.json 2> gpurun_out/r2h/bench.err; echo "bench rc=$?"; cut -c1-700 gpurun_out/r2h/bench.json
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2h/prof -- python3 bench.py --no-cpu-baseline --no-sweep > gpurun_out/r2h/bench_under_rocprof.json 2> gpurun_out/r2h/bench_under_rocprof.err; echo "rocprof rc=$?"
find gpurun_out/r2h/prof -name "*kernel_stats.csv" | head -2
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-24)
  rocprofv3 --pmc $set -d gpurun_out/r2h/pmc_dw/$tag --output-format csv -- python3 tools/dw_ab.py tn_block -1 --rounds 1 --reps 6 > gpurun_out/r2h/pmc_dw_$tag.log 2>&1; echo "pmc $tag rc=$?"
done
python tools/pmc_traffic.py gpurun_out/r2h/pmc_dw gpurun_out/r2h/r02_traffic.json > gpurun_out/r2h/traffic.log 2>&1; tail -25 gpurun_out/r2h/traffic.log
