set -o pipefail
mkdir -p gpurun_out/r2m
python -m pytest tests -m gpu -q > gpurun_out/r2m/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r2m/pytest_gpu.log
python __graft_entry__.py smoke > gpurun_out/r2m/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r2m/smoke.log
python tools/race_screen.py > gpurun_out/r2m/race_screen.log 2>&1; echo "race rc=$?"; grep -E "CLEAN|MISMATCH|spread|screen" gpurun_out/r2m/race_screen.log | tail -6
python bench.py > gpurun_out/r2m/bench.json 2> gpurun_out/r2m/bench.err; echo "bench rc=$?"; cut -c1-260 gpurun_out/r2m/bench.json
python bench.py --aug --no-cpu-baseline --no-sweep > gpurun_out/r2m/bench_aug.json 2> gpurun_out/r2m/bench_aug.err; cut -c1-160 gpurun_out/r2m/bench_aug.json
python bench.py --autograd --no-cpu-baseline --no-sweep > gpurun_out/r2m/bench_autograd.json 2> gpurun_out/r2m/bench_autograd.err; cut -c1-160 gpurun_out/r2m/bench_autograd.json
python bench.py --aug --batch 64 --no-cpu-baseline --no-sweep > gpurun_out/r2m/bench_aug64.json 2> gpurun_out/r2m/bench_aug64.err; cut -c1-160 gpurun_out/r2m/bench_aug64.json
python bench.py --batch 64 --no-cpu-baseline --no-sweep > gpurun_out/r2m/bench_64.json 2> gpurun_out/r2m/bench_64.err; cut -c1-160 gpurun_out/r2m/bench_64.json
