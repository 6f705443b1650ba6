set -e
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_b8b -o b8 -- python3 $R/bench.py --batch 8 --steps 20 --warmup 5 --no-cpu-baseline --no-sweep > $R/gpurun_out/r3i_b8.log 2>&1
cd $R
f=$(find gpurun_out/prof_b8b -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/r3i_b8_kernel_stats.csv
find gpurun_out/prof_b8b -name "*kernel_trace.csv" -delete
