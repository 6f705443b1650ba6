set -e
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_vitb -o vitb -- python3 $R/bench.py --model base --classes 10 --batch 256 --steps 10 --warmup 3 --no-cpu-baseline --no-sweep > $R/gpurun_out/r2z_vitb.log 2>&1
cd $R
f=$(find gpurun_out/prof_vitb -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/r2z_vitb_kernel_stats.csv
find gpurun_out/prof_vitb -name "*kernel_trace.csv" -delete
grep '^{' gpurun_out/r2z_vitb.log | cut -c1-400
