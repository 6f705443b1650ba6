set -o pipefail
mkdir -p gpurun_out/r2k
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2k/prof64 -- python3 bench.py --batch 64 --steps 20 --warmup 5 --no-cpu-baseline --no-sweep > gpurun_out/r2k/bench64.json 2> gpurun_out/r2k/bench64.err; echo rc=$?
cut -c1-200 gpurun_out/r2k/bench64.json
python tools/trace_summary.py $(find gpurun_out/r2k/prof64 -name "*kernel_trace.csv" | head -1) > gpurun_out/r2k/trace_summary64.txt 2>&1; tail -5 gpurun_out/r2k/trace_summary64.txt
find gpurun_out/r2k -name "*kernel_trace.csv" -delete
