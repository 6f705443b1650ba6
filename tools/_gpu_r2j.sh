set -o pipefail
mkdir -p gpurun_out/r2j
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python tools/infer_bench.py > gpurun_out/r2j/infer_large.json 2> gpurun_out/r2j/infer.err; cat gpurun_out/r2j/infer_large.json
python tools/infer_bench.py --model base --classes 10 > gpurun_out/r2j/infer_base.json 2>> gpurun_out/r2j/infer.err; cat gpurun_out/r2j/infer_base.json
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d gpurun_out/r2j/pmc_step/p$i --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-sweep > gpurun_out/r2j/pmc_step_p$i.log 2>&1; echo "pmc pass $i rc=$?"
done
python tools/pmc_summary.py gpurun_out/r2j/pmc_step > gpurun_out/r2j/pmc_step_summary.txt 2>&1; wc -l gpurun_out/r2j/pmc_step_summary.txt
find gpurun_out/r2j/pmc_step -name "*counter_collection.csv" -size +20M -delete
python bench.py --gpus 2 --rehearse-one-gpu --bf16-buckets --batch 32 --steps 4 --warmup 2 --no-sweep --no-cpu-baseline > gpurun_out/r2j/bench2_bf16.json 2> gpurun_out/r2j/bench2_bf16.err; echo "rehearsal rc=$?"; cut -c1-200 gpurun_out/r2j/bench2_bf16.json
