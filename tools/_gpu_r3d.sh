set -e
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
D=$R/gpurun_out/pmc_dw3
rm -rf $D
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
tag=$(echo $set | cut -d' ' -f1)
rocprofv3 --pmc $set -d $D/$tag --output-format csv -- python3 $R/tools/dw_ab.py tn_block -1 --rounds 1 --reps 6 > $D.$tag.log 2>&1
echo "pass $tag done"
done
cd $R
python3 tools/pmc_traffic.py gpurun_out/pmc_dw3 gpurun_out/r3d_traffic.json > gpurun_out/r3d_pmc_summary.txt 2>&1
cat gpurun_out/r3d_pmc_summary.txt | tail -5
find gpurun_out/pmc_dw3 -name "*.csv" -size +2M -delete
timeout -k 10 500 python tools/race_screen.py > gpurun_out/r3d_race.log 2>&1 || { tail -20 gpurun_out/r3d_race.log; exit 1; }
tail -5 gpurun_out/r3d_race.log
