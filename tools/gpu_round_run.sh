#!/bin/bash
# Round artefacts in ONE GPU call (profiles/ files are copied from gpurun_out/ afterwards):
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/gpu_round_run.sh [notests]'
# 1. kernel trace of the benchmark command  -> r02_step_kernel_stats.txt (its top row names the dominant kernel)
# 2. counter passes (separate runs, --pmc only) over exactly the dominant kernel's launches of one step
#    (`bench.py --dominant-only`) -> r02_pmc_dominant_{mfma,fetch}.csv; and over a whole eager step (k = 4)
# 3. the benchmark itself (reads the files of 1 and 2)  -> r02_bench.json
# 4. the GPU test tier + smoke                          -> r02_gpu_tests.log, smoke.log
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/prof -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /tmp/prof.log 2>&1
DB=$(find /tmp/prof -name "*.db" | head -1)
{ echo "# cd /tmp && rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline   (round 2; tools/gpu_round_run.sh)"
  echo "# SD1.5 bs=2 512^2 rank-4 LECO step; summarised from the rocpd database with tools/rocpd_stats.py"
  python $R/tools/rocpd_stats.py $DB 60; } > $O/r02_step_kernel_stats.txt 2>&1
cp $O/r02_step_kernel_stats.txt $R/profiles/r02_step_kernel_stats.txt       # bench.py --dominant-only reads the top row
timeout 100 rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/pmc_dm -o dom -- python $R/bench.py --dominant-only --no-cpu-baseline > /tmp/pmc_dm.log 2>&1
python $R/tools/pmc_summary.py /tmp/pmc_dm > $O/r02_pmc_dominant_mfma.csv 2>&1
timeout 100 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d /tmp/pmc_df -o dom -- python $R/bench.py --dominant-only --no-cpu-baseline > /tmp/pmc_df.log 2>&1
python $R/tools/pmc_summary.py /tmp/pmc_df > $O/r02_pmc_dominant_fetch.csv 2>&1
cp $O/r02_pmc_dominant_mfma.csv $O/r02_pmc_dominant_fetch.csv $R/profiles/
timeout 130 rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/pmc_mfma -o bench -- python $R/bench.py --steps 1 --warmup 0 --k 4 --no-cpu-baseline --no-graphs > /tmp/pmc1.log 2>&1
python $R/tools/pmc_summary.py /tmp/pmc_mfma > $O/r02_pmc_mfma_step.csv 2>&1
timeout 110 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d /tmp/pmc_fetch -o bench -- python $R/bench.py --steps 1 --warmup 0 --k 4 --no-cpu-baseline --no-graphs > /tmp/pmc2.log 2>&1
python $R/tools/pmc_summary.py /tmp/pmc_fetch > $O/r02_pmc_fetch_step.csv 2>&1
cd $R
( timeout 400 python bench.py 2>/dev/null | tail -1 ) > $O/r02_bench.json
if [ "$1" != "notests" ]; then
  ( timeout 900 python -m pytest tests -q -m gpu -s 2>&1 | grep -vE "^W2026|Warn|warn|hipGraph|\^~|^ +[0-9]+ \|" | tail -90 ) > $O/r02_gpu_tests.log 2>&1
  ( timeout 100 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 ) > $O/smoke.log 2>&1
  tail -4 $O/r02_gpu_tests.log; cat $O/smoke.log
fi
cut -c1-1500 $O/r02_bench.json; head -12 $O/r02_step_kernel_stats.txt; head -4 $O/r02_pmc_dominant_mfma.csv; head -3 $O/r02_pmc_dominant_fetch.csv; tail -3 /tmp/pmc_dm.log
