R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 200 python -m pytest tests -q -m gpu -s 2>&1 | grep -vE "^W2026|Warn" | tail -25 ) > gpurun_out/r01_gpu_tests.log 2>&1
( timeout 100 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 ) > gpurun_out/smoke.log 2>&1
( timeout 150 python bench.py 2>/dev/null | tail -1 ) > gpurun_out/r01_bench.json
cd /tmp
timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/prof -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /tmp/prof.log 2>&1
DB=$(find /tmp/prof -name "*.db" | head -1); python $R/tools/rocpd_stats.py $DB 60 > $R/gpurun_out/r01_step_kernel_stats.txt 2>&1
timeout 110 rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/pmc_mfma -o bench -- python $R/bench.py --steps 1 --warmup 0 --k 4 --no-cpu-baseline --no-graphs > /tmp/pmc1.log 2>&1
python $R/tools/pmc_summary.py /tmp/pmc_mfma > $R/gpurun_out/r01_pmc_mfma_step.csv 2>&1
timeout 90 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d /tmp/pmc_fetch -o bench -- python $R/bench.py --steps 1 --warmup 0 --k 4 --no-cpu-baseline --no-graphs > /tmp/pmc2.log 2>&1
python $R/tools/pmc_summary.py /tmp/pmc_fetch > $R/gpurun_out/r01_pmc_fetch_step.csv 2>&1
cd $R; tail -4 gpurun_out/r01_gpu_tests.log; cat gpurun_out/smoke.log; cut -c1-300 gpurun_out/r01_bench.json; head -5 gpurun_out/r01_step_kernel_stats.txt; head -4 gpurun_out/r01_pmc_mfma_step.csv; head -3 gpurun_out/r01_pmc_fetch_step.csv
