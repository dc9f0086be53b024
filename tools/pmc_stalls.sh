#!/bin/bash
# Where do the dominant kernel's wave-cycles go?  Extra counter passes over `bench.py --dominant-only` (the launches of the
# kernel-trace top row), one `--pmc` set per run:  gpurun -- 'bash tools/pmc_stalls.sh'  -> gpurun_out/rNN_pmc_dominant_{waits,insts}.csv
RN=${ROUND:-r03}
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
( rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ|TA|TCP|TCC|TD|GRBM)_[A-Z0-9_]+" | sort -u | tr '\n' ' ' ) > $O/${RN}_counters_available.txt
timeout 120 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES -d /tmp/pmc_w -o dom -- python $R/bench.py --dominant-only --no-cpu-baseline > /tmp/pmc_w.log 2>&1
python $R/tools/pmc_summary.py /tmp/pmc_w > $O/${RN}_pmc_dominant_waits.csv 2>&1
timeout 120 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU -d /tmp/pmc_i -o dom -- python $R/bench.py --dominant-only --no-cpu-baseline > /tmp/pmc_i.log 2>&1
python $R/tools/pmc_summary.py /tmp/pmc_i > $O/${RN}_pmc_dominant_insts.csv 2>&1
timeout 120 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU -d /tmp/pmc_a -o dom -- python $R/bench.py --dominant-only --no-cpu-baseline > /tmp/pmc_a.log 2>&1
python $R/tools/pmc_summary.py /tmp/pmc_a > $O/${RN}_pmc_dominant_active.csv 2>&1
head -5 $O/${RN}_pmc_dominant_waits.csv $O/${RN}_pmc_dominant_insts.csv $O/${RN}_pmc_dominant_active.csv; tail -2 /tmp/pmc_i.log /tmp/pmc_a.log; wc -c $O/${RN}_counters_available.txt
