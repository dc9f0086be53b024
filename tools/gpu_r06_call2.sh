#!/bin/bash
# round 6, call 2: micro-benchmarks against the round-5 library (graph replay: no eager launch floor) + a kernel trace
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
( timeout 250 python tools/bench_norm.py tools/_scratch/libs/libleco_hip_r05.so 2>&1 | grep -v Warn ) > $O/r06_c2_bench_norm.txt
( timeout 300 python tools/bench_gemm_plain.py tools/_scratch/libs/libleco_hip_r05.so 2>&1 | grep -v Warn ) > $O/r06_c2_bench_gemm_plain.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-dominant --no-telemetry > /tmp/prof.log 2>&1
DB=$(find /tmp/prof -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $DB 60 > $O/r06_c2_kernel_stats.txt 2>&1
cd $R
tail -40 $O/r06_c2_bench_norm.txt; tail -26 $O/r06_c2_bench_gemm_plain.txt; head -45 $O/r06_c2_kernel_stats.txt
