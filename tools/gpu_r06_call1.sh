#!/bin/bash
# round 6, call 1: norm kernels + split-K finish + residual prefetch against the round-5 library
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_kernels.py tests/test_table_shapes.py -q -m gpu -x 2>&1 | tail -8 ) > $O/r06_c1_tests.log
( timeout 200 python tools/bench_norm.py tools/_scratch/libs/libleco_hip_r05.so 2>&1 | grep -v Warn ) > $O/r06_c1_bench_norm.txt
( timeout 300 python tools/bench_gemm_plain.py tools/_scratch/libs/libleco_hip_r05.so 2>&1 | grep -v Warn ) > $O/r06_c1_bench_gemm_plain.txt
( timeout 200 python tools/plan_profile.py --list denoise --top 60 2>/dev/null ) > $O/r06_c1_plan_denoise.txt
( timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dominant 2>/dev/null | tail -1 ) > $O/r06_c1_bench.json
tail -3 $O/r06_c1_tests.log; tail -12 $O/r06_c1_bench_norm.txt; tail -8 $O/r06_c1_bench_gemm_plain.txt; head -3 $O/r06_c1_plan_denoise.txt; cut -c1-400 $O/r06_c1_bench.json
