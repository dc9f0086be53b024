#!/bin/bash
# Scratch: re-measure the launch shapes of the c3lier configuration (the K-extension convolutions may now take the patch kernel) + bench
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
T=leco_amd/gemm_tune_gfx950.json
( timeout 300 python -m pytest tests/test_kernels.py -q -m gpu -k "patch_staged" 2>&1 | tail -3 ) > $O/exp_tests.log 2>&1
timeout 250 python tools/tune_report.py --arch sd15 --res 512 --bs 4 --rank 8 --c3lier --out $T > $O/r04_tune_c3lier.txt 2>/dev/null; tail -2 $O/r04_tune_c3lier.txt
cp $T $O/gemm_tune_gfx950.json
( timeout 200 python bench.py --bs 4 --rank 8 --c3lier --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 ) > $O/r04_bench_sd15_c3lier_bs4.json
cat $O/exp_tests.log; cut -c1-330 $O/r04_bench_sd15_c3lier_bs4.json
