#!/bin/bash
# Scratch experiments on the GPU box (outputs under gpurun_out/exp_*): A/B of switches on the denoising pass and a short bench each.
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
( timeout 400 python -m pytest tests/test_kernels.py -q -m gpu -k "groupnorm or layernorm" 2>&1 | tail -3 ) > $O/exp_tests.log 2>&1
for v in 0 1; do
  ( LECO_GN_REGS=$v timeout 150 python tools/plan_profile.py --list denoise --top 80 2>/dev/null ) > $O/exp_plan_gnregs_$v.txt
  ( LECO_GN_REGS=$v timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-dominant 2>/dev/null | tail -1 | cut -c1-260 ) > $O/exp_bench_gnregs_$v.json
done
cat $O/exp_tests.log; for v in 0 1; do head -2 $O/exp_plan_gnregs_$v.txt | tail -1; grep "launches.*groupnorm_fwd\|launches.*layernorm" $O/exp_plan_gnregs_$v.txt; cat $O/exp_bench_gnregs_$v.json; echo; done
