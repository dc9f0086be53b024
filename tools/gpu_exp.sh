#!/bin/bash
# Scratch experiments on the GPU box (outputs under gpurun_out/exp_*): A/B of switches on the denoising pass and a short bench each.
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_kernels.py -q -m gpu -k "conv_in_out or attention" 2>&1 | tail -3 ) > $O/exp_tests.log 2>&1
for v in 1; do
  ( LECO_ATTN_DMA=$v timeout 150 python tools/plan_profile.py --list denoise --top 75 2>/dev/null ) > $O/exp_plan_attn_$v.txt
  ( LECO_ATTN_DMA=$v timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-dominant 2>/dev/null | tail -1 | cut -c1-260 ) > $O/exp_bench_attn_$v.json
done
cat $O/exp_tests.log; for v in 1; do head -2 $O/exp_plan_attn_$v.txt | tail -1; grep "attn_fwd B=4 H=8 Sq=4096\|attn_fwd B=2 H=8 Sq=4096\|conv_out\|conv_in" $O/exp_plan_attn_$v.txt; cat $O/exp_bench_attn_$v.json; echo; done
