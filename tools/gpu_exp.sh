#!/bin/bash
# Scratch experiments on the GPU box (outputs under gpurun_out/exp_*): A/B of switches on the denoising pass and a short bench each.
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
( timeout 400 python -m pytest tests/test_stripe.py -q -m gpu -k "forked or forward_only or shared_prefix" 2>&1 | tail -5 ) > $O/exp_tests.log 2>&1
for v in 0 1; do
  ( LECO_FORK=$v timeout 150 python tools/plan_profile.py --list denoise --top 5 2>/dev/null | head -3 ) > $O/exp_plan_fork_$v.txt
  ( LECO_FORK=$v timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-dominant 2>/dev/null | tail -1 | cut -c1-260 ) > $O/exp_bench_fork_$v.json
done
cat $O/exp_tests.log; for v in 0 1; do head -2 $O/exp_plan_fork_$v.txt | tail -1; cat $O/exp_bench_fork_$v.json; echo; done
