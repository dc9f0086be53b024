#!/bin/bash
# Stripe-kernel checks on the GPU box:
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/gpu_stripe_run.sh'
# 1. parity of the stripe kernels on gfx950; 2. every launch of the denoising / frozen pass in isolation with and without
# them (LECO_STRIPE=0); 3. short benchmark runs both ways.
RN=${ROUND:-r04}
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_stripe.py -q -m gpu -s 2>&1 | grep -vE "^W2026|Warn|warn" | tail -25 ) > $O/${RN}_stripe_tests.log 2>&1
tail -6 $O/${RN}_stripe_tests.log
( timeout 200 python tools/plan_profile.py --list denoise --top 60 2>&1 | tail -80 ) > $O/${RN}_plan_denoise_stripe.txt
( LECO_STRIPE=0 timeout 200 python tools/plan_profile.py --list denoise --top 60 2>&1 | tail -80 ) > $O/${RN}_plan_denoise_nostripe.txt
( timeout 200 python tools/plan_profile.py --list frozen --top 60 2>&1 | tail -80 ) > $O/${RN}_plan_frozen_stripe.txt
( LECO_STRIPE=0 timeout 200 python tools/plan_profile.py --list frozen --top 60 2>&1 | tail -80 ) > $O/${RN}_plan_frozen_nostripe.txt
head -3 $O/${RN}_plan_denoise_stripe.txt; grep xblock $O/${RN}_plan_denoise_stripe.txt; head -1 $O/${RN}_plan_denoise_nostripe.txt
head -1 $O/${RN}_plan_frozen_stripe.txt; grep xblock $O/${RN}_plan_frozen_stripe.txt; head -1 $O/${RN}_plan_frozen_nostripe.txt
( timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-dominant 2>&1 | tail -1 ) > $O/${RN}_bench_stripe_short.json
( LECO_STRIPE=0 timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-dominant 2>&1 | tail -1 ) > $O/${RN}_bench_nostripe_short.json
for f in ${RN}_bench_stripe_short ${RN}_bench_nostripe_short; do python - $O/$f.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read()); print(sys.argv[1].split('/')[-1], round(d['value'],3),'steps/s', round(d['ms_per_step'],1),'ms k_mean',d['config']['k_mean'],'loss',d['config']['loss'])
except Exception as e: print(sys.argv[1], 'FAILED', e, open(sys.argv[1]).read()[-1500:])
PY
done
