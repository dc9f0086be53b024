#!/bin/bash
# round 6, call 18: softmax row reductions through v_permlane16/32_swap instead of ds_bpermute: tests, isolated attention timings, bench
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_kernels.py tests/test_stripe.py tests/test_table_shapes.py -q -m gpu -x -k "attention or attn or stripe or xblock or tail or head or table" 2>&1 | tail -3 ) > $O/r06_c18_tests.log; cat $O/r06_c18_tests.log
( timeout 200 python tools/bench_small.py 2>&1 | grep -A12 "attention fwd" ) > $O/r06_c18_attn_fwd.txt; cat $O/r06_c18_attn_fwd.txt
( LECO_HIP_LIB=$R/tools/_scratch/libs/libleco_hip_r05.so timeout 200 python tools/bench_small.py 2>&1 | grep -A12 "attention fwd" ) > $O/r06_c18_attn_fwd_r05.txt; echo "round-5 library:"; cat $O/r06_c18_attn_fwd_r05.txt
( timeout 200 python tools/plan_profile.py --list denoise --top 12 2>/dev/null | grep -E "attn_fwd|xblock|^# sd15" ) > $O/r06_c18_plan.txt; cat $O/r06_c18_plan.txt
( timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dominant 2>/dev/null | tail -1 ) > $O/r06_c18_bench.json
python - $O/r06_c18_bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read()); dd=d.get('dedup') or {}
print(round(d['value'],3), 'steps/s', round(d['ms_per_step'],2), 'ms; dedup', round(dd.get('value',0),3), [f"{x:.4g}" for x in d['config']['losses'][:4]])
PY
