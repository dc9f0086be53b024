#!/bin/bash
# round 6, call 15: GroupNorm blocks walked XCD-contiguously: kernel tests, micro-benchmark vs round 5 (compare the ratios with call 13's)
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_kernels.py tests/test_table_shapes.py -q -m gpu -x -k "groupnorm or layernorm or table" 2>&1 | tail -3 ) > $O/r06_c15_tests.log; cat $O/r06_c15_tests.log
( timeout 250 python tools/bench_norm.py tools/_scratch/libs/libleco_hip_r05.so 2>&1 | grep -v "Warn\|amdgpu.ids" ) > $O/r06_c15_bench_norm.txt; grep -E "groupnorm_fwd B|^# sum" $O/r06_c15_bench_norm.txt | cut -c1-200
( timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dominant 2>/dev/null | tail -1 ) > $O/r06_c15_bench.json
python - $O/r06_c15_bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read()); dd=d.get('dedup') or {}
print(round(d['value'],3), 'steps/s', round(d['ms_per_step'],2), 'ms; dedup', round(dd.get('value',0),3), [f"{x:.4g}" for x in d['config']['losses'][:4]])
PY
