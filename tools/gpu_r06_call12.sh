#!/bin/bash
# round 6, call 12: GroupNorm with one group per block (8- / 4-byte vectors): kernel tests, micro-benchmark vs round 5, whole-step A/B
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_kernels.py tests/test_table_shapes.py -q -m gpu -x -k "groupnorm or layernorm or table" 2>&1 | tail -4 ) > $O/r06_c12_tests.log; cat $O/r06_c12_tests.log
( timeout 250 python tools/bench_norm.py tools/_scratch/libs/libleco_hip_r05.so 2>&1 | grep -v "Warn\|amdgpu.ids" ) > $O/r06_c12_bench_norm.txt; grep -E "B= 4|^# sum|layernorm" $O/r06_c12_bench_norm.txt | cut -c1-200
show() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read()); dd=d.get('dedup') or {}
    print(sys.argv[2], round(d['value'],3), 'steps/s', round(d['ms_per_step'],2), 'ms; dedup', round(dd.get('value',0),3), [f"{x:.4g}" for x in d['config']['losses'][:4]])
except Exception as e: print(sys.argv[2], 'FAILED', e)
PY
}
for i in 1 2; do
( timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dominant 2>/dev/null | tail -1 ) > $O/r06_c12_bench_vec_$i.json; show $O/r06_c12_bench_vec_$i.json "narrow-vector GroupNorm #$i"
( LECO_GN_VEC=8 timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dominant --no-dedup 2>/dev/null | tail -1 ) > $O/r06_c12_bench_vec8_$i.json; show $O/r06_c12_bench_vec8_$i.json "LECO_GN_VEC=8 #$i"
done
