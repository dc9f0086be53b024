#!/bin/bash
# round 6, call 16: attention grid walked XCD-contiguously ((sample, head) pairs per XCD): tests, isolated timings A/B, bench A/B
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_kernels.py tests/test_table_shapes.py -q -m gpu -x -k "attention or attn" 2>&1 | tail -3 ) > $O/r06_c16_tests.log; cat $O/r06_c16_tests.log
for v in 0 1; do
  ( LECO_ATTN_XCD=$v timeout 200 python tools/bench_small.py 2>&1 | grep -A12 "attention fwd" ) > $O/r06_c16_attn_fwd_xcd$v.txt; echo "LECO_ATTN_XCD=$v"; cat $O/r06_c16_attn_fwd_xcd$v.txt
  ( LECO_ATTN_XCD=$v timeout 200 python tools/plan_profile.py --list bwd --top 8 2>/dev/null | grep -E "attn_bwd|^# sd15" ) > $O/r06_c16_attn_bwd_xcd$v.txt; cat $O/r06_c16_attn_bwd_xcd$v.txt
done
show() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read())
    print(sys.argv[2], round(d['value'],3), 'steps/s', round(d['ms_per_step'],2), 'ms', [f"{x:.4g}" for x in d['config']['losses'][:4]])
except Exception as e: print(sys.argv[2], 'FAILED', e)
PY
}
for i in 1 2; do
for v in 1 0; do
( LECO_ATTN_XCD=$v timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dominant --no-dedup 2>/dev/null | tail -1 ) > $O/r06_c16_bench_xcd${v}_$i.json; show $O/r06_c16_bench_xcd${v}_$i.json "LECO_ATTN_XCD=$v #$i"
done; done
