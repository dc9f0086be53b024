#!/bin/bash
# round 6, call 8: dedup with graphs after returning to the round-5 GroupNorm statistics rule
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
show() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read()); dd=d.get('dedup') or {}
    print(sys.argv[2], round(d['value'],3), 'steps/s faithful', [f"{x:.4g}" for x in d['config']['losses'][:6]], 'dedup', round(dd.get('value',0),3), [f"{x:.4g}" for x in dd.get('losses',[])[:6]])
except Exception as e: print(sys.argv[2], 'FAILED', e)
PY
}
( timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dominant 2>/dev/null | tail -1 ) > $O/r06_c8_bench.json; show $O/r06_c8_bench.json bench
( timeout 300 python tools/graph_vs_eager.py --dedup --bench-like --plan frozen --which fwd_off 2>&1 | grep -v Warn | tail -8 ) > $O/r06_c8_gve.txt; cut -c1-600 $O/r06_c8_gve.txt
( timeout 600 python -m pytest tests/test_fullsize.py -q -m gpu -x -k "dedup or two_arith" -s 2>&1 | grep -E "rel_hip|loss=|two arith|passed|failed|Error" | tail -40 ) > $O/r06_c8_fullsize.txt; cat $O/r06_c8_fullsize.txt
