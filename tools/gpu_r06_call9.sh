#!/bin/bash
# round 6, call 9: dedup with graphs with the statistics arena zeroed by a kernel node instead of a memset node
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
show() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read()); dd=d.get('dedup') or {}
    print(sys.argv[2], round(d['value'],3), 'steps/s faithful', [f"{x:.4g}" for x in d['config']['losses'][:6]], 'dedup', round(dd.get('value',0),3), [f"{x:.4g}" for x in dd.get('losses',[])[:6]])
except Exception as e: print(sys.argv[2], 'FAILED', e)
PY
}
( timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dominant 2>/dev/null | tail -1 ) > $O/r06_c9_bench.json; show $O/r06_c9_bench.json bench
( timeout 300 python tools/graph_vs_eager.py --dedup --bench-like --plan frozen --which fwd_off 2>&1 | grep -v Warn | tail -5 ) > $O/r06_c9_gve.txt; cut -c1-500 $O/r06_c9_gve.txt
