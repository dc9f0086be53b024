#!/bin/bash
# round 6, call 6: bisect the graph-only NaN of the dedup step
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-dominant --no-telemetry"
show() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read()); dd=d.get('dedup') or {}
    print(sys.argv[2], 'faithful', [f"{x:.3g}" for x in d['config']['losses']], 'dedup', [f"{x:.3g}" for x in dd.get('losses',[])])
except Exception as e: print(sys.argv[2], 'FAILED', e)
PY
}
run() { ( env $1 timeout 200 $B 2>/dev/null | tail -1 ) > $O/r06_c6_tmp.json; show $O/r06_c6_tmp.json "$1"; }
run LECO_EAGER_LISTS=fwd_off:4
run LECO_EAGER_LISTS=fwd_on:2
run LECO_EAGER_LISTS=bwd:2
run LECO_EAGER_LISTS=fwd_on:2,bwd:2
run LECO_STRIPE=0
run LECO_SHARE_PREFIX=0
run LECO_GN_FUSED=0
run LECO_XGEMM=0
