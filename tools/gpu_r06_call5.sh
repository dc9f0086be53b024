#!/bin/bash
# round 6, call 5: the dedup loop's NaN loss -- graphs on / off, after faithful steps or alone, long chains
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read()); dd=d.get('dedup') or {}
    print(sys.argv[1].split('/')[-1], 'faithful losses', [f"{x:.3g}" for x in d['config']['losses']], 'dedup losses', [f"{x:.3g}" for x in dd.get('losses',[])], 'k', d['timing']['k'])
except Exception as e: print(sys.argv[1], 'FAILED', e)
PY
}
( timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-dominant --no-telemetry 2>/dev/null | tail -1 ) > $O/r06_c5_a.json; show $O/r06_c5_a.json
( timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-dominant --no-telemetry --no-graphs 2>/dev/null | tail -1 ) > $O/r06_c5_b.json; show $O/r06_c5_b.json
( timeout 300 python tools/nan_probe.py --dedup --k 45 --steps 3 2>&1 | grep -v Warn | tail -12 ) > $O/r06_c5_probe_k45.txt; cat $O/r06_c5_probe_k45.txt
