#!/bin/bash
RN=${ROUND:-r05}
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
bench() {
  tag=$1; shift
  ( env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dominant 2>/dev/null | tail -1 ) > $O/${RN}_bench_$tag.json
  python - $O/${RN}_bench_$tag.json $tag <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
print("bench", sys.argv[2], "value", round(d["value"], 3), "ms/step", round(d["ms_per_step"], 2), "loss", d["config"]["loss"])
PY
}
bench lnfold_qkvq_a LECO_LNFOLD=qkv,q
bench lnfold_off_a LECO_LNFOLD=0
bench lnfold_qkvq_b LECO_LNFOLD=qkv,q
bench lnfold_off_b LECO_LNFOLD=0
