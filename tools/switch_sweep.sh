#!/bin/bash
# One GPU call that measures the launch-shape switches on the whole step (each bench ~35 s):
#   /usr/local/graft/bin/gpurun --timeout 600 -- 'bash tools/switch_sweep.sh'
# Prints "<setting>  <steps/s>  <ms/step>  loss" per line; the first line is the default configuration.
run() {
    local label="$1"; shift
    local out
    out=$(env "$@" python bench.py --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | tail -1)
    python - "$label" "$out" <<'PY'
import json, sys
try:
    d = json.loads(sys.argv[2])
    print(f"{sys.argv[1]:44s} {d['value']:.3f} steps/s  {d['ms_per_step']:.1f} ms  loss {d['config']['loss']:.3e}")
except Exception as e:
    print(f"{sys.argv[1]:44s} FAILED {e!r} {sys.argv[2][:200]}")
PY
}
run "default (W4 for plain grids >= 512)" LECO_NOP=1
run "W4 never" LECO_GEMM_W4_MIN_BLOCKS=1000000000
run "W4 blocks>=257" LECO_GEMM_W4_MIN_BLOCKS=257
run "attention QF=1 everywhere" LECO_ATTN_QF=1
run "attention QF=2 everywhere" LECO_ATTN_QF=2
