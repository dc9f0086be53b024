#!/bin/bash
# One GPU call that measures every launch-shape switch on the whole step (each bench ~35 s):
#   /usr/local/graft/bin/gpurun --timeout 600 -- 'bash tools/switch_sweep.sh'
# Prints "<setting>  <steps/s>  <ms/step>" per line; the first line is the default configuration.
run() {
    local label="$1"; shift
    local out
    out=$(env "$@" python bench.py --no-cpu-baseline --steps 12 --warmup 3 2>/dev/null | tail -1)
    python - "$label" "$out" <<'PY'
import json, sys
try:
    d = json.loads(sys.argv[2])
    print(f"{sys.argv[1]:44s} {d['value']:.3f} steps/s  {d['ms_per_step']:.1f} ms  dominant conv {d['roofline']['dominant_kernel'].get('us_per_launch', 0):.1f} us")
except Exception as e:
    print(f"{sys.argv[1]:44s} FAILED {e!r}")
PY
}
run "default" LECO_NOP=1
run "W4 (2 WG/CU, 4 waves) blocks>=2048" LECO_GEMM_W4_MIN_BLOCKS=2048
run "W4 blocks>=1024" LECO_GEMM_W4_MIN_BLOCKS=1024
run "W4 blocks>=512" LECO_GEMM_W4_MIN_BLOCKS=512
run "W4 blocks>=257" LECO_GEMM_W4_MIN_BLOCKS=257
run "persistent GEMM tiles>=512" LECO_GEMM_PERSISTENT_MIN_TILES=512
run "persistent GEMM tiles>=1024" LECO_GEMM_PERSISTENT_MIN_TILES=1024
run "NS2 (2 WG/CU, 8 waves) blocks>384" LECO_GEMM_NS2_MIN_BLOCKS=384
run "attention QF=1 everywhere" LECO_ATTN_QF=1
run "attention QF=2 everywhere" LECO_ATTN_QF=2
