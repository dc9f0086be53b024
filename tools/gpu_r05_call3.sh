#!/bin/bash
# Round-5 GPU call 3: the A-stationary GEMM (csrc/xgemm.hip) -- parity on gfx950, per-launch A/B of the denoising and the
# batched frozen pass with and without it, the step with it.
RN=${ROUND:-r05}
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_kernels.py tests/test_stripe.py -q -x -m gpu -k "xgemm or step_glue or a_stationary" -s 2>&1 | grep -vE "Warn|warn" | tail -8 ) > $O/${RN}_xgemm_tests.log
cat $O/${RN}_xgemm_tests.log
for v in 1 0; do
  ( LECO_XGEMM=$v timeout 200 python tools/plan_profile.py --list denoise --top 60 2>/dev/null ) > $O/${RN}_plan_denoise_xgemm$v.txt
  head -1 $O/${RN}_plan_denoise_xgemm$v.txt; grep -E "^#.*(gemm|xgemm)" $O/${RN}_plan_denoise_xgemm$v.txt
done
grep -E "xgemm" $O/${RN}_plan_denoise_xgemm1.txt | head -20
( LECO_XGEMM=1 LECO_XGEMM_MAX_M=100000 timeout 200 python tools/plan_profile.py --list frozen --top 60 2>/dev/null ) > $O/${RN}_plan_frozen_xgemm1.txt
( LECO_XGEMM=0 timeout 200 python tools/plan_profile.py --list frozen --top 60 2>/dev/null ) > $O/${RN}_plan_frozen_xgemm0.txt
head -1 $O/${RN}_plan_frozen_xgemm1.txt; head -1 $O/${RN}_plan_frozen_xgemm0.txt
grep -E "xgemm" $O/${RN}_plan_frozen_xgemm1.txt | head -20
( timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dominant 2>/dev/null | tail -1 ) > $O/${RN}_bench_xgemm.json
python - $O/${RN}_bench_xgemm.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
print("bench xgemm: value", round(d["value"], 3), "ms/step", round(d["ms_per_step"], 1), "loss", d["config"]["loss"])
PY
