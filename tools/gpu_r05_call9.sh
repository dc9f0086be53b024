#!/bin/bash
# Round-5 GPU call 9: LayerNorm fold + split-K finish inside GroupNorm -- parity on gfx950, per-launch A/B of the denoising
# pass, the step with the candidates.
RN=${ROUND:-r05}
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
( timeout 400 python -m pytest tests/test_kernels.py tests/test_stripe.py -q -m gpu -k "layernorm_folded or finishes_a_split or fold_layernorm or finish_split_k or xgemm" -s 2>&1 | grep -vE "Warn|warn" | grep -E "rel|passed|failed|FAILED|Error" | tail -14 ) > $O/${RN}_fusion_tests.log
cat $O/${RN}_fusion_tests.log
prof() {  # tag, env...
  tag=$1; shift
  ( env "$@" timeout 200 python tools/plan_profile.py --list denoise --top 70 2>/dev/null ) > $O/${RN}_plan_denoise_$tag.txt
  echo "== $tag: $(head -1 $O/${RN}_plan_denoise_$tag.txt)"
}
prof base LECO_LNFOLD=0 LECO_GN_FINISH=0
prof ln LECO_LNFOLD=1 LECO_GN_FINISH=0
prof gnfin LECO_LNFOLD=0 LECO_GN_FINISH=1
prof both LECO_LNFOLD=1 LECO_GN_FINISH=1
grep -E "lnfold|layernorm|splitk|nofinish" $O/${RN}_plan_denoise_both.txt | head -24
echo "-- base rows of the same sites"
grep -E "layernorm|N=1920 K=640|N=3840 K=1280|N=640 K=640 \+lora32\(fusedT\)$|N=1280 K=1280 \+lora32\(fusedT\)$|geglu" $O/${RN}_plan_denoise_base.txt | head -14
bench() {
  tag=$1; shift
  ( env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dominant 2>/dev/null | tail -1 ) > $O/${RN}_bench_$tag.json
  python - $O/${RN}_bench_$tag.json $tag <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
print("bench", sys.argv[2], "value", round(d["value"], 3), "ms/step", round(d["ms_per_step"], 2), "loss", d["config"]["loss"])
PY
}
bench fuse_both LECO_LNFOLD=1 LECO_GN_FINISH=1
bench fuse_none LECO_LNFOLD=0 LECO_GN_FINISH=0
