#!/bin/bash
# round 6, call 3: GPU tests of the changed parts, the bench line with the dedup loop, GroupNorm producer-statistics rule A/B,
# and one A/B per default-off switch on SDXL (verdict item 8)
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_unet.py tests/test_kernels.py -q -m gpu -x -k "dedup or groupnorm or layernorm or fused_step or producer or splitk" 2>&1 | tail -6 ) > $O/r06_c3_tests.log
( timeout 500 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dominant 2>/dev/null | tail -1 ) > $O/r06_c3_bench.json
( LECO_GN_FUSED=auto3 timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dominant --no-dedup 2>/dev/null | tail -1 ) > $O/r06_c3_bench_gn_auto3.json
( timeout 200 python tools/plan_profile.py --list denoise --top 60 2>/dev/null ) > $O/r06_c3_plan_denoise.txt
XL="--arch sdxl --res 1024 --bs 1 --rank 16 --steps 6 --warmup 2 --no-cpu-baseline --no-dominant --no-dedup --no-telemetry"
( timeout 400 python bench.py $XL 2>/dev/null | tail -1 ) > $O/r06_c3_sdxl_default.json
( LECO_LNFOLD=1 timeout 400 python bench.py $XL 2>/dev/null | tail -1 ) > $O/r06_c3_sdxl_lnfold.json
( LECO_GN_FINISH=1 timeout 400 python bench.py $XL 2>/dev/null | tail -1 ) > $O/r06_c3_sdxl_gnfinish.json
( LECO_FORK=1 timeout 400 python bench.py $XL 2>/dev/null | tail -1 ) > $O/r06_c3_sdxl_fork.json
( LECO_OVERLAP_FROZEN=1 timeout 400 python bench.py $XL 2>/dev/null | tail -1 ) > $O/r06_c3_sdxl_overlap.json
( LECO_XGEMM_MAX_M=1024 timeout 400 python bench.py $XL 2>/dev/null | tail -1 ) > $O/r06_c3_sdxl_xgemm1024.json
( timeout 400 python bench.py $XL 2>/dev/null | tail -1 ) > $O/r06_c3_sdxl_default_b.json
cat $O/r06_c3_tests.log
for f in r06_c3_bench r06_c3_bench_gn_auto3 r06_c3_sdxl_default r06_c3_sdxl_lnfold r06_c3_sdxl_gnfinish r06_c3_sdxl_fork r06_c3_sdxl_overlap r06_c3_sdxl_xgemm1024 r06_c3_sdxl_default_b; do python - $O/$f.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read()); dd=d.get('dedup') or {}
    print(sys.argv[1].split('/')[-1], round(d['value'],3),'steps/s', round(d['ms_per_step'],1),'ms k_mean',d['config']['k_mean'],'loss',d['config']['loss'], 'dedup', round(dd.get('value',0),3), round(dd.get('ms_per_step',0),1), dd.get('loss'))
except Exception as e: print(sys.argv[1], 'FAILED', e)
PY
done
head -3 $O/r06_c3_plan_denoise.txt
