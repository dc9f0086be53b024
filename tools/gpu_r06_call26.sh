#!/bin/bash
# KS2 A/B: two K groups of 64-row wave tiles in conv_patch_kernel<128, *>
mkdir -p gpurun_out
{
echo "# LECO_CONV_KS2=0"; LECO_CONV_KS2=0 python tools/ablate_conv.py --case
echo "# LECO_CONV_KS2=1"; LECO_CONV_KS2=1 python tools/ablate_conv.py --case
} > gpurun_out/r06_c26_conv.txt 2>&1
LECO_CONV_KS2=1 python -m pytest tests/test_kernels.py -q -m gpu -k conv > gpurun_out/r06_c26_tests.log 2>&1
LECO_CONV_KS2=0 python bench.py --steps 10 --warmup 3 > gpurun_out/r06_c26_bench0.json 2> gpurun_out/r06_c26_bench0.err
LECO_CONV_KS2=1 python bench.py --steps 10 --warmup 3 > gpurun_out/r06_c26_bench1.json 2> gpurun_out/r06_c26_bench1.err
tail -3 gpurun_out/r06_c26_tests.log; cat gpurun_out/r06_c26_conv.txt
python - <<'PY'
import json
for i in (0,1):
    try:
        d=json.loads(open(f"gpurun_out/r06_c26_bench{i}.json").read().strip().splitlines()[-1])
        print(i, d["value"], d["ms_per_step"], d.get("dedup",{}).get("value"))
    except Exception as e: print(i, "ERR", e)
PY
