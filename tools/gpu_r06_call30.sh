#!/bin/bash
# loader-wave variant of conv_patch_kernel<128, *>: A/B on the conv shapes (with clock stamps), GPU tests, whole step
mkdir -p gpurun_out
python tools/ablate_conv.py > gpurun_out/r06_c30_conv.txt 2>&1
LECO_CONV_LW=1 python -m pytest tests/test_kernels.py -q -m gpu -k "conv and not upsampled_input" > gpurun_out/r06_c30_tests.log 2>&1
LECO_CONV_LW=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r06_c30_bench0.json 2> gpurun_out/r06_c30_bench0.err
LECO_CONV_LW=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r06_c30_bench1.json 2> gpurun_out/r06_c30_bench1.err
tail -3 gpurun_out/r06_c30_tests.log; grep -v amdgpu.ids gpurun_out/r06_c30_conv.txt | cut -c1-230
python - <<'PY'
import json
for i in (0,1):
    try:
        d=json.loads(open(f"gpurun_out/r06_c30_bench{i}.json").read().strip().splitlines()[-1])
        print(i, d["value"], d["ms_per_step"], d.get("dedup",{}).get("value"), d["roofline"]["frac"])
    except Exception as e: print(i, "ERR", e)
PY
