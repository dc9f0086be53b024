#!/bin/bash
# loader waves, whole step, order-balanced: LW = 1, 0, 0, 1, 1, 0
mkdir -p gpurun_out
i=0
for lw in 1 0 0 1 1 0; do
  i=$((i+1))
  LECO_CONV_LW=$lw python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-dominant > gpurun_out/r06_c31_bench_${i}_lw${lw}.json 2> gpurun_out/r06_c31_bench_${i}.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r06_c31_bench_${i}_lw${lw}.json").read().strip().splitlines()[-1])
print("run ${i} LW=${lw}", round(d["ms_per_step"],2), round(d.get("dedup",{}).get("ms_per_step",0),2))
PY
done
