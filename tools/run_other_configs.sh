#!/bin/bash
# BASELINE configs 3-5 (SD2.1-768 v-pred, SDXL 1024 rank 16, SD1.5 c3lier rank 8 batch 4) in one GPU call:
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/run_other_configs.sh [notune]'
# (re-)tunes their GEMM launch shapes into leco_amd/gemm_tune_gfx950.json (skip with `notune`), then benches each;
# results land in gpurun_out/r02_bench_*.json, the merged table in gpurun_out/gemm_tune_gfx950.json.
R=$PWD; O=$R/gpurun_out; mkdir -p $O
T=leco_amd/gemm_tune_gfx950.json
if [ "$1" != "notune" ]; then
timeout 200 python tools/tune_report.py --arch sd21 --res 768 --bs 2 --rank 4 --out $T > $O/r02_tune_sd21.txt 2>/dev/null; tail -2 $O/r02_tune_sd21.txt
timeout 250 python tools/tune_report.py --arch sdxl --res 1024 --bs 1 --rank 16 --out $T > $O/r02_tune_sdxl.txt 2>/dev/null; tail -2 $O/r02_tune_sdxl.txt
timeout 200 python tools/tune_report.py --arch sd15 --res 512 --bs 4 --rank 8 --c3lier --out $T > $O/r02_tune_c3lier.txt 2>/dev/null; tail -2 $O/r02_tune_c3lier.txt
cp $T $O/gemm_tune_gfx950.json
fi
( timeout 200 python bench.py --arch sd21 --res 768 --v-pred --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 ) > $O/r02_bench_sd21_768.json
( timeout 300 python bench.py --arch sdxl --res 1024 --bs 1 --rank 16 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 ) > $O/r02_bench_sdxl_1024.json
( timeout 200 python bench.py --bs 4 --rank 8 --c3lier --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 ) > $O/r02_bench_sd15_c3lier_bs4.json
for f in r02_bench_sd21_768 r02_bench_sdxl_1024 r02_bench_sd15_c3lier_bs4; do python - $O/$f.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read()); print(sys.argv[1].split('/')[-1], round(d['value'],3),'steps/s', round(d['ms_per_step'],1),'ms k_mean',d['config']['k_mean'],'whole-step frac',round(d['roofline']['whole_step']['frac'],3),'loss',d['config']['loss'], 'dom', d['roofline']['kernel'].get('name'), round(d['roofline'].get('frac',0),3))
except Exception as e: print(sys.argv[1], 'FAILED', e)
PY
done
