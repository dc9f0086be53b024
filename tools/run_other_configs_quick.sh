O=gpurun_out; mkdir -p $O
( timeout 70 python bench.py --arch sd21 --res 768 --v-pred --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 ) > $O/r02_bench_sd21_768.json
( timeout 70 python bench.py --bs 4 --rank 8 --c3lier --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 ) > $O/r02_bench_sd15_c3lier_bs4.json
( timeout 110 python bench.py --arch sdxl --res 1024 --bs 1 --rank 16 --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 ) > $O/r02_bench_sdxl_1024.json
for f in r02_bench_sd21_768 r02_bench_sd15_c3lier_bs4 r02_bench_sdxl_1024; do python - $O/$f.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read()); print(sys.argv[1].split('/')[-1], round(d['value'],3),'steps/s', round(d['ms_per_step'],1),'ms k_mean',d['config']['k_mean'],'frac',round(d['roofline']['whole_step']['frac'],3))
except Exception as e: print(sys.argv[1], 'FAILED', e)
PY
done
