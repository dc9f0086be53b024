#!/bin/bash
# round 6, call 19: re-tune SD1.5 with the 128x64 tile among the candidates; bench before / after on the same box
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
T=leco_amd/gemm_tune_gfx950.json
show() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read()); dd=d.get('dedup') or {}
    print(sys.argv[2], round(d['value'],3), 'steps/s', round(d['ms_per_step'],2), 'ms; dedup', round(dd.get('value',0),3), [f"{x:.4g}" for x in d['config']['losses'][:4]])
except Exception as e: print(sys.argv[2], 'FAILED', e)
PY
}
( timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dominant 2>/dev/null | tail -1 ) > $O/r06_c19_bench_before.json; show $O/r06_c19_bench_before.json "before re-tune"
timeout 400 python tools/tune_report.py --arch sd15 --res 512 --bs 2 --rank 4 --dedup --out $T > $O/r06_c19_tune_sd15.txt 2>/dev/null; tail -2 $O/r06_c19_tune_sd15.txt; grep -c "(11," $O/r06_c19_tune_sd15.txt; grep "(11," $O/r06_c19_tune_sd15.txt | head -20
cp $T $O/gemm_tune_gfx950_c19.json
( timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dominant 2>/dev/null | tail -1 ) > $O/r06_c19_bench_after.json; show $O/r06_c19_bench_after.json "after re-tune"
( timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dominant 2>/dev/null | tail -1 ) > $O/r06_c19_bench_after2.json; show $O/r06_c19_bench_after2.json "after re-tune (2)"
