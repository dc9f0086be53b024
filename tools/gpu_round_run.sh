#!/bin/bash
# Round artefacts on the GPU box (profiles/ files are copied from gpurun_out/ afterwards):
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/gpu_round_run.sh [measure|tests|configs]'
# measure: 1. kernel trace of the benchmark command at the driver's --steps 20 --warmup 5 (--no-dominant: the step's own launches only) -> rNN_step_kernel_stats.txt
#             (its top row names the dominant kernel)
#          2. counter passes (separate runs, --pmc only) over exactly the dominant kernel's launches of one step
#             (`bench.py --dominant-only`) -> rNN_pmc_dominant_{mfma,fetch}.csv (headers carry the kernel-source hash that
#             bench.py checks before quoting them); and over a whole eager step (k = 4)
#          3. the benchmark itself (reads the files of 1 and 2)  -> rNN_bench.json
#          4. every launch of the denoising pass in isolation   -> rNN_plan_denoise.txt;  smoke()
# tests:   the GPU test tier                                    -> rNN_gpu_tests.log
# configs: BASELINE configs 3-5: launch-shape tuning + bench    -> rNN_bench_*.json, rNN_tune_*.txt
RN=${ROUND:-r06}
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
what=${1:-measure}
if [ "$what" = "measure" ]; then
cd /tmp
timeout 420 rocprofv3 --kernel-trace --stats -d /tmp/prof -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dominant --no-dedup > /tmp/prof.log 2>&1
DB=$(find /tmp/prof -name "*.db" | head -1)
{ echo "# cd /tmp && rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dominant --no-dedup   (tools/gpu_round_run.sh; the driver's own --steps/--warmup, so the k sequence and launch mix are the benchmark's)"
  echo "# SD1.5 bs=2 512^2 rank-4 LECO step; summarised from the rocpd database with tools/rocpd_stats.py"
  echo "# csrc_sha1=$(cd $R && python -c 'import bench; print(bench.kernel_sources_hash())')"
  python $R/tools/rocpd_stats.py $DB 60; } > $O/${RN}_step_kernel_stats.txt 2>&1
cp $O/${RN}_step_kernel_stats.txt $R/profiles/${RN}_step_kernel_stats.txt       # bench.py --dominant-only reads the top row
timeout 100 rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/pmc_dm -o dom -- python $R/bench.py --dominant-only --no-cpu-baseline > /tmp/pmc_dm.log 2>&1
python $R/tools/pmc_summary.py /tmp/pmc_dm > $O/${RN}_pmc_dominant_mfma.csv 2>&1
timeout 100 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d /tmp/pmc_df -o dom -- python $R/bench.py --dominant-only --no-cpu-baseline > /tmp/pmc_df.log 2>&1
python $R/tools/pmc_summary.py /tmp/pmc_df > $O/${RN}_pmc_dominant_fetch.csv 2>&1
cp $O/${RN}_pmc_dominant_mfma.csv $O/${RN}_pmc_dominant_fetch.csv $R/profiles/
timeout 130 rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/pmc_mfma -o bench -- python $R/bench.py --steps 1 --warmup 0 --k 4 --no-cpu-baseline --no-graphs --no-dominant --no-dedup > /tmp/pmc1.log 2>&1
python $R/tools/pmc_summary.py /tmp/pmc_mfma > $O/${RN}_pmc_mfma_step.csv 2>&1
timeout 110 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d /tmp/pmc_fetch -o bench -- python $R/bench.py --steps 1 --warmup 0 --k 4 --no-cpu-baseline --no-graphs --no-dominant --no-dedup > /tmp/pmc2.log 2>&1
python $R/tools/pmc_summary.py /tmp/pmc_fetch > $O/${RN}_pmc_fetch_step.csv 2>&1
cd $R
( timeout 900 python bench.py --steps 20 --warmup 5 --dump-shapes $O/${RN}_dominant_shapes.txt 2>/dev/null | tail -1 ) > $O/${RN}_bench.json
( timeout 150 python tools/plan_profile.py --list denoise --top 45 2>/dev/null ) > $O/${RN}_plan_denoise.txt
( timeout 150 python tools/plan_profile.py --list frozen --top 30 2>/dev/null ) > $O/${RN}_plan_frozen.txt
( timeout 100 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 ) > $O/${RN}_smoke.log 2>&1
cut -c1-1800 $O/${RN}_bench.json; head -14 $O/${RN}_step_kernel_stats.txt; head -4 $O/${RN}_pmc_dominant_mfma.csv; head -4 $O/${RN}_pmc_dominant_fetch.csv; tail -3 /tmp/pmc_dm.log; cat $O/${RN}_smoke.log
fi
if [ "$what" = "trace" ]; then      # the kernel trace + the dominant kernel's counter passes only (after a source change that does not alter the product build)
cd /tmp
timeout 420 rocprofv3 --kernel-trace --stats -d /tmp/prof -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dominant --no-dedup > /tmp/prof.log 2>&1
DB=$(find /tmp/prof -name "*.db" | head -1)
{ echo "# cd /tmp && rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dominant --no-dedup   (tools/gpu_round_run.sh; the driver's own --steps/--warmup, so the k sequence and launch mix are the benchmark's)"
  echo "# SD1.5 bs=2 512^2 rank-4 LECO step; summarised from the rocpd database with tools/rocpd_stats.py"
  echo "# csrc_sha1=$(cd $R && python -c 'import bench; print(bench.kernel_sources_hash())')"
  python $R/tools/rocpd_stats.py $DB 60; } > $O/${RN}_step_kernel_stats.txt 2>&1
cp $O/${RN}_step_kernel_stats.txt $R/profiles/${RN}_step_kernel_stats.txt
timeout 100 rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/pmc_dm -o dom -- python $R/bench.py --dominant-only --no-cpu-baseline > /tmp/pmc_dm.log 2>&1
python $R/tools/pmc_summary.py /tmp/pmc_dm > $O/${RN}_pmc_dominant_mfma.csv 2>&1
timeout 100 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d /tmp/pmc_df -o dom -- python $R/bench.py --dominant-only --no-cpu-baseline > /tmp/pmc_df.log 2>&1
python $R/tools/pmc_summary.py /tmp/pmc_df > $O/${RN}_pmc_dominant_fetch.csv 2>&1
cd $R
head -16 $O/${RN}_step_kernel_stats.txt; head -4 $O/${RN}_pmc_dominant_mfma.csv; head -4 $O/${RN}_pmc_dominant_fetch.csv
fi
if [ "$what" = "tests" ]; then
  timeout 1300 python -m pytest tests -q -m gpu -s > /tmp/gpu_tests_full.log 2>&1; echo "pytest rc=$?" >> /tmp/gpu_tests_full.log
  { grep -vE "^W2026|Warn|warn|hipGraph|\^~|^ +[0-9]+ \|" /tmp/gpu_tests_full.log | tail -150; grep -E " passed| failed|^FAILED|^ERROR|pytest rc=" /tmp/gpu_tests_full.log | tail -12; } > $O/${RN}_gpu_tests.log 2>&1
  tail -5 $O/${RN}_gpu_tests.log
fi
if [ "$what" = "configs" ]; then
T=leco_amd/gemm_tune_gfx950.json
timeout 200 python tools/tune_report.py --arch sd21 --res 768 --bs 2 --rank 4 --out $T > $O/${RN}_tune_sd21.txt 2>/dev/null; tail -2 $O/${RN}_tune_sd21.txt
timeout 250 python tools/tune_report.py --arch sdxl --res 1024 --bs 1 --rank 16 --out $T > $O/${RN}_tune_sdxl.txt 2>/dev/null; tail -2 $O/${RN}_tune_sdxl.txt
timeout 200 python tools/tune_report.py --arch sd15 --res 512 --bs 4 --rank 8 --c3lier --out $T > $O/${RN}_tune_c3lier.txt 2>/dev/null; tail -2 $O/${RN}_tune_c3lier.txt
timeout 300 python tools/tune_report.py --arch sd15 --res 512 --bs 2 --rank 4 --dedup --out $T > $O/${RN}_tune_sd15.txt 2>/dev/null; tail -2 $O/${RN}_tune_sd15.txt
cp $T $O/gemm_tune_gfx950.json
( timeout 400 python bench.py --arch sd21 --res 768 --v-pred --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 ) > $O/${RN}_bench_sd21_768.json
( timeout 600 python bench.py --arch sdxl --res 1024 --bs 1 --rank 16 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 ) > $O/${RN}_bench_sdxl_1024.json
( timeout 400 python bench.py --bs 4 --rank 8 --c3lier --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 ) > $O/${RN}_bench_sd15_c3lier_bs4.json
( timeout 200 python bench.py --no-cpu-baseline --no-dominant 2>/dev/null | tail -1 ) > $O/${RN}_bench_after_tune.json
for f in ${RN}_bench_sd21_768 ${RN}_bench_sdxl_1024 ${RN}_bench_sd15_c3lier_bs4 ${RN}_bench_after_tune; do python - $O/$f.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read()); print(sys.argv[1].split('/')[-1], round(d['value'],3),'steps/s', round(d['ms_per_step'],1),'ms k_mean',d['config']['k_mean'],'whole-step frac',round(d['roofline']['whole_step']['frac'],3),'loss',d['config']['loss'], 'dom', d['roofline']['kernel'].get('name'), round(d['roofline'].get('frac',0),3))
except Exception as e: print(sys.argv[1], 'FAILED', e)
PY
done
fi
if [ "$what" = "configs_bench" ]; then      # BASELINE configs 3-5, 20 timed steps, full accounting, the committed launch-shape table (no re-tuning)
( timeout 400 python bench.py --arch sd21 --res 768 --v-pred --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 ) > $O/${RN}_bench_sd21_768.json
( timeout 600 python bench.py --arch sdxl --res 1024 --bs 1 --rank 16 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 ) > $O/${RN}_bench_sdxl_1024.json
( timeout 400 python bench.py --bs 4 --rank 8 --c3lier --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 ) > $O/${RN}_bench_sd15_c3lier_bs4.json
for f in ${RN}_bench_sd21_768 ${RN}_bench_sdxl_1024 ${RN}_bench_sd15_c3lier_bs4; do python - $O/$f.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read()); print(sys.argv[1].split('/')[-1], round(d['value'],3),'steps/s', round(d['ms_per_step'],1),'ms dedup', round(d['dedup']['value'],3), 'k_mean',d['config']['k_mean'],'whole-step frac',round(d['roofline']['whole_step']['frac'],3),'loss',d['config']['loss'], 'dom', d['roofline']['kernel'].get('name'), round(d['roofline'].get('frac',0),3), d['roofline'].get('kernel_sources'))
except Exception as e: print(sys.argv[1], 'FAILED', e)
PY
done
fi
if [ "$what" = "configs_quick" ]; then      # BASELINE configs 3-5 with the committed launch-shape table (no re-tuning)
( timeout 200 python bench.py --arch sd21 --res 768 --v-pred --steps 6 --warmup 2 --no-cpu-baseline --no-dominant 2>/dev/null | tail -1 ) > $O/${RN}_bench_sd21_768.json
( timeout 300 python bench.py --arch sdxl --res 1024 --bs 1 --rank 16 --steps 6 --warmup 2 --no-cpu-baseline --no-dominant 2>/dev/null | tail -1 ) > $O/${RN}_bench_sdxl_1024.json
( timeout 200 python bench.py --bs 4 --rank 8 --c3lier --steps 6 --warmup 2 --no-cpu-baseline --no-dominant 2>/dev/null | tail -1 ) > $O/${RN}_bench_sd15_c3lier_bs4.json
for f in ${RN}_bench_sd21_768 ${RN}_bench_sdxl_1024 ${RN}_bench_sd15_c3lier_bs4; do python - $O/$f.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read()); b=d.get('box_calibration',{})
    print(sys.argv[1].split('/')[-1], round(d['value'],3),'steps/s', round(d['ms_per_step'],1),'ms k_mean',d['config']['k_mean'],'whole-step frac',round(d['roofline']['whole_step']['frac'],3),'loss',d['config']['loss'], 'box', {k: round(v,1) for k,v in b.items() if isinstance(v,(int,float))})
except Exception as e: print(sys.argv[1], 'FAILED', e)
PY
done
fi
