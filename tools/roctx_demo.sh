#!/bin/bash
# roctx ranges of the step phases (LECO_ROCTX=1, leco_amd/trace.py) as rocprofv3 records them:
#   gpurun -- 'bash tools/roctx_demo.sh'   -> gpurun_out/rNN_roctx_ranges.txt
RN=${ROUND:-r03}
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
LECO_ROCTX=1 timeout 200 rocprofv3 --marker-trace --kernel-trace --output-format csv -d /tmp/roctx -o run -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-dominant > /tmp/roctx.log 2>&1
python - <<'PY' > $O/${RN}_roctx_ranges.txt 2>&1
import csv, glob, collections
files = glob.glob("/tmp/roctx/**/*marker*trace*.csv", recursive=True)
print("# LECO_ROCTX=1 rocprofv3 --marker-trace --kernel-trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-dominant")
print("# marker files:", [f.split("/")[-1] for f in files])
agg = collections.OrderedDict()
for f in files:
    for r in csv.DictReader(open(f)):
        name = r.get("Function") or r.get("Name") or r.get("Message") or str(r)
        s, e = int(r.get("Start_Timestamp", 0)), int(r.get("End_Timestamp", 0))
        a = agg.setdefault(name.split(" k=")[0], [0, 0])
        a[0] += 1
        a[1] += e - s
print(f"{'range':40s} {'count':>6s} {'host ms (push..pop)':>20s}")
for k, (n, ns) in agg.items():
    print(f"{k:40s} {n:6d} {ns / 1e6:20.3f}")
PY
cat $O/${RN}_roctx_ranges.txt; tail -2 /tmp/roctx.log
