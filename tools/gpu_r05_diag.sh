#!/bin/bash
# Round-5 diagnosis of the round-4 builder/driver gap (205 vs 238 ms per step on the same kernels): the benchmark at the
# driver's flags with per-step HIP events, host-enqueue lead, clock / power / throttle telemetry and the busy fraction;
# then a 60-step sustained run (clock drift) and an eager (no hipGraph) run (launch-path sensitivity).
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/gpu_r05_diag.sh [tag]'
TAG=${1:-lease1}
RN=${ROUND:-r05}
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
{ echo "# host"; nproc; lscpu | grep -E "Model name|Socket|Thread|MHz" ; uptime; echo "# rocm-smi"; rocm-smi --showclocks --showpower --showmaxpower --showperflevel 2>&1 | grep -vE "^=|^$" | head -30; } > $O/${RN}_box_${TAG}.txt 2>&1
python tools/gpu_telemetry.py > $O/${RN}_telemetry_idle_${TAG}.json 2>&1
( timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>$O/${RN}_bench_${TAG}.err | tail -1 ) > $O/${RN}_bench_${TAG}.json
if [ "$2" != "short" ]; then
( timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-dominant 2>/dev/null | tail -1 ) > $O/${RN}_bench_sustained60_${TAG}.json
( timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-dominant --no-graphs 2>/dev/null | tail -1 ) > $O/${RN}_bench_eager_${TAG}.json
fi
python - $O/${RN}_bench_${TAG}.json $O/${RN}_bench_sustained60_${TAG}.json $O/${RN}_bench_eager_${TAG}.json <<'PY'
import json, sys
for p in sys.argv[1:]:
    try:
        d = json.loads(open(p).read())
    except Exception as e:
        print(p.split('/')[-1], "unreadable:", e); continue
    t = d.get("timing", {}); te = d.get("telemetry", {})
    print(p.split('/')[-1], "value", round(d["value"], 3), "ms/step", round(d["ms_per_step"], 1), "k_mean", d["config"]["k_mean"])
    print("  step_ms min/med/max", [round(x, 1) for x in t.get("step_ms_min_median_max", [])], "ms/fe halves", [round(x, 3) for x in t.get("ms_per_forward_equivalent_first_half_vs_second_half", [])])
    print("  host enqueue ms/step", round(t.get("host_enqueue_ms_per_step", 0), 1), "host_lead_ms first/last", t.get("host_lead_ms", [None])[0], t.get("host_lead_ms", [None])[-1])
    print("  busy_frac_mean", t.get("gpu_busy_frac_mean"), "isolated sum ms/step", t.get("isolated_launch_sum_ms_per_step"))
    print("  during", json.dumps(te.get("during_timed")))
    print("  isolated", json.dumps(te.get("during_isolated_launch_timing")))
    print("  acc", json.dumps(te.get("accumulated_over_timed")))
    b = te.get("before_timed") or {}
    print("  before", {k: b.get(k) for k in ("source", "current_gfxclks", "current_socket_power", "power_cap_w", "max_power_cap_w", "gfx_clk_limits_mhz", "temperature_hotspot", "throttle_status", "amdsmi_error", "metrics_error")})
    r = d.get("roofline", {})
    print("  roofline frac", r.get("frac"), "kernel", r.get("kernel", {}).get("name"), "us", r.get("kernel", {}).get("us_per_launch"), "trace us", r.get("kernel", {}).get("us_per_launch_trace"))
PY
