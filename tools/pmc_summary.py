"""Aggregate a rocprofv3 `--pmc ... --output-format csv` counter_collection.csv per kernel name."""
import csv
import glob
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"leco::\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*\)$", "", name)
    return re.sub(r"^void ", "", name)[:90]


def main(d):
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not files:
        print("no counter_collection.csv under", d)
        return
    agg = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(set)
    for f in files:
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[k].add(r["Dispatch_Id"])
    names = sorted({c for v in agg.values() for c in v})
    try:      # which kernel sources these counters describe (bench.py refuses summaries taken on other sources)
        import os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        print(f"# csrc_sha1={bench.kernel_sources_hash()}")
    except Exception as e:      # noqa: BLE001 -- the summary itself must never fail on this
        print(f"# csrc_sha1=unknown ({e!r})")
    print("kernel,calls," + ",".join(names))
    key = names[0]
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get(key, 0))[:40]:
        print(f"{k},{len(calls[k])}," + ",".join(f"{v.get(c, 0):.6g}" for c in names))


if __name__ == "__main__":
    main(sys.argv[1])
