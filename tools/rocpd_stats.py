"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into a per-kernel table."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"leco::\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*\)$", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:100]


def main(path, top=45):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    rows = db.execute("select name, grid_x, grid_y, grid_z, start, end from kernels").fetchall()
    agg = {}
    for name, gx, gy, gz, s, e in rows:
        a = agg.setdefault(short(name), [0, 0.0, 1e30, 0.0])
        d = (e - s) / 1e3
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    print(f"# {len(rows)} kernel dispatches, {tot/1e3:.1f} ms total GPU kernel time")
    # timeline occupancy: union of the kernel intervals vs the span they cover, and the idle gaps between
    # consecutive dispatches (launch / dependency bubbles inside a graph replay show up here)
    iv = sorted((s, e, short(n)) for n, _, _, _, s, e in rows)
    busy, gaps, cur_s, cur_e, cur_n = 0, [], iv[0][0], iv[0][1], iv[0][2]
    pair = {}
    for s, e, n in iv[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append(s - cur_e)
            if s - cur_e < 50_000:
                a = pair.setdefault((cur_n[:48], n[:48]), [0, 0])
                a[0] += 1; a[1] += s - cur_e
            cur_s, cur_e, cur_n = s, e, n
        else:
            if e > cur_e:
                cur_e, cur_n = e, n
    busy += cur_e - cur_s
    small = [g for g in gaps if g < 50_000]
    print(f"# timeline: span {(iv[-1][1]-iv[0][0])/1e6:.1f} ms, busy {busy/1e6:.1f} ms; {len(small)} gaps < 50 us "
          f"totalling {sum(small)/1e6:.1f} ms (mean {sum(small)/max(1,len(small))/1e3:.2f} us); "
          f"{len(gaps)-len(small)} longer gaps totalling {(sum(gaps)-sum(small))/1e6:.1f} ms")
    for (a, b), (cnt, tot_ns) in sorted(pair.items(), key=lambda kv: -kv[1][1])[:8]:
        print(f"#   gap after {a}  ->  {b}: {cnt} x {tot_ns/cnt/1e3:.1f} us = {tot_ns/1e6:.2f} ms")
    print(f"{'%':>6} {'calls':>8} {'total_ms':>10} {'avg_us':>9} {'min_us':>8} {'max_us':>9}  kernel")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"{100*a[1]/tot:6.2f} {a[0]:8d} {a[1]/1e3:10.2f} {a[1]/a[0]:9.1f} {a[2]:8.1f} {a[3]:9.1f}  {k}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 45)
