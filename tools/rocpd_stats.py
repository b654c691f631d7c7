#!/usr/bin/env python
"""Per-kernel summary (count, total, avg, min, max, %) from a rocprofv3 rocpd SQLite database
(`rocprofv3 --kernel-trace --stats -d DIR -o NAME` writes NAME_results.db).
    python tools/rocpd_stats.py gpurun_out/prof/r1_results.db [--csv out.csv] [--grid]
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\[clone.*?\]", "", name)
    m = re.match(r"void\s+(.*)", name)
    if m:
        name = m.group(1)
    return name[:110]


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
    sym_cols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    name_col = "display_name" if "display_name" in sym_cols else "kernel_name"
    q = (f"select s.{name_col}, d.end - d.start, d.grid_size_x, d.workgroup_size_x from rocpd_kernel_dispatch d "
         "join rocpd_info_kernel_symbol s on d.kernel_id = s.id")
    agg = {}
    for name, dur, gx, wx in cur.execute(q):
        key = short(name) + (f"  grid={gx}/{wx}" if "--grid" in sys.argv else "")
        a = agg.setdefault(key, [0, 0, 10 ** 18, 0])
        a[0] += 1
        a[1] += dur
        a[2] = min(a[2], dur)
        a[3] = max(a[3], dur)
    total = sum(a[1] for a in agg.values())
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    lines = ["name,calls,total_us,avg_us,min_us,max_us,percent"]
    for k, (n, t, mn, mx) in rows:
        lines.append(f"\"{k}\",{n},{t / 1e3:.1f},{t / n / 1e3:.2f},{mn / 1e3:.2f},{mx / 1e3:.2f},{100 * t / total:.2f}")
    out = "\n".join(lines)
    if "--csv" in sys.argv:
        open(sys.argv[sys.argv.index("--csv") + 1], "w").write(out + "\n")
    print(out)
    print(f"# total kernel time {total / 1e6:.3f} ms over {sum(a[0] for a in agg.values())} dispatches")


if __name__ == "__main__":
    main()
